"""Heterogeneous sweep: many bsuite_ids advanced together on one or several GPUs (BASELINE config 5).

The reference runs the sweep as one OS process per bsuite_id (`bsuite/baselines/utils/pool.py:28-54`
mapping `run(bsuite_id)` over `sweep.SWEEP`).  Here every bsuite_id is a *segment* of lanes with its
own environment family and settings; a "sweep step" advances every segment once.  Segments are
independent, so:

* across GPUs — whole segments are bin-packed onto ranks by `lanes x bytes-per-step` (longest
  processing time first); no communication per step; summaries are all-gathered at the end
  (`bsuite_amd.distributed`);
* on one GPU — segment launches are spread round-robin over a few HIP streams and the whole sweep
  step (several hundred small launches) is captured once into a HIP graph and replayed, so the host
  issues one graph launch per sweep step instead of ~10^3 kernel launches.  Call indices of the
  draw stream live in device memory (`device_step_counter=True`) so replays stay reproducible.

Global lane ids are unique across the sweep (segment k starts where segment k-1 ended), so any
assignment of segments to ranks reproduces the same per-lane trajectories.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

import bsuite_amd
from bsuite_amd import sweep as _sweep


def segment_table(bsuite_ids: Sequence[str], total_lanes: int) -> List[Tuple[str, int, int]]:
  """[(bsuite_id, lane_begin, n_lanes)]: total_lanes split evenly, remainder to the last id."""
  n = len(bsuite_ids)
  per = total_lanes // n
  if per < 1:
    raise ValueError('fewer lanes than bsuite_ids')
  table, begin = [], 0
  for i, bid in enumerate(bsuite_ids):
    lanes = per + (total_lanes - per * n if i == n - 1 else 0)
    table.append((bid, begin, lanes))
    begin += lanes
  return table


def bytes_per_step(obs_numel: int) -> int:
  """Algorithmic bytes of one env-step with an 8-byte packed state."""
  return 13 + 4 * obs_numel + 8


def assign_segments(costs: Sequence[float], world_size: int) -> List[int]:
  """Greedy longest-processing-time bin packing: returns the rank of each segment."""
  order = sorted(range(len(costs)), key=lambda i: -costs[i])
  load = [0.0] * world_size
  rank_of = [0] * len(costs)
  for i in order:
    r = min(range(world_size), key=lambda k: load[k])
    rank_of[i] = r
    load[r] += costs[i]
  return rank_of


class SweepBatch:
  """All (or some) bsuite_ids as lane segments on this rank's GPU."""

  def __init__(self, bsuite_ids: Optional[Sequence[str]] = None, total_lanes: int = 1 << 20, *,
               device=None, seed: int = 0, rank: int = 0, world_size: int = 1, num_streams: int = 32,
               env_kwargs: Optional[Dict[str, dict]] = None):
    self.ids = list(_sweep.SWEEP if bsuite_ids is None else bsuite_ids)
    self.table = segment_table(self.ids, total_lanes)
    self.device = torch.device('cuda:0' if device is None else device)
    env_kwargs = env_kwargs or {}
    probe = []
    for bid, _, lanes in self.table:
      name = bid.split(_sweep.SEPARATOR)[0]
      kw = dict(env_kwargs.get(name, {}))
      env = bsuite_amd.load_from_id(bid, **kw)      # host-only construction: specs for the cost model
      probe.append(lanes * bytes_per_step(int(np.prod(env.observation_spec().shape))))
    self.rank_of = assign_segments(probe, world_size)
    self.local = [i for i, r in enumerate(self.rank_of) if r == rank]
    self.envs, self.segments = [], []
    # one device-resident call counter for the whole sweep: bumped once per sweep step
    self._step_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
    for i in self.local:
      bid, begin, lanes = self.table[i]
      name = bid.split(_sweep.SEPARATOR)[0]
      kw = dict(env_kwargs.get(name, {}))
      settings = _sweep.SETTINGS[bid]
      if 'seed' in settings and settings['seed'] is None:
        kw['seed'] = seed
      elif 'seed' not in settings:
        kw.setdefault('seed', seed)
      env = bsuite_amd.load_from_id(bid, batch=lanes, device=self.device, lane_offset=begin,
                                    num_buffers=1, shared_step_counter=self._step_counter, **kw)
      self.envs.append(env)
      self.segments.append((bid, begin, lanes))
    self.num_streams = max(1, int(num_streams))
    self._graph = None
    self._streams = None

  # ---------------------------------------------------------------------------------------
  def random_actions(self, seed: int = 0) -> List[torch.Tensor]:
    """One int32 action tensor per local segment (uniform over each action_spec)."""
    g = torch.Generator(device=self.device)
    g.manual_seed(seed)
    return [torch.randint(env.action_spec().num_values, (lanes,), generator=g, device=self.device,
                          dtype=torch.int32)
            for env, (_, _, lanes) in zip(self.envs, self.segments)]

  def _bump(self):
    from bsuite_amd import _native  # pylint: disable=import-outside-toplevel
    _native.check(_native.lib.bsx_counter_add(self._step_counter.data_ptr(), 1,
                                              torch.cuda.current_stream(self.device).cuda_stream),
                  'sweep step counter')

  def step(self, actions: Sequence[torch.Tensor]):
    """Eager sweep step: one launch (pair) per local segment on the current stream."""
    outs = [env.step(a) for env, a in zip(self.envs, actions)]
    self._bump()
    return outs

  def capture(self, actions: Sequence[torch.Tensor]):
    """Captures one sweep step reading `actions` (static tensors) into a HIP graph."""
    self.step(actions)                         # first call outside capture: allocation, call 0
    torch.cuda.synchronize(self.device)
    main = torch.cuda.Stream(device=self.device)
    self._streams = [torch.cuda.Stream(device=self.device) for _ in range(self.num_streams)]
    main.wait_stream(torch.cuda.current_stream(self.device))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
      with torch.cuda.graph(graph, stream=main):
        for s in self._streams:
          s.wait_stream(main)                  # fork
        outs = []
        for k, (env, a) in enumerate(zip(self.envs, actions)):
          with torch.cuda.stream(self._streams[k % self.num_streams]):
            outs.append(env.step(a))
        for s in self._streams:
          main.wait_stream(s)                  # join
        self._bump()                           # one call-index bump for all segments
    torch.cuda.current_stream(self.device).wait_stream(main)
    self._graph, self._outs = graph, outs
    return outs

  def replay(self):
    """One sweep step from the captured graph (outputs land in the tensors `capture` returned)."""
    self._graph.replay()
    return self._outs

  # ---------------------------------------------------------------------------------------
  def lanes(self) -> int:
    return sum(l for _, _, l in self.segments)

  def summary(self) -> Dict[str, Dict[str, float]]:
    """Per local bsuite_id: lanes, episodes finished/started, sum of every bsuite_info column."""
    from bsuite_amd import distributed as bdist  # pylint: disable=import-outside-toplevel
    out = {}
    for env, (bid, _, _) in zip(self.envs, self.segments):
      vec, names = bdist.local_summary(env)
      out[bid] = dict(zip(names, vec.tolist()))
    return out
