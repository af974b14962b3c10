"""Heterogeneous sweep: many bsuite_ids advanced together on one or several GPUs (BASELINE config 5).

The reference runs the sweep as one OS process per bsuite_id (`bsuite/baselines/utils/pool.py:28-54`
mapping `run(bsuite_id)` over `sweep.SWEEP`).  Here every bsuite_id is a *segment* of lanes with its
own environment family and settings; a "sweep step" advances every segment once.  Segments are
independent, so:

* across GPUs — whole segments are bin-packed onto ranks by `lanes x bytes-per-step` (longest
  processing time first); no communication per step; summaries are all-gathered at the end
  (`bsuite_amd.distributed`);
* on one GPU — **grouped launches** (`prepare_groups` / `step_grouped`): every workgroup finds its
  segment's arguments — configuration, state columns, output buffers, exactly what
  `bsx_<family>_step` takes — in a device-resident table (include/bsuite_amd.h `bsx_group_*`).  By
  default the whole sweep is ONE group (`BSX_FAM_SWEEP_MIXED`) and a sweep step is TWO launches: phase 0
  advances every lane of every family and bumps the shared call counter, phase 1 is the observation
  store stream of deep_sea / catch / mnist.  The finer-grained groups (one per family,
  mixed small families, mixed two-kernel families) and other schedules (`step_grouped_streams`,
  `capture_grouped`) are kept for A/B; the oldest path (`capture` / `replay`) spreads per-segment
  launches over HIP streams and replays them as one HIP graph.  Call indices of the draw stream live
  in one device-resident counter, so steps stay reproducible.

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


# The closed-loop schedule `prepare_groups()` / `step_grouped()` run by default: False = phase 0 | store stream
# (bsx_group_step), True = the split cut of the same two launches (bsx_group_step_split): 2.8-4.5 us per
# sweep step faster in every same-call comparison of round 5 (profiles/r05/ab_sweep_v1.log, ab_sweep_v2.log).
# DEFAULT_ROWS_IN_STREAM: the chains' wide rows written by the group's store stream instead of phase 0 (measured
# 0.8-2 us SLOWER per sweep step in both of its forms, same logs: off).
DEFAULT_SPLIT = True
DEFAULT_ROWS_IN_STREAM = False


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
    self._numel = []
    for bid, _, lanes in self.table:
      name = bid.split(_sweep.SEPARATOR)[0]
      kw = dict(env_kwargs.get(name, {}))
      env = bsuite_amd.load_from_id(bid, **kw)      # host-only construction: specs for the cost model
      self._numel.append(int(np.prod(env.observation_spec().shape)))
      probe.append(lanes * bytes_per_step(self._numel[-1]))
    self.rank_of = assign_segments(probe, world_size)
    self.local = [i for i, r in enumerate(self.rank_of) if r == rank]
    self.envs, self.segments = [], []
    # one device-resident call counter for the whole sweep: bumped once per sweep step
    self._step_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
    # One arena for every segment's observation buffer, in segment order, each on a 4 KiB boundary: the
    # grouped store-stream kernels then walk one ascending address range in whole 4 KiB runs, like the
    # single-environment kernels do, instead of several hundred separately placed allocations.
    self._arena = None
    self._arena_used = 0
    self._arena_floats = sum((self.table[i][2] * p_numel + 1023) // 1024 * 1024
                             for i, p_numel in ((i, self._numel[i]) for i in self.local))
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
                                    num_buffers=1, shared_step_counter=self._step_counter,
                                    obs_allocator=self._alloc_obs, **kw)
      self.envs.append(env)
      self.segments.append((bid, begin, lanes))
    self.num_streams = max(1, int(num_streams))
    self._graph = None
    self._streams = None
    self._groups = []
    self._groups_by_cost = []
    self._grouped_graph = None
    self._pipe = None                          # step_grouped_streams(): the two HIP streams + their events
    self._small = None
    self._pending_steps = 0

  def _alloc_obs(self, shape):
    n = int(np.prod(shape))
    if self._arena is None:
      self._arena = torch.empty(self._arena_floats + 1024, dtype=torch.float32, device=self.device)
      self._arena_used = (-self._arena.data_ptr() // 4) % 1024            # first 4 KiB boundary
    begin = self._arena_used
    if begin + n > self._arena.numel():
      return torch.empty(shape, dtype=torch.float32, device=self.device)  # (never in practice) plain allocation
    self._arena_used = begin + (n + 1023) // 1024 * 1024
    return self._arena[begin:begin + n].view(shape)

  # ---------------------------------------------------------------------------------------
  def random_actions(self, seed: int = 0, ring: int = 0) -> List[torch.Tensor]:
    """One int32 action tensor per local segment (uniform over each action_spec).  The generator is
    re-seeded per segment from (seed, global segment index), so a segment gets the same actions
    whichever rank it was packed onto.  ring = R (a power of two): tensors [R, lanes] — an action ring the
    grouped launches walk on the device, row (sweep step mod R) per step (`prepare_groups`): the batched,
    pre-generated form of bsuite/baselines/random/agent.py:35-37 (actions [T,B])."""
    if ring and (ring < 1 or ring & (ring - 1)):
      raise ValueError('an action ring needs a power of two of rows')
    g = torch.Generator(device=self.device)
    out = []
    for k, (env, (_, _, lanes)) in zip(self.local, zip(self.envs, self.segments)):
      g.manual_seed(int(seed) * 1000003 + k)
      out.append(torch.randint(env.action_spec().num_values, (ring, lanes) if ring else (lanes,), generator=g,
                               device=self.device, dtype=torch.int32))
    return out

  def _bump(self, grouped: bool = False):
    if grouped and getattr(self, '_self_bump', False):
      return                                 # BSX_FAM_SWEEP_MIXED: phase 0 bumps the counter when its last workgroup retires
    from bsuite_amd import _native  # pylint: disable=import-outside-toplevel
    _native.check(_native.lib.bsx_counter_add(self._step_counter.data_ptr(), 1,
                                              torch.cuda.current_stream(self.device).cuda_stream),
                  'sweep step counter')

  def step(self, actions: Sequence[torch.Tensor]):
    """Eager sweep step: one launch (pair) per local segment on the current stream."""
    self.flush_step_indices()
    outs = [env.step(a) for env, a in zip(self.envs, actions)]
    self._bump()
    return outs

  # -- grouped launches --------------------------------------------------------------------
  def prepare_groups(self, actions: Sequence[torch.Tensor], mix_small: bool = True, mix_pairs: bool = True,
                     mix_all: bool = True, pipelined: bool = False, heavy_first: bool = True,
                     rows_in_stream: Optional[bool] = None, split: Optional[bool] = None):
    """Builds the launch groups.  With `mix_all` (default) ONE group for the whole sweep
    (BSX_FAM_SWEEP_MIXED): a sweep step is two launches — phase 0 advances every lane of every family
    and bumps the shared call counter, phase 1 is the observation store stream of the two-kernel
    families — and `step_grouped()` needs nothing else.  Otherwise (A/B, and what the ABI offers
    piecewise): with `mix_pairs` ONE mixed group for the two-kernel families
    deep_sea, catch and mnist together (BSX_FAM_PAIR_MIXED: one advance launch + one observation-stream
    launch for all their segments), else one group per family; with `mix_small` ONE mixed group for
    all small-observation families together (BSX_FAM_SMALL_MIXED), else one per family.  Records every local segment with its static `actions` tensor and uploads the
    argument tables.  Returns the per-segment output TimeSteps (tensors that every `step_grouped()`
    overwrites).

    `rows_in_stream` (with `mix_all`; default DEFAULT_ROWS_IN_STREAM = False): the wide rows of memory_chain /
    umbrella_chain segments are left as flat bit planes in a scratch by phase 0 and written by the phase-1 store stream
    (bsx_call_t.row_scratch) instead of being built and written by phase 0 itself.

    `split` (with `mix_all`, closed-loop): `step_grouped()` cuts the two launches differently (bsx_group_step_split) —
    launch 1 = only the phase-0 workgroups the store stream depends on (lane advance of deep_sea / mnist / large catch
    boards, packed rows of the chains), launch 2 = the store stream BESIDE the whole step of every other
    small-observation segment.  Same results, same closed-loop contract (everything in step s reads the actions of
    step s only; its TimeSteps are complete when launch 2 ends).

    `pipelined` (with `mix_all`): the sweep's actions are static, so sweep step s+1 does not need the
    observations of step s — ONE launch per step then carries the observation store stream of step s
    beside the lane advance of step s+1 (bsx_group_step_pipelined).  Two whole-sweep groups over the same
    segments alternate: their two-kernel segments have the state columns swapped (the stream of step s
    keeps reading the column the advance of step s+1 does not write) and each has its own reward /
    discount / step_type buffers (and observation buffers of the small families, which the advance
    writes).  `step_grouped()` then returns the TimeSteps of the step whose stream it launched — the lanes
    (state, info, episode counters) are one advance ahead of them; `release_groups()` leaves the state in
    the environments' own columns."""
    import ctypes  # pylint: disable=import-outside-toplevel
    from bsuite_amd import _native  # pylint: disable=import-outside-toplevel
    from bsuite_amd import dm_env_compat as dm_env  # pylint: disable=import-outside-toplevel
    self.release_groups()
    rows_in_stream = DEFAULT_ROWS_IN_STREAM if rows_in_stream is None else bool(rows_in_stream)
    torch.cuda.set_device(self.device)       # the argument tables are allocated on the current device
    buckets = {}
    for k, env in enumerate(self.envs):
      raw = env.raw_env if hasattr(env, 'raw_env') else env
      raw._ensure_allocated()  # pylint: disable=protected-access
      numel = int(np.prod(raw.observation_spec().shape))
      small = raw._abi_name not in ('deep_sea', 'catch', 'mnist')  # pylint: disable=protected-access
      klass = _native.lib.bsx_group_small_class(numel) if small else 0
      group_family = ('sweep_mixed' if mix_all else 'small_mixed' if (small and mix_small) else
                      'pair_mixed' if (not small and mix_pairs) else raw._abi_name)  # pylint: disable=protected-access
      if mix_all:
        klass = 0
      buckets.setdefault((group_family, klass), []).append(k)
    if pipelined and not mix_all:
      raise ValueError('pipelined sweep steps need the whole-sweep group (mix_all)')
    outs = [None] * len(self.envs)
    outs_of = [outs, [None] * len(self.envs)]
    self._state_alt = {}
    self._row_scratch = {}
    costs = []
    for (name, _), members in sorted(buckets.items()):
      if name == 'sweep_mixed' and heavy_first:
        # Phase 0 ends when its slowest workgroup retires: put the long-running ones — wide observation rows
        # (umbrella_distract draws and writes up to 103 floats per lane) — at the front of the grid and the
        # cheap lane-advance workgroups of the two-kernel families at its end.
        def weight(k):
          raw_k = self.envs[k].raw_env if hasattr(self.envs[k], 'raw_env') else self.envs[k]
          numel_k = int(np.prod(raw_k.observation_spec().shape))
          # (small catch boards are written by phase 0 itself, 200 bytes per lane: they belong with the heavy ones)
          small_k = (raw_k._abi_name not in ('deep_sea', 'catch', 'mnist') or  # pylint: disable=protected-access
                     (raw_k._abi_name == 'catch' and numel_k <= _native.FUSED_CATCH_MAX_CELLS))  # pylint: disable=protected-access
          # (segments with a share of the phase-1 store stream — the two-kernel families, and the chains' wide rows when the
          # stream writes them — are the tail of the group: what bsx_group_step_split launches first, on its own)
          if small_k and rows_in_stream and raw_k._row_scratch_words():  # pylint: disable=protected-access
            return 1
          return -numel_k if small_k else 2
        members = sorted(members, key=weight)
      def build(parity):
        handle = ctypes.c_void_p()
        _native.check(_native.lib.bsx_group_create(_native.FAMILY_IDS[name], len(members), ctypes.byref(handle)),
                      'bsx_group_create')
        self._groups.append(handle)
        for idx, k in enumerate(members):
          raw = self.envs[k].raw_env if hasattr(self.envs[k], 'raw_env') else self.envs[k]
          extra = {}
          o = raw._out[0]  # pylint: disable=protected-access
          if pipelined:
            # (small catch boards are written by phase 0 itself — fused tiles, csrc/catch.hip — so in a pipelined pair
            # of groups they behave like a small-observation family: one state column, an observation buffer per group)
            pair = (raw._abi_name in ('deep_sea', 'mnist') or  # pylint: disable=protected-access
                    (raw._abi_name == 'catch' and int(np.prod(raw.observation_spec().shape)) > _native.FUSED_CATCH_MAX_CELLS))  # pylint: disable=protected-access
            if pair:
              if k not in self._state_alt:
                self._state_alt[k] = raw._state['state'].clone()  # pylint: disable=protected-access
              extra = dict(state_alt=self._state_alt[k], swap_state=(parity == 1))
          if name == 'sweep_mixed' and rows_in_stream:
            # memory_chain / umbrella_chain with a wide row: phase 0 leaves the row packed in a scratch and the group's
            # store stream writes the observation (csrc/row_stream.h); the groups of a pipelined pair bring one each
            key = (k, parity)
            if key not in self._row_scratch:
              rs = raw._row_scratch(fresh=(parity == 1))  # pylint: disable=protected-access
              self._row_scratch[key] = rs
            if self._row_scratch[key] is not None:
              extra['row_scratch'] = self._row_scratch[key]
          elif name == 'sweep_mixed':
            extra['row_scratch'] = False             # (A/B: not even the scratch a large segment steps with on its own)
          if pipelined:
            if parity == 1:
              o = dict(reward=torch.empty_like(o['reward']), discount=torch.empty_like(o['discount']),
                       step_type=torch.empty_like(o['step_type']),
                       observation=o['observation'] if pair else torch.empty_like(o['observation']))
              extra['out'] = o
          _native.check(raw._group_set(handle, idx, actions[k], **extra), f'bsx_group_set_{raw._abi_name}')  # pylint: disable=protected-access
          raw._grouped_by = self  # pylint: disable=protected-access  (enable_logging refuses until release_groups())
          outs_of[parity][k] = dm_env.TimeStep(step_type=o['step_type'], reward=o['reward'], discount=o['discount'],
                                               observation=o['observation'])
        _native.check(_native.lib.bsx_group_commit(handle), 'bsx_group_commit')
      build(0)
      if pipelined:
        build(1)
      costs.append(sum(self.segments[k][2] * bytes_per_step(int(np.prod(self.envs[k].observation_spec().shape)))
                       for k in members))
    order = sorted(range(len(costs)), key=lambda j: -costs[j])
    self._groups_by_cost = [self._groups[j] for j in order]     # heaviest store streams first
    self._self_bump = mix_all                 # a whole-sweep group moves the call counter on by itself
    self._group_actions = list(actions)      # keep the static action tensors alive
    self._group_outs = outs
    self._pipelined = bool(pipelined)
    # (the split cut needs the segments with a share of the store stream at the tail of the group: the heavy_first order)
    self._split = bool(DEFAULT_SPLIT if split is None else split) and mix_all and not pipelined and heavy_first
    self._pipelined_outs = outs_of
    self._pipelined_step = 0
    return outs_of if pipelined else outs

  def step_grouped(self):
    """One sweep step = one grouped launch per (family, class) + one call-counter bump."""
    from bsuite_amd import _native  # pylint: disable=import-outside-toplevel
    stream = torch.cuda.current_stream(self.device).cuda_stream
    if getattr(self, '_pipelined', False):
      # one launch: observation stream of sweep step s (group s & 1) | lane advance of step s + 1 (the other)
      s = self._pipelined_step
      if s == 0:                               # prologue: the lane advance of step 0 on its own
        _native.check(_native.lib.bsx_group_step_phase(self._groups[0], 0, stream), 'bsx_group_step_phase')
        self._pending_steps += 1
      rc = _native.lib.bsx_group_step_pipelined(self._groups[s & 1], self._groups[(s + 1) & 1], stream)
      if rc != 0:
        _native.check(rc, 'bsx_group_step_pipelined')
      self._pipelined_step = s + 1
      self._pending_steps += 1
      return self._pipelined_outs[s & 1]
    step = _native.lib.bsx_group_step_split if getattr(self, '_split', False) else _native.lib.bsx_group_step
    for handle in self._groups:
      rc = step(handle, stream)
      if rc == _native.BSX_EMODE and step is _native.lib.bsx_group_step_split:
        # the library's commit found the segments with a share of the store stream NOT to be the tail of the group
        # (its blocks2 > 0 rule and weight() here disagree on some segment): nothing was launched — take the ordinary cut
        # of the same two launches, from now on
        self._split = False
        step = _native.lib.bsx_group_step
        rc = step(handle, stream)
      if rc != 0:
        _native.check(rc, 'bsx_group_step')
    self._bump(grouped=True)
    self._pending_steps += 1                   # host-side call index: settled lazily (flush_step_indices)
    return self._group_outs

  # -- eager two-stream schedule ----------------------------------------------------------
  def step_grouped_streams(self, small_beside: str = 'advance'):
    """One sweep step, eager, on HIP streams owned by the batch (no graph).  Two phases, separated by what
    bounds the kernels:
      1. everything latency-bound at once — the advance kernel of the two-kernel families on the `pipe`
         stream, each small-observation group on a stream of its own;
      2. the observation store stream (HBM-bound, ~96 % of the sweep's bytes) alone on the machine.
    The shared call counter is bumped as soon as phase 1 — its only readers — is done, i.e. beside the
    store stream.  (Schedules that let the small groups run beside the store stream measured slower and
    bimodal: both kernels stretch (profiles/r02/ab_sweep_pair_mixed.log); a captured HIP
    graph starts dependent nodes 6-14 us apart and consecutive replays ~20 us apart,
    profiles/r02/sweep_graph_timeline_*.txt.)  Call `join_streams()` before reading the outputs on
    the current stream."""
    if getattr(self, '_pipelined', False):
      raise RuntimeError('pipelined groups alternate inside step_grouped(); prepare_groups(pipelined=False) for this schedule')
    from bsuite_amd import _native  # pylint: disable=import-outside-toplevel
    if self._pipe is None:
      cur = torch.cuda.current_stream(self.device)
      self._pairs = [h for h in self._groups_by_cost if _native.lib.bsx_group_phases(h) == 2]
      self._singles = [h for h in self._groups_by_cost if _native.lib.bsx_group_phases(h) == 1]
      self._pipe = torch.cuda.Stream(device=self.device)
      self._smalls = [torch.cuda.Stream(device=self.device) for _ in self._singles] or [torch.cuda.Stream(device=self.device)]
      self._small = self._smalls[0]
      self._ev_adv, self._ev_bump, self._ev_stream = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
      self._ev_small = [torch.cuda.Event() for _ in self._smalls]
      self._small_beside_stream = small_beside == 'stream'      # A/B: small groups beside the store stream (slower)
      for st in [self._pipe] + self._smalls:
        st.wait_stream(cur)
      first = True
    else:
      first = False
    step_phase = _native.lib.bsx_group_step_phase
    pipe = self._pipe.cuda_stream
    if not first:
      self._pipe.wait_event(self._ev_bump)     # this step's kernels read the counter the last step bumped
    for handle in self._pairs:
      rc = step_phase(handle, 0, pipe)
      if rc != 0:
        _native.check(rc, 'bsx_group_step_phase')
    self._ev_adv.record(self._pipe)
    for j, handle in enumerate(self._singles):
      st = self._smalls[j]
      if not first:
        st.wait_event(self._ev_bump)
        if not self._small_beside_stream:
          st.wait_event(self._ev_stream)       # not beside the previous step's store stream
      rc = step_phase(handle, 0, st.cuda_stream)
      if rc != 0:
        _native.check(rc, 'bsx_group_step_phase')
      self._ev_small[j].record(st)
    if not self._small_beside_stream:
      for j in range(len(self._singles)):
        self._pipe.wait_event(self._ev_small[j])
    for handle in self._pairs:
      rc = step_phase(handle, 1, pipe)
      if rc != 0:
        _native.check(rc, 'bsx_group_step_phase')
    self._ev_stream.record(self._pipe)
    # the bump: after the advance kernel and every small group of this step
    self._small.wait_event(self._ev_adv)
    for j in range(1, len(self._singles)):
      self._small.wait_event(self._ev_small[j])
    if not self._self_bump:
      _native.check(_native.lib.bsx_counter_add(self._step_counter.data_ptr(), 1, self._small.cuda_stream), 'sweep step counter')
    self._ev_bump.record(self._small)
    self._pending_steps += 1                   # host-side call index: settled lazily (flush_step_indices)
    return self._group_outs

  def flush_step_indices(self):
    """Settles the environments' host-side call index (`env.step_index`) after grouped steps: a grouped
    step costs the host a handful of launches, not a loop over several hundred environment objects;
    the authoritative index lives in the shared device counter anyway."""
    n, self._pending_steps = self._pending_steps, 0
    if n:
      for env in self.envs:
        raw = env.raw_env if hasattr(env, 'raw_env') else env
        raw._step_index += n  # pylint: disable=protected-access

  def sync(self):
    """join_streams() + flush_step_indices() + a device synchronisation: call before inspecting the
    environments after grouped steps."""
    self.join_streams()
    self.flush_step_indices()
    torch.cuda.synchronize(self.device)

  def join_streams(self):
    """Makes the current stream wait for everything `step_grouped_streams()` has issued."""
    if self._pipe is not None:
      cur = torch.cuda.current_stream(self.device)
      cur.wait_stream(self._pipe)
      for st in self._smalls:
        cur.wait_stream(st)

  def capture_grouped(self, num_streams: int = 2, phased: bool = True):
    """Captures one grouped sweep step as a HIP graph.

    phased=True (default): the step is split by what bounds each kernel (bsx_group_step_phase).  The
    main branch runs the lane-advance kernel(s) of the two-kernel families; every observation stream
    kernel (HBM-bound: deep_sea, mnist, catch carry ~850 of the sweep's 886 MB) starts on a side branch
    as soon as ITS advance kernel is done; the small-observation groups (latency-bound, little data) run
    on another side branch from the start; the shared call counter — read by the advance and small
    kernels only — is bumped as soon as those are done, beside the store streams.  With the default
    mixed groups that is: advance -> stream on the critical path and nothing else.
    phased=False: whole groups as `num_streams` round-robin branches (the r01 topology).
    Groups are independent (disjoint segments); the shared call counter is bumped after the join.
    Call prepare_groups() first; then `replay_grouped()` per sweep step."""
    if getattr(self, '_pipelined', False):
      raise RuntimeError('pipelined groups alternate inside step_grouped(); prepare_groups(pipelined=False) for this schedule')
    from bsuite_amd import _native  # pylint: disable=import-outside-toplevel
    if not self._groups:
      raise RuntimeError('capture_grouped() needs prepare_groups() first')
    self.step_grouped()                        # one eager step: first-use work stays out of the capture
    torch.cuda.synchronize(self.device)
    main = torch.cuda.Stream(device=self.device)
    n_side = max(2, int(num_streams)) if phased else max(1, int(num_streams)) - 1
    side = [torch.cuda.Stream(device=self.device) for _ in range(n_side)]
    main.wait_stream(torch.cuda.current_stream(self.device))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
      with torch.cuda.graph(graph, stream=main):
        for st in side:
          st.wait_stream(main)                 # fork
        bumped = False
        if phased:
          pairs = [h for h in self._groups_by_cost if _native.lib.bsx_group_phases(h) == 2]
          singles = [h for h in self._groups_by_cost if _native.lib.bsx_group_phases(h) == 1]
          # `small` runs the small-observation groups from t = 0; main runs the advance kernels; every
          # stream kernel goes to a side branch as soon as ITS advance kernel is done.
          small = side[-1] if (len(side) > 1 and singles) else main
          streams = side[:-1] if small is not main else side
          for handle in singles:
            _native.check(_native.lib.bsx_group_step_phase(handle, 0, small.cuda_stream), 'bsx_group_step_phase')
          for j, handle in enumerate(pairs):   # heaviest store stream first
            _native.check(_native.lib.bsx_group_step_phase(handle, 0, main.cuda_stream), 'bsx_group_step_phase')
            ev = torch.cuda.Event()
            ev.record(main)
            st = streams[j % len(streams)]
            st.wait_event(ev)
            _native.check(_native.lib.bsx_group_step_phase(handle, 1, st.cuda_stream), 'bsx_group_step_phase')
          # The call counter is read by the advance and small kernels only (stream kernels decode
          # states): bump it as soon as THOSE are done, beside the store streams, not after them.
          if small is not main:
            main.wait_stream(small)
          self._bump(grouped=True)
          bumped = True
        else:
          lanes = [main] + side
          for j, handle in enumerate(self._groups_by_cost):
            st = lanes[j % len(lanes)]
            _native.check(_native.lib.bsx_group_step(handle, st.cuda_stream), 'bsx_group_step')
        for st in side:
          main.wait_stream(st)                 # join
        if not bumped:
          self._bump(grouped=True)
    torch.cuda.current_stream(self.device).wait_stream(main)
    self._grouped_graph = graph
    self._grouped_streams = [main] + side
    return self._group_outs

  def replay_grouped(self):
    """One sweep step from the graph captured by capture_grouped()."""
    self._grouped_graph.replay()
    self._pending_steps += 1
    return self._group_outs

  def release_groups(self):
    from bsuite_amd import _native  # pylint: disable=import-outside-toplevel
    self._grouped_graph = None
    if hasattr(self, '_pending_steps'):
      self.flush_step_indices()
    if getattr(self, '_pipe', None) is not None:
      self.join_streams()
      torch.cuda.synchronize(self.device)
      self._pipe = self._small = None
    if getattr(self, '_pipelined', False) and self._groups:
      # advances of odd steps wrote the alternate state columns: hand the lanes back in the environments' own
      if self._pipelined_step % 2 == 1:
        for k, alt in self._state_alt.items():
          raw = self.envs[k].raw_env if hasattr(self.envs[k], 'raw_env') else self.envs[k]
          raw._state['state'].copy_(alt)  # pylint: disable=protected-access
      torch.cuda.synchronize(self.device)
      self._pipelined, self._state_alt = False, {}
    for handle in self._groups:
      _native.lib.bsx_group_destroy(handle)
    if self._groups:
      for e in self.envs:
        raw = e.raw_env if hasattr(e, 'raw_env') else e
        raw._grouped_by = None  # pylint: disable=protected-access
    self._groups = []
    self._groups_by_cost = []

  def __del__(self):
    try:
      self.release_groups()
    except Exception:  # pylint: disable=broad-except
      pass

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
    self.join_streams()
    self.flush_step_indices()
    out = {}
    for env, (bid, _, _) in zip(self.envs, self.segments):
      vec, names = bdist.local_summary(env)
      out[bid] = dict(zip(names, vec.tolist()))
    return out
