"""Batched base class for bsuite environments on MI355X.

Mirrors bsuite/environments/base.py:34-77 (`Environment`: `reset()`, `step(action)`, auto-reset
after a LAST step, `bsuite_info()`, `bsuite_num_episodes`) for a *batch* of independent environment
instances ("lanes") that live on the GPU as struct-of-arrays columns and are advanced by one HIP
kernel launch per call through the C ABI (include/bsuite_amd.h).

Two views of the same engine:

* ``batch=None`` (default) — the reference's scalar protocol: ``step(int)`` returns a genuine
  ``dm_env.TimeStep`` with numpy observation and ``reward is None`` on FIRST, so an object from
  ``bsuite_amd.load_from_id(id)`` drops into code written against ``bsuite.load_from_id(id)``.
* ``batch=B`` — the vectorised protocol: ``step(actions)`` takes an int32 device tensor ``[B]`` and
  returns a ``TimeStep`` whose fields are device tensors (``step_type`` int8 ``[B]``, ``reward`` /
  ``discount`` f32 ``[B]``, ``observation`` f32 ``[B, *obs_shape]``).  FIRST lanes carry reward 0 and
  discount 1.  Nothing synchronises with the host.

Ownership: the engine owns state and output tensors.  Outputs rotate through ``num_buffers``
(default 2) buffers, so the TimeStep returned by call k stays valid until call k+num_buffers — enough
for the reference's run loop, which holds `timestep` and `new_timestep`
(bsuite/baselines/experiment.py:43-57).

There is no CPU fallback: without a HIP device the first reset()/step() raises.
"""
import contextlib
import secrets
from typing import Any, Dict, Optional

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd import dm_env_compat as dm_env
from bsuite_amd.dm_env_compat import specs

_MASK63 = (1 << 63) - 1


# torch.cuda.current_stream(dev).cuda_stream / torch.cuda.current_device() through their Python wrappers cost
# 1.9 us / 0.4 us per call — a fifth of a step() of the tiny families; the C getters behind them ~0.1 us.
_current_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)   # pylint: disable=protected-access
if _current_raw_stream is None:
  def _current_raw_stream(index):
    return torch.cuda.current_stream(index).cuda_stream
_current_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device   # pylint: disable=protected-access


def _resolve_seed(seed: Optional[int]) -> int:
  """seed=None means fresh OS entropy, as np.random.RandomState(None) does in the reference."""
  if seed is None:
    return secrets.randbits(63)
  return int(seed) & _MASK63


class Environment(dm_env.EnvironmentBase):
  """A batch of bsuite environments advanced on the GPU.  Subclasses provide one family."""

  # Number of episodes that this environment should be run for (base.py:49).
  bsuite_num_episodes: int

  # Subclass constants.
  _supports_delta = False  # families whose observation is a board with <= 2 hot cells
  _pipelined_rollout = False  # two-kernel families whose rollouts are software-pipelined (state_alt)
  _state_alt = None
  scalar_host_buffers = True   # scalar view: TimeStep / action buffers in pinned host memory mapped into the device (class
                               # attribute: set False before the first step to A/B against device buffers + read-backs)
  # memory_chain / umbrella_chain with a row of more than 8 floats: from this many bytes of observations per step() a call
  # brings a row scratch (bsx_call_t.row_scratch) and is lane advance + wide-row store stream (csrc/row_stream.h) instead
  # of the one launch that builds the rows as bit planes in LDS.  None = never, the default: measured in both of its
  # forms in round 5, the pair path loses to the one launch at every size (2^20 lanes: umbrella_length 35.7 vs 31.0 us,
  # umbrella_distract 100-106 vs 97.3, memory_size 50.8 vs 44.9; profiles/r05/ab_wide_rows_v2_flat_planes.log) — its
  # lane advance alone takes 18 us (4096 workgroups of Philox + an f64 division, 3.5 waves per SIMD), as long as the
  # one launch needs for advance AND stores once other workgroups' stores hide behind it.  Kept as an option (tests,
  # A/B: bench.py --row-path on).
  row_path_min_bytes = None
  _rows = None
  # Families with a single-launch step (deep_sea): the bit of the packed state word that carries the parity of the
  # call index that reads the word next, and all the bits of the word that are the library's bookkeeping (never part of a
  # state_dict).  The flag BSX_CALL_STATE_TAGGED is only set where every call index is exactly the previous one plus 1 —
  # a repeated index finds the words already carrying the next tag.  That holds for an environment that owns its counter
  # (the host count, or a device counter bumped after every call, captured or not) and does not hold
  #  * for a segment of a shared counter (SweepBatch bumps once per sweep step; nothing stops a caller from stepping one
  #    segment twice in between): never tagged;
  #  * while a HIP graph is being captured with the HOST count (every replay repeats the captured index): the captured
  #    call is the two-launch step, whose advance keeps the tags valid for the eager calls that follow.
  _state_tag_bit = None
  _state_lib_bits = 0
  _tag_calls = False
  _info_keys = ()          # names of the f64 info columns, in native column order
  _info_int_keys = ()      # keys the reference reports as Python ints

  def __init__(self, obs_shape, num_actions, *, seed=None, batch=None, device=None,
               lane_offset=0, num_buffers=2, device_step_counter=False, shared_step_counter=None,
               rng='philox', observation_mode='dense', obs_allocator=None):
    self._scalar = batch is None
    self._scalar_last_type = None    # scalar view: step_type of the previous TimeStep (DiscountingChain._check_scalar_action)
    self._batch = 1 if batch is None else int(batch)
    if self._batch < 1:
      raise ValueError('batch must be >= 1')
    self._device = torch.device('cuda:0' if device is None else device)
    if self._device.type == 'cuda' and self._device.index is None:
      self._device = torch.device('cuda', torch.cuda.current_device() if torch.cuda.is_available() else 0)
    self._lane_offset = int(lane_offset)
    self._seed = _resolve_seed(seed if seed is None or isinstance(seed, (int, np.integer)) else 0)
    self._obs_shape = tuple(int(d) for d in obs_shape)
    self._num_actions = int(num_actions)
    self._num_buffers = max(1, int(num_buffers))
    # device_step_counter=True keeps the draw-stream call index in device memory and bumps it
    # with a one-thread kernel after every call, so step() carries no host-side state and can be
    # captured into (and replayed from) a HIP graph.
    self._device_step_counter = bool(device_step_counter) or shared_step_counter is not None
    # shared_step_counter: an int64[1] device tensor owned by the caller (e.g. SweepBatch) who bumps
    # it once per sweep step for all its segments instead of one bump kernel per environment.
    self._shared_step_counter = shared_step_counter
    # obs_allocator(shape) -> float32 device tensor: lets a caller that owns many environments (SweepBatch)
    # place all their observation buffers in one arena, in launch order, each on a 4 KiB boundary.
    self._obs_allocator = obs_allocator
    # rng='mt19937': every lane carries the reference's own generator (np.random.RandomState(seed),
    # MT19937 + numpy's legacy samplers) in HBM, so seeded runs reproduce the reference without any
    # replay shim.  `seed` may be a sequence of B seeds; an int s seeds lane i with
    # s + i.  2.5 KB of state per lane: meant for small batches.
    # observation_mode='delta' (deep_sea, catch): the engine keeps its observation buffers persistent
    # and per call only clears the cells that went stale and sets the new hot cells (a few 4-byte
    # stores per lane instead of the whole board).  The tensors returned are identical to the dense
    # mode's; the caller must treat them as read-only.  Its throughput is reported separately from
    # the dense contract (bench.py --observation-mode delta).
    if observation_mode not in ('dense', 'delta'):
      raise ValueError("observation_mode must be 'dense' or 'delta'")
    if observation_mode == 'delta' and not self._supports_delta:
      raise ValueError(f'{type(self).__name__} has no delta observation mode (its observations are small and dense)')
    self._delta = observation_mode == 'delta'
    if rng not in ('philox', 'mt19937'):
      raise ValueError("rng must be 'philox' or 'mt19937'")
    self._rng_mode = rng
    self._mt_seeds = None
    if rng == 'mt19937':
      if seed is None or isinstance(seed, (int, np.integer)):
        self._mt_seeds = [(self._seed + i) & 0xFFFFFFFF for i in range(self._batch)]
      else:
        self._mt_seeds = [int(x) for x in seed]
        if len(self._mt_seeds) != self._batch:
          raise ValueError('need one seed per lane')
    self._wrap = (_native.WRAP_NONE, 0.0, 0, 0.0)   # fused reward epilogue: kind, param, wrapper seed, param2
    self._wrap_mt_seeds = None         # rng='mt19937' + RewardNoise: the wrapper's own RandomState seeds
    self._wrap_mt = None
    self._logging = None
    self._deferred_steps = None
    self._step_index = 0
    self._buf = 0
    self._allocated = False
    self._reset_next_step = True       # base.py:52 (every lane starts with its reset flag set)

  # ----------------------------------------------------------------------------------------
  # device buffers
  @property
  def batch_size(self) -> int:
    return self._batch

  @property
  def device(self) -> torch.device:
    return self._device

  @property
  def seed(self) -> int:
    return self._seed

  @property
  def lane_offset(self) -> int:
    return self._lane_offset

  @property
  def step_index(self) -> int:
    """Index the next reset()/step() call will use in the draw stream (host-side count; with
    device_step_counter=True and graph replays, `device_step_index()` is authoritative)."""
    return self._step_index

  def device_step_index(self) -> int:
    self._ensure_allocated()
    return int(self._step_base.item()) if self._device_step_counter else self._step_index

  @contextlib.contextmanager
  def step_counter_deferred(self):
    """For HIP-graph capture of K consecutive step() calls (needs device_step_counter=True): inside
    the block the calls take their call index as `device counter + position`, and the counter is
    bumped ONCE, by K, when the block ends — one small kernel per K steps instead of one per step
    (a captured graph otherwise carries 2K nodes for K steps)."""
    if not self._device_step_counter or self._shared_step_counter is not None:
      raise ValueError('step_counter_deferred() needs an environment built with device_step_counter=True')
    self._ensure_allocated()
    self._deferred_steps = 0
    try:
      yield self
    finally:
      n, self._deferred_steps = self._deferred_steps, None
      if n:
        with torch.cuda.device(self._device):
          _native.check(_native.lib.bsx_counter_add(self._step_base.data_ptr(), n,
                                                    torch.cuda.current_stream(self._device).cuda_stream),
                        'step counter bump')

  def _state_tensors(self) -> Dict[str, torch.Tensor]:
    """Subclass hook: allocate the family's SoA state columns with their initial values."""
    raise NotImplementedError

  _abi_name = None   # subclass: family name in the C ABI (bsx_<name>_step / bsx_group_set_<name>)

  def _native_args(self, call, action_ptr, out):
    """Subclass hook: the argument tuple of bsx_<family>_step for this environment."""
    raise NotImplementedError

  def _launch(self, call, action_ptr, out) -> int:
    """Calls the family's C-ABI entry point."""
    return getattr(_native.lib, f'bsx_{self._abi_name}_step')(*self._native_args(call, action_ptr, out))

  def _group_set(self, group, index: int, action: torch.Tensor, *, out=None, state_alt=None,
                 swap_state: bool = False, row_scratch=None) -> int:
    """Records this environment as segment `index` of a grouped launch (bsx_group_set_<family>):
    static arguments — `action` is read in place every group step, outputs go to buffer 0 (or to `out`, a
    dict of reward / discount / step_type / observation tensors), the call index comes from the (shared)
    device step counter.  `action` is int32 [B], or an action RING [R, B] with R a power of two: group step s
    then reads row s mod R (bsx_call_t.action_ring) — pre-generated random actions that change every step.  `row_scratch`
    (memory_chain / umbrella_chain with wide rows, whole-sweep groups): the segment leaves its rows packed there and the
    group's store stream decodes them.  `state_alt` (two-kernel families, pipelined sweeps): the lane advance reads that
    column and writes the environment's own; with `swap_state` the roles are exchanged."""
    self._ensure_allocated()
    ring = int(action.shape[0]) if (torch.is_tensor(action) and action.dim() == 2) else 0
    if (not torch.is_tensor(action) or action.dtype != torch.int32 or action.device != self._device
        or tuple(action.shape) not in ((self._batch,), (ring, self._batch)) or not action.is_contiguous()
        or (action.dim() == 2 and (ring < 1 or ring & (ring - 1)))):
      raise ValueError(f'grouped launches read the action tensor in place every step: need a contiguous int32 '
                       f'tensor of shape ({self._batch},) — or an action ring (R, {self._batch}) with R a power '
                       f'of two — on {self._device}')
    if not self._device_step_counter:
      raise ValueError('grouped launches need device_step_counter / shared_step_counter')
    if self._delta:
      raise ValueError("grouped launches write dense observations; use observation_mode='dense'")
    call = self._call_desc
    call.force_reset, call.n_steps = 0, 0
    kind, param, wseed, param2 = self._wrap
    call.wrap.kind, call.wrap.param, call.wrap.seed, call.wrap.param2 = kind, param, wseed, param2
    self._wrap_applied = self._wrap
    call.stream.step_index = 0
    call.action_ring = ring
    self._buf = 1 % self._num_buffers
    ptrs = self._out_ptrs[0] if out is None else _native.TimeStepPtrs(
        out['reward'].data_ptr(), out['discount'].data_ptr(), out['step_type'].data_ptr(), out['observation'].data_ptr())
    own = self._state.get('state')
    own_rows = call.row_scratch
    if row_scratch is not None:                # whole-sweep groups: the chains' wide rows go through the group's store stream
      call.row_scratch = row_scratch.data_ptr() if row_scratch is not False else None
    try:
      if state_alt is not None:
        if swap_state:
          self._state['state'], call.state_alt = state_alt, own.data_ptr()
        else:
          call.state_alt = state_alt.data_ptr()
      return getattr(_native.lib, f'bsx_group_set_{self._abi_name}')(
          group, index, *self._native_args(call, action.data_ptr(), ptrs))
    finally:
      call.state_alt = None
      call.action_ring = 0
      call.row_scratch = own_rows
      if state_alt is not None:
        self._state['state'] = own

  def _row_scratch_words(self) -> int:
    """uint32 words of this environment's row scratch (bsx_row_scratch_bytes / 4), 0 = the family has no row path."""
    fam = _native.FAMILY_IDS.get(self._abi_name, -1)
    return int(_native.lib.bsx_row_scratch_bytes(fam, int(np.prod(self._obs_shape)), self._batch)) // 4 if fam >= 0 else 0

  def _row_scratch(self, fresh: bool = False) -> Optional[torch.Tensor]:
    """The row scratch of this environment (allocated on first use; contents are irrelevant between calls), or a
    second one (`fresh`: the other group of a pipelined pair must not share it); None for a family without row path."""
    words = self._row_scratch_words()
    if not words:
      return None
    if fresh or self._rows is None:
      with torch.cuda.device(self._device):
        t = torch.empty(words, dtype=torch.int32, device=self._device)
      if fresh:
        return t
      self._rows = t
    return self._rows

  def _set_wrap_mt_seeds(self, seeds):
    """rng='mt19937': RewardNoise's own np.random.RandomState(seed) per lane (wrappers.py:267)."""
    self._wrap_mt_seeds = [int(s) & 0xFFFFFFFF for s in seeds]
    if self._allocated:
      self._upload_wrap_mt()

  @staticmethod
  def _mt_columns(seeds, device, prepare=None):
    """[624,B] key words, [B] pos, [B] gauss, [B] has_gauss of np.random.RandomState(seed) per lane."""
    B = len(seeds)
    keys = np.empty((B, 624), np.uint32)
    pos = np.empty(B, np.int32)
    gauss = np.zeros(B, np.float64)
    has = np.zeros(B, np.int32)
    for i, s_i in enumerate(seeds):
      rs = np.random.RandomState(s_i)
      if prepare is not None:
        prepare(rs)
      _, key, p, hg, cg = rs.get_state()
      keys[i], pos[i], has[i], gauss[i] = key, p, hg, cg
    return dict(state=torch.from_numpy(np.ascontiguousarray(keys.T).view(np.int32)).to(device),
                pos=torch.from_numpy(pos).to(device), gauss=torch.from_numpy(gauss).to(device),
                has_gauss=torch.from_numpy(has).to(device))

  def _upload_wrap_mt(self):
    with torch.cuda.device(self._device):
      self._wrap_mt = self._mt_columns(self._wrap_mt_seeds, self._device)
    w = self._call_desc.wrap
    w.mt_state, w.mt_pos = self._wrap_mt['state'].data_ptr(), self._wrap_mt['pos'].data_ptr()
    w.mt_gauss, w.mt_has_gauss = self._wrap_mt['gauss'].data_ptr(), self._wrap_mt['has_gauss'].data_ptr()

  def _mt_constructor_draws(self, rs: np.random.RandomState):
    """Subclass hook (rng='mt19937'): consume from `rs` exactly what the reference constructor
    draws from `self._rng` before the first reset (most families: nothing)."""

  def _ensure_allocated(self):
    if self._allocated:
      return
    if self._device.type != 'cuda' or not torch.cuda.is_available():
      raise RuntimeError(
          'bsuite_amd runs its environment dynamics in HIP kernels on an MI355X; no HIP device is '
          'visible to torch and there is deliberately no CPU fallback.')
    B, dev = self._batch, self._device
    with torch.cuda.device(dev):
      self._state = self._state_tensors()
      n_info = max(1, len(self._info_keys))
      self._info = torch.zeros((n_info, B), dtype=torch.float64, device=dev)
      self._counters = torch.zeros((_native.COUNTER_SHARDS, _native.COUNTER_STRIDE),
                                   dtype=torch.int64, device=dev)
      self._step_base = (self._shared_step_counter if self._shared_step_counter is not None
                         else torch.zeros(1, dtype=torch.int64, device=dev))
      self._out = []
      self._out_ptrs = []
      # Scalar view: TimeStep buffers and the action live in pinned host memory, which HIP maps into
      # the device address space — the kernels write the TimeStep straight into host RAM and a step
      # costs one stream synchronisation instead of a fill kernel + four device-to-host reads.
      self._host_out = self._scalar and self.scalar_host_buffers
      place = dict(pin_memory=True) if self._host_out else dict(device=dev)
      for _ in range(self._num_buffers):
        o = dict(
            reward=torch.empty(B, dtype=torch.float32, **place),
            discount=torch.empty(B, dtype=torch.float32, **place),
            step_type=torch.empty(B, dtype=torch.int8, **place),
            observation=(self._obs_allocator((B,) + self._obs_shape)
                         if (self._obs_allocator is not None and not self._scalar and not self._delta) else
                         (torch.zeros if self._delta else torch.empty)((B,) + self._obs_shape, dtype=torch.float32, **place)))
        self._out.append(o)
        self._out_ptrs.append(_native.TimeStepPtrs(
            o['reward'].data_ptr(), o['discount'].data_ptr(), o['step_type'].data_ptr(),
            o['observation'].data_ptr()))
      # delta mode: per buffer, the packed state whose hot cells the buffer currently shows (-1: none)
      self._paint = ([torch.full((B,), -1, dtype=torch.int32, device=dev) for _ in range(self._num_buffers)]
                     if self._delta else None)
      self._scalar_action = torch.zeros(1, dtype=torch.int32, **place)
      # scalar view: the reward as the f64 the reference returns (not its f32 rounding)
      self._reward_f64 = torch.zeros(1, dtype=torch.float64, **place) if self._scalar else None
      self._out_np = [{k: v.numpy() for k, v in o.items()} for o in self._out] if self._host_out else None
    mt_state_ptr = mt_pos_ptr = mt_gauss_ptr = mt_has_ptr = None
    if self._rng_mode == 'mt19937':
      with torch.cuda.device(dev):
        # initial states built by numpy itself, after the draws each reference constructor makes
        cols = self._mt_columns(self._mt_seeds, dev, prepare=self._mt_constructor_draws)
      self._mt_state, self._mt_pos = cols['state'], cols['pos']              # [624, B], [B]
      self._mt_gauss, self._mt_has_gauss = cols['gauss'], cols['has_gauss']
      mt_state_ptr, mt_pos_ptr = self._mt_state.data_ptr(), self._mt_pos.data_ptr()
      mt_gauss_ptr, mt_has_ptr = self._mt_gauss.data_ptr(), self._mt_has_gauss.data_ptr()
    # One persistent call descriptor: only step_index / force_reset / stream change per call.
    self._call_desc = _native.Call(
        n_lanes=B, force_reset=0,
        stream=_native.Stream(self._seed, self._lane_offset, 0,
                              self._step_base.data_ptr() if self._device_step_counter else None,
                              mt_state_ptr, mt_pos_ptr, mt_gauss_ptr, mt_has_ptr),
        wrap=_native.RewardWrap(_native.WRAP_NONE, 0, 0.0, 0, None, None, None, None, 0.0),
        counters=self._counters.data_ptr(), hip_stream=None)
    if self._reward_f64 is not None:
      self._call_desc.reward_f64 = self._reward_f64.data_ptr()
    if (not self._scalar and self.row_path_min_bytes is not None and self._row_scratch_words()
        and B * int(np.prod(self._obs_shape)) * 4 >= self.row_path_min_bytes):
      self._call_desc.row_scratch = self._row_scratch().data_ptr()
    if self._wrap_mt_seeds is not None:
      self._upload_wrap_mt()
    # The host side of a step() call is part of the hot path: the tiny families' kernels run 5-7 us at 2^20
    # lanes and a slower host leaves the GPU idle between launches (tools/host_overhead.py).  Everything that
    # does not change between calls is resolved once: the entry point, its argument list per output buffer
    # (state / info pointers never move: load_state_dict copies in place), the nested structs of the call
    # descriptor, and the TimeStep tuples the batched view returns.
    self._fn = getattr(_native.lib, f'bsx_{self._abi_name}_step')
    self._argv = [list(self._native_args(self._call_desc, 0, p)) for p in self._out_ptrs]
    self._call_stream, self._call_wrap = self._call_desc.stream, self._call_desc.wrap     # views of the same memory
    self._wrap_applied = None
    if self._state_tag_bit is not None:
      self._tag_calls = self._shared_step_counter is None
      self._call_desc.flags = _native.CALL_STATE_TAGGED if self._tag_calls else 0
    self._tag_host_count = self._tag_calls and not self._device_step_counter
    self._dev_index = self._device.index
    self._timesteps = None if self._scalar else [
        dm_env.TimeStep(step_type=o['step_type'], reward=o['reward'], discount=o['discount'], observation=o['observation'])
        for o in self._out]
    self._allocated = True

  # ----------------------------------------------------------------------------------------
  # the hot path
  def _call(self, action_ptr: int, force_reset: bool) -> int:
    """One reset()/step() launch; returns the index of the output buffer it wrote."""
    if _current_device() != self._dev_index:
      # kernels launch in the current device's context: step an environment that lives elsewhere
      # (several GPUs driven from one process) under its own device
      with torch.cuda.device(self._device):
        return self._call(action_ptr, force_reset)
    b = self._buf
    self._buf = b + 1 if b + 1 < self._num_buffers else 0
    call = self._call_desc
    if self._delta:
      call.obs_paint = self._paint[b].data_ptr()
    call.force_reset = 1 if force_reset else 0
    if self._tag_host_count:                       # (deep_sea with the host-side call count: not while capturing.  The
      # current device IS this environment's here — the redirect at the top of _call — so this asks about the stream the
      # kernels are about to be launched on)
      call.flags = 0 if torch.cuda.is_current_stream_capturing() else _native.CALL_STATE_TAGGED
    if self._wrap is not self._wrap_applied:       # the wrappers install a NEW tuple when they change it
      w = self._call_wrap
      w.kind, w.param, w.seed, w.param2 = self._wrap
      self._wrap_applied = self._wrap
    hip_stream = _current_raw_stream(self._dev_index)
    call.hip_stream = hip_stream
    argv = self._argv[b]
    argv[2] = action_ptr
    if self._device_step_counter:
      if self._deferred_steps is not None:
        # inside `step_counter_deferred()`: call index = device counter + position in the block
        self._call_stream.step_index = self._deferred_steps
        rc = self._fn(*argv)
        self._call_stream.step_index = 0
        self._deferred_steps += 1
      else:
        rc = self._fn(*argv)
        if rc == 0 and self._shared_step_counter is None:
          rc = _native.lib.bsx_counter_add(self._step_base.data_ptr(), 1, hip_stream)
    else:
      self._call_stream.step_index = self._step_index
      rc = self._fn(*argv)
    if rc != 0:
      _native.check(rc, f'{type(self).__name__} step')
    self._step_index += 1
    return b

  # ----------------------------------------------------------------------------------------
  # batched `Logging` bookkeeping (bsuite/utils/wrappers.py:34-147), fused into the kernels
  def enable_logging(self, log_by_step: bool = False, log_every: bool = False,
                     max_rows: Optional[int] = None, max_count: Optional[int] = None):
    """Turns on per-lane steps/episode/return tracking and log-spaced snapshot rows.

    max_count: largest episode (or step) count to tabulate log points for (default 10^18: every
    count a 63-bit counter can reach — 14 points per decade, a few hundred table entries, cheap).
    max_rows: snapshot rows kept PER LANE — the rows tensor is B x max_rows x (5 + n_info) f64, so this is
    what sizes the HBM footprint.  Default (batched view): the rows a run of 100 x bsuite_num_episodes
    episodes (at least 10^4; by step: 10^3 x that many steps) produces at log points — + 2, twice that with
    log_by_step (a log-point LAST and the FIRST after it share one step count and both log,
    wrappers.py:96-102) — ~70 rows instead of the ~250 of every reachable count (at B = 2^20 that is 4 GB
    instead of 14-30 GB per wrapped environment); 4096 with log_every.  The scalar view hands
    each row to its logger right after the step and reuses the buffer, so it never fills.  Rows past
    max_rows are counted, not stored, and `Logging.rows()` raises: pass max_rows / max_count for longer runs."""
    from bsuite_amd.utils import wrappers as _w  # pylint: disable=import-outside-toplevel
    if getattr(self, '_grouped_by', None) is not None:
      # (prepared groups hold this environment's column pointers — a pipelined pair of groups also a CLONE of the packed
      # state column, pending-miss bits of catch included — and the kernels they launch were chosen without Logging)
      raise RuntimeError('enable_logging() on a segment of prepared sweep groups: SweepBatch.release_groups() first')
    self._ensure_allocated()
    if self._logging is None:
      # Families that fold an info column only at episode ends (cartpole, mountain_car) switch to the
      # reference's per-step accumulation under Logging, whose rows snapshot the columns mid-episode:
      # bring the columns up to date with the running episodes first.
      for j, pending in self._pending_info().items():
        self._info[j] += pending
      self._clear_pending_info()
    explicit_count = max_count is not None
    if max_count is None:
      max_count = 10 ** 18
    points = _w.logarithmic_logging_points(max_count)
    if max_rows is None:
      if self._scalar:
        max_rows = 8
      elif log_every:
        max_rows = 4096
      else:
        horizon = max_count if explicit_count else (
            max(10 ** 4, 100 * int(getattr(self, 'bsuite_num_episodes', 0) or 0)) * (1000 if log_by_step else 1))
        max_rows = (2 if log_by_step else 1) * sum(1 for p_ in points if p_ <= horizon) + 2
    B, dev = self._batch, self._device
    n_info = len(self._info_keys)
    # scalar view: the one lane's counters and rows sit in mapped host memory like its TimeStep, so
    # the Logging wrapper reads them after the step's synchronisation without device-to-host copies
    place = dict(pin_memory=True) if self._host_out else dict(device=dev)
    lg = dict(
        steps=torch.zeros(B, dtype=torch.int64, **place),
        episode=torch.zeros(B, dtype=torch.int64, **place),
        total_return=torch.zeros(B, dtype=torch.float64, **place),
        episode_len=torch.zeros(B, dtype=torch.int64, **place),
        episode_return=torch.zeros(B, dtype=torch.float64, **place),
        rows=torch.zeros((B, max_rows, 5 + n_info), dtype=torch.float64, **place),
        n_rows=torch.zeros(B, dtype=torch.int32, **place),
        log_points=torch.tensor(points, dtype=torch.int64, device=dev))
    self._logging = lg
    self._logging_desc = _native.Logging(
        steps=lg['steps'].data_ptr(), episode=lg['episode'].data_ptr(),
        total_return=lg['total_return'].data_ptr(), episode_len=lg['episode_len'].data_ptr(),
        episode_return=lg['episode_return'].data_ptr(), rows=lg['rows'].data_ptr(),
        n_rows=lg['n_rows'].data_ptr(), info=self._info.data_ptr() if n_info else None,
        log_points=lg['log_points'].data_ptr(), n_log_points=len(points), max_rows=max_rows,
        n_info=n_info, log_by_step=int(bool(log_by_step)), log_every=int(bool(log_every)))
    import ctypes  # pylint: disable=import-outside-toplevel
    self._call_desc.logging = ctypes.pointer(self._logging_desc)
    return lg

  def logging_columns(self):
    """Column names of a snapshot row, in row order (STANDARD_KEYS first, wrappers.py:30-31)."""
    return ('steps', 'episode', 'total_return', 'episode_len', 'episode_return') + tuple(self._info_keys)

  def _coerce_actions(self, action) -> torch.Tensor:
    if self._scalar:
      a = int(action)
      # base.py:59-62: a step that auto-resets (fresh environment, or the previous TimeStep was LAST) never looks at its
      # action — the reference raises nothing there, whatever it is
      if self._scalar_last_type is not None and self._scalar_last_type != _native.LAST:
        self._check_scalar_action(a)
      if self._host_out:
        self._scalar_action.numpy()[0] = a      # host write; the kernel reads it through the mapping
      else:
        self._scalar_action.fill_(a)
      return self._scalar_action
    if not torch.is_tensor(action):
      action = torch.as_tensor(np.asarray(action), device=self._device)
    if action.device != self._device:
      action = action.to(self._device)
    if action.dtype != torch.int32:
      action = action.to(torch.int32)
    if action.shape != (self._batch,):
      raise ValueError(f'expected actions of shape ({self._batch},), got {tuple(action.shape)}')
    return action.contiguous()

  def _check_scalar_action(self, action: int):
    """Scalar view only: subclasses raise what the reference raises for an invalid action."""

  def _wrap_output(self, b: int):
    if not self._scalar:
      return self._timesteps[b]
    out = self._out[b]
    if self._host_out:
      torch.cuda.current_stream(self._device).synchronize()     # the TimeStep is now in host memory
      o = self._out_np[b]
      st, reward, discount = int(o['step_type'][0]), float(self._reward_f64.numpy()[0]), float(o['discount'][0])
      obs = o['observation'][0].copy()                          # fresh array per step, like the reference
    else:
      st = int(out['step_type'].item())
      obs = out['observation'][0].cpu().numpy()
      reward, discount = float(self._reward_f64.item()), float(out['discount'].item())
    self._scalar_last_type = st
    if st == _native.FIRST:
      return dm_env.restart(obs)
    if st == _native.LAST:
      return dm_env.TimeStep(dm_env.StepType.LAST, reward, discount, obs)
    return dm_env.transition(reward=reward, observation=obs, discount=discount)

  def reset(self) -> dm_env.TimeStep:
    """Resets every lane (base.py:54-57) and returns the FIRST TimeStep."""
    self._reset_next_step = False
    if not self._allocated:
      self._ensure_allocated()
    return self._wrap_output(self._call(0, force_reset=True))

  def step(self, action) -> dm_env.TimeStep:
    """Steps every lane; lanes whose previous step was LAST (or that are fresh) reset instead and
    ignore their action (base.py:59-65)."""
    if not self._allocated:
      self._ensure_allocated()
    if (type(action) is torch.Tensor and action.dtype is torch.int32 and action.is_cuda and action.dim() == 1
        and action.size(0) == self._batch and action.is_contiguous() and action.get_device() == self._dev_index
        and not self._scalar):
      ptr = action.data_ptr()                  # the common call: nothing to convert
    else:
      ptr = self._coerce_actions(action).data_ptr()
    return self._wrap_output(self._call(ptr, force_reset=False))

  def rollout(self, actions) -> dm_env.TimeStep:
    """T consecutive step() calls in one entry-point call (batched view only).

    actions: int32 device tensor [T, B].  Returns a TimeStep whose fields carry a leading T
    dimension (step_type/reward/discount [T,B], observation [T,B,*obs_shape]) — exactly what T
    step() calls would have returned, in order.  The small-observation families run the T steps
    inside ONE kernel launch (state stays in L2, HBM sees only actions in / TimeSteps out); mnist issues
    its kernel pair T times; deep_sea and catch are software-pipelined — after the first lane advance every
    launch carries the observation stream of step t beside the advance of step t+1 (T+1 launches).  Output buffers are cached per T and
    overwritten by the next rollout of the same length."""
    if self._scalar:
      raise TypeError('rollout() needs the batched view (batch=B)')
    if torch.cuda.is_available() and torch.cuda.current_device() != self._device.index:
      with torch.cuda.device(self._device):
        return self.rollout(actions)
    if self._delta:
      raise ValueError("rollout() writes T separate observation arrays; use observation_mode='dense'")
    self._ensure_allocated()
    if not torch.is_tensor(actions) or actions.dim() != 2 or actions.shape[1] != self._batch:
      raise ValueError(f'expected actions of shape (T, {self._batch})')
    if actions.device != self._device or actions.dtype != torch.int32:
      actions = actions.to(device=self._device, dtype=torch.int32)
    actions = actions.contiguous()
    T = int(actions.shape[0])
    if T < 1:
      raise ValueError('rollout needs at least one step')
    cache = self.__dict__.setdefault('_rollout_out', {})
    if T not in cache:
      B, dev = self._batch, self._device
      o = dict(reward=torch.empty((T, B), dtype=torch.float32, device=dev),
               discount=torch.empty((T, B), dtype=torch.float32, device=dev),
               step_type=torch.empty((T, B), dtype=torch.int8, device=dev),
               observation=torch.empty((T, B) + self._obs_shape, dtype=torch.float32, device=dev))
      cache[T] = (o, _native.TimeStepPtrs(o['reward'].data_ptr(), o['discount'].data_ptr(),
                                          o['step_type'].data_ptr(), o['observation'].data_ptr()))
    out, ptrs = cache[T]
    call = self._call_desc
    call.force_reset = 0
    call.n_steps = T
    if self._pipelined_rollout and T > 1:
      # deep_sea / catch: a scratch state column lets every launch after the first carry the observation
      # stream of step t beside the lane advance of step t+1 (bsx_call_t.state_alt)
      if self._state_alt is None:
        self._state_alt = torch.empty_like(self._state['state'])
      call.state_alt = self._state_alt.data_ptr()
    kind, param, wseed, param2 = self._wrap
    call.wrap.kind, call.wrap.param, call.wrap.seed, call.wrap.param2 = kind, param, wseed, param2
    self._wrap_applied = self._wrap
    hip_stream = torch.cuda.current_stream(self._device).cuda_stream
    call.hip_stream = hip_stream
    try:
      if self._device_step_counter:
        rc = self._launch(call, actions.data_ptr(), ptrs)
        if rc == 0 and self._shared_step_counter is None:
          rc = _native.lib.bsx_counter_add(self._step_base.data_ptr(), T, hip_stream)
      else:
        call.stream.step_index = self._step_index
        rc = self._launch(call, actions.data_ptr(), ptrs)
    finally:
      call.n_steps = 0
      call.state_alt = None
    if rc != 0:
      _native.check(rc, f'{type(self).__name__} rollout')
    self._step_index += T
    return dm_env.TimeStep(step_type=out['step_type'], reward=out['reward'], discount=out['discount'],
                           observation=out['observation'])

  def _step(self, action):
    raise NotImplementedError('The batched engine fuses _step/_reset into one kernel; call step().')

  def _reset(self):
    raise NotImplementedError('The batched engine fuses _step/_reset into one kernel; call reset().')

  # ----------------------------------------------------------------------------------------
  # specs / metadata
  def observation_spec(self):
    return specs.Array(shape=self._obs_shape, dtype=np.float32, name='observation')

  def action_spec(self):
    return specs.DiscreteArray(self._num_actions, name='action')

  def _pending_info(self) -> Dict[int, torch.Tensor]:
    """Subclass hook: {info column index: f64 [B] tensor} — what the RUNNING episodes have earned
    so far in columns that the kernel folds into `_info` only at episode ends (an exact function of
    the lane's step counter; see cartpole / mountain_car).  Empty for every other family."""
    return {}

  def _clear_pending_info(self):
    """Subclass hook: called once `_pending_info()` has been added to the columns (enable_logging), for families
    that keep the pending amount in state bits of their own (catch) rather than deriving it from a step counter."""

  _info_pending_column = None   # subclass: the state column that carries the not-yet-folded part of an info column
  _info_variant = 0             # subclass: bsx_bsuite_info's `variant` (cartpole: 1 = swing-up)

  def _info_columns(self) -> torch.Tensor:
    """The f64 [K, B] bsuite_info accumulators as the reference would report them right now: the columns themselves,
    or — families that fold part of an accumulator lazily (catch, cartpole, mountain_car; never under Logging) — a
    fresh tensor from bsx_bsuite_info, the C ABI's form of this method."""
    if self._logging is not None or self._info_pending_column is None:
      return self._info
    cols = torch.empty_like(self._info)
    with torch.cuda.device(self._device):
      _native.check(_native.lib.bsx_bsuite_info(
          _native.FAMILY_IDS[self._abi_name], self._info_variant, self._batch,
          self._state[self._info_pending_column].data_ptr(), self._info.data_ptr(), self._info.shape[0], 1,
          cols.data_ptr(), torch.cuda.current_stream(self._device).cuda_stream), 'bsx_bsuite_info')
    return cols

  def bsuite_info(self) -> Dict[str, Any]:
    """Logging metadata (base.py:75-77).  Scalar view: Python numbers as in the reference.
    Batched view: one f64 device tensor [B] per key (the engine's accumulators; for cartpole and
    mountain_car `raw_return` is a fresh tensor that includes the running episodes)."""
    self._ensure_allocated()
    info = self._info_columns()
    if self._scalar:
      vals = info[:, 0].cpu().numpy()
      out = {}
      for j, k in enumerate(self._info_keys):
        if not k.startswith('_'):      # '_x' columns are engine-internal accumulators
          out[k] = int(vals[j]) if k in self._info_int_keys else float(vals[j])
      return out
    return {k: info[j] for j, k in enumerate(self._info_keys) if not k.startswith('_')}

  def episode_counters(self) -> torch.Tensor:
    """int64 [2] device tensor: lanes that emitted LAST, lanes that emitted FIRST (all calls so
    far).  The kernels accumulate wave-ballot popcounts into 256 sharded counters; this sums them."""
    self._ensure_allocated()
    return self._counters[:, :2].sum(dim=0)

  def invalid_action_count(self) -> torch.Tensor:
    """int64 scalar device tensor: lane-steps so far whose action was outside the action_spec in a
    family where the reference raises IndexError (bandit.py:61, catch.py:84; discounting_chain.py:80 for an episode's first
    action outside -5..4 — Python's negative indices are legal there and the kernels wrap them the same way).  The batched
    kernels clamp such actions instead of faulting; assert this is 0 to get the reference's strictness without a per-step
    host check.  The scalar view raises IndexError exactly where the reference does (`_check_scalar_action` of the three
    families; tests/test_gpu_dm_env_conformance.py)."""
    self._ensure_allocated()
    return self._counters[:, 2].sum()

  def state_dict(self) -> Dict[str, Any]:
    """Everything needed to resume this batch bit-exactly (device tensors are cloned)."""
    self._ensure_allocated()
    d = {k: v.clone() for k, v in self._state.items()}
    if self._state_lib_bits:
      d['state'] &= ~self._state_lib_bits          # (a dict is not tied to a call index or to a batch's call schedule)
    d['__info'] = self._info.clone()
    d['__counters'] = self._counters.clone()
    d['__step_index'] = self.device_step_index()
    d['__seed'] = self._seed
    if self._rng_mode == 'mt19937':
      d['__mt_state'], d['__mt_pos'] = self._mt_state.clone(), self._mt_pos.clone()
      d['__mt_gauss'], d['__mt_has_gauss'] = self._mt_gauss.clone(), self._mt_has_gauss.clone()
      if self._wrap_mt is not None:
        for k, v in self._wrap_mt.items():
          d['__wrap_mt_' + k] = v.clone()
    d['__wrap'] = tuple(self._wrap)                       # fused RewardNoise / RewardScale epilogue
    if self._logging is not None:                         # fused Logging bookkeeping (counters + rows)
      for k in ('steps', 'episode', 'total_return', 'episode_len', 'episode_return', 'rows', 'n_rows'):
        d['__logging_' + k] = self._logging[k].clone()
    return d

  def load_state_dict(self, d: Dict[str, Any]):
    self._ensure_allocated()
    self._scalar_last_type = None       # (scalar view: no action check until the next TimeStep says where the episode is)
    for k, v in self._state.items():
      v.copy_(d[k])
    self._info.copy_(d['__info'])
    self._counters.copy_(d['__counters'])
    self._step_index = int(d['__step_index'])
    if self._state_tag_bit is not None:
      st = self._state['state']
      st &= ~self._state_lib_bits
      if self._step_index & 1:
        st |= self._state_tag_bit
    if self._device_step_counter:
      self._step_base.fill_(self._step_index)
    self._seed = int(d['__seed'])
    self._call_desc.stream.seed = self._seed
    if self._rng_mode == 'mt19937':
      self._mt_state.copy_(d['__mt_state'])
      self._mt_pos.copy_(d['__mt_pos'])
      self._mt_gauss.copy_(d['__mt_gauss'])
      self._mt_has_gauss.copy_(d['__mt_has_gauss'])
      if self._wrap_mt is not None:
        for k, v in self._wrap_mt.items():
          v.copy_(d['__wrap_mt_' + k])
    if '__wrap' in d:
      self._wrap = (tuple(d['__wrap']) + (0.0,))[:4]
    has_log = '__logging_steps' in d
    if has_log != (self._logging is not None):
      raise ValueError('state_dict was taken %s the Logging wrapper but this environment runs %s it: '
                       'wrap (or unwrap) the environment before load_state_dict'
                       % (('with', 'without') if has_log else ('without', 'with')))
    if has_log:
      if d['__logging_rows'].shape != self._logging['rows'].shape:
        raise ValueError('Logging row buffers differ in shape (max_rows): construct Logging with the same max_rows')
      for k in ('steps', 'episode', 'total_return', 'episode_len', 'episode_return', 'rows', 'n_rows'):
        self._logging[k].copy_(d['__logging_' + k])
