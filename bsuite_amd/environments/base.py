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
import secrets
from typing import Any, Dict, Optional

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd import dm_env_compat as dm_env
from bsuite_amd.dm_env_compat import specs

_MASK63 = (1 << 63) - 1


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
  _info_keys = ()          # names of the f64 info columns, in native column order
  _info_int_keys = ()      # keys the reference reports as Python ints

  def __init__(self, obs_shape, num_actions, *, seed=None, batch=None, device=None,
               lane_offset=0, num_buffers=2):
    self._scalar = batch is None
    self._batch = 1 if batch is None else int(batch)
    if self._batch < 1:
      raise ValueError('batch must be >= 1')
    self._device = torch.device('cuda:0' if device is None else device)
    self._lane_offset = int(lane_offset)
    self._seed = _resolve_seed(seed)
    self._obs_shape = tuple(int(d) for d in obs_shape)
    self._num_actions = int(num_actions)
    self._num_buffers = max(1, int(num_buffers))
    self._wrap = (_native.WRAP_NONE, 0.0, 0)
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
    """Index the next reset()/step() call will use in the draw stream."""
    return self._step_index

  def _state_tensors(self) -> Dict[str, torch.Tensor]:
    """Subclass hook: allocate the family's SoA state columns with their initial values."""
    raise NotImplementedError

  def _launch(self, call, action_ptr, out) -> int:
    """Subclass hook: call the family's C-ABI entry point."""
    raise NotImplementedError

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
      self._counters = torch.zeros(2, dtype=torch.int64, device=dev)
      self._out = []
      for _ in range(self._num_buffers):
        self._out.append(dict(
            reward=torch.empty(B, dtype=torch.float32, device=dev),
            discount=torch.empty(B, dtype=torch.float32, device=dev),
            step_type=torch.empty(B, dtype=torch.int8, device=dev),
            observation=torch.empty((B,) + self._obs_shape, dtype=torch.float32, device=dev)))
      self._scalar_action = torch.zeros(1, dtype=torch.int32, device=dev)
    self._allocated = True

  # ----------------------------------------------------------------------------------------
  # the hot path
  def _call(self, action, force_reset: bool):
    self._ensure_allocated()
    out = self._out[self._buf]
    self._buf = (self._buf + 1) % self._num_buffers
    kind, param, wseed = self._wrap
    call = _native.Call(
        n_lanes=self._batch, force_reset=int(force_reset),
        stream=_native.Stream(self._seed, self._lane_offset, self._step_index, None),
        wrap=_native.RewardWrap(kind, 0, param, wseed),
        counters=self._counters.data_ptr(),
        hip_stream=torch.cuda.current_stream(self._device).cuda_stream)
    ptrs = _native.TimeStepPtrs(out['reward'].data_ptr(), out['discount'].data_ptr(),
                                out['step_type'].data_ptr(), out['observation'].data_ptr())
    rc = self._launch(call, 0 if action is None else action.data_ptr(), ptrs)
    _native.check(rc, f'{type(self).__name__} step')
    self._step_index += 1
    return out

  def _coerce_actions(self, action) -> torch.Tensor:
    if self._scalar:
      a = int(action)
      self._check_scalar_action(a)
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

  def _wrap_output(self, out):
    if not self._scalar:
      return dm_env.TimeStep(step_type=out['step_type'], reward=out['reward'],
                             discount=out['discount'], observation=out['observation'])
    st = int(out['step_type'].item())
    obs = out['observation'][0].cpu().numpy()
    if st == _native.FIRST:
      return dm_env.restart(obs)
    reward = float(out['reward'].item())
    if st == _native.LAST:
      return dm_env.TimeStep(dm_env.StepType.LAST, reward, float(out['discount'].item()), obs)
    return dm_env.transition(reward=reward, observation=obs,
                             discount=float(out['discount'].item()))

  def reset(self) -> dm_env.TimeStep:
    """Resets every lane (base.py:54-57) and returns the FIRST TimeStep."""
    self._reset_next_step = False
    self._ensure_allocated()
    return self._wrap_output(self._call(None, force_reset=True))

  def step(self, action) -> dm_env.TimeStep:
    """Steps every lane; lanes whose previous step was LAST (or that are fresh) reset instead and
    ignore their action (base.py:59-65)."""
    self._ensure_allocated()
    return self._wrap_output(self._call(self._coerce_actions(action), force_reset=False))

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

  def bsuite_info(self) -> Dict[str, Any]:
    """Logging metadata (base.py:75-77).  Scalar view: Python numbers as in the reference.
    Batched view: one f64 device tensor [B] per key (views of the engine's accumulators)."""
    self._ensure_allocated()
    if self._scalar:
      vals = self._info[:, 0].cpu().numpy()
      out = {}
      for j, k in enumerate(self._info_keys):
        if not k.startswith('_'):      # '_x' columns are engine-internal accumulators
          out[k] = int(vals[j]) if k in self._info_int_keys else float(vals[j])
      return out
    return {k: self._info[j] for j, k in enumerate(self._info_keys) if not k.startswith('_')}

  def episode_counters(self) -> torch.Tensor:
    """int64 [2] device tensor: lanes that emitted LAST, lanes that emitted FIRST (all calls)."""
    self._ensure_allocated()
    return self._counters

  def state_dict(self) -> Dict[str, Any]:
    """Everything needed to resume this batch bit-exactly (device tensors are cloned)."""
    self._ensure_allocated()
    d = {k: v.clone() for k, v in self._state.items()}
    d['__info'] = self._info.clone()
    d['__counters'] = self._counters.clone()
    d['__step_index'] = self._step_index
    d['__seed'] = self._seed
    return d

  def load_state_dict(self, d: Dict[str, Any]):
    self._ensure_allocated()
    for k, v in self._state.items():
      v.copy_(d[k])
    self._info.copy_(d['__info'])
    self._counters.copy_(d['__counters'])
    self._step_index = int(d['__step_index'])
    self._seed = int(d['__seed'])
