"""Reward wrappers (counterpart of bsuite/utils/wrappers.py: RewardNoise :250-310, RewardScale
:313-373).

In the reference these are Python objects that post-process every TimeStep.  Here they configure a
*fused epilogue* of the wrapped environment's kernel (bsx_reward_wrap_t in include/bsuite_amd.h):
non-FIRST rewards become `r + noise_scale * randn()` or `r * reward_scale`, evaluated in f64 like
the reference; step_type / discount / observation are untouched and `bsuite_info()` passes through
un-perturbed (wrappers.py:305-306, :368-369).  The wrapper classes keep the reference's surface:
`reset/step/observation_spec/action_spec/raw_env/bsuite_info` and attribute delegation.

`Logging` (wrappers.py:34-137) is the first "next" row: its per-step bookkeeping is
fused into the kernels' emit epilogue (bsx_logging_t) so that every lane accumulates exactly what
the reference wrapper would and snapshots a row at the same log-spaced counts; the Python class
below keeps the reference constructor and forwards rows to a `logger.write(dict)` object.
`ImageObservation` / `to_image` (wrappers.py:150-247) run as one store-stream kernel
over the whole batch (bsx_image_observation, csrc/image.hip).
"""
from typing import Any, Dict, List, Optional, Sequence

import ctypes

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd import dm_env_compat as dm_env
from bsuite_amd.environments import base


class _RewardWrapper(dm_env.EnvironmentBase):
  """Shared surface of the two reward wrappers."""

  def __init__(self, env):
    # `env` is an engine environment or another wrapper around one: the reference composes its wrappers
    # freely (utils/wrappers_test.py:123-131 stacks RewardNoise on RewardScale); here every reward
    # wrapper in the stack folds into the ONE fused epilogue of the raw environment's kernel.
    raw = env.raw_env if hasattr(env, 'raw_env') else env
    if not isinstance(raw, base.Environment):
      raise TypeError('bsuite_amd reward wrappers fuse into a bsuite_amd environment kernel; got '
                      f'{type(env).__name__}')
    if getattr(raw, '_logging', None) is not None:
      # The fused Logging bookkeeping tracks the reward the kernel's epilogue returns, i.e. it always behaves as the
      # OUTERMOST wrapper.  In the reference a Logging *inside* a reward wrapper records the un-perturbed rewards
      # (utils/wrappers.py:74-77 sees the inner env's timestep): that composition would silently log other
      # total_return / episode_return rows here, so it is refused.
      raise NotImplementedError('a reward wrapper around an environment that is already wrapped in Logging would log '
                                'the perturbed rewards here but the raw ones in the reference; wrap in this order: '
                                'Logging(RewardNoise(env), ...)')
    self._env = env
    self._raw = raw

  def reset(self):
    return self._env.reset()

  def step(self, action):
    return self._env.step(action)

  def rollout(self, actions):
    return self._env.rollout(actions)

  def observation_spec(self):
    return self._env.observation_spec()

  def action_spec(self):
    return self._env.action_spec()

  def _step(self, action: int):
    raise NotImplementedError('Please call step() instead of _step().')

  def _reset(self):
    raise NotImplementedError('Please call reset() instead of _reset().')

  @property
  def raw_env(self):
    # Recursively unwrap until we reach the true 'raw' env.
    wrapped = self._env
    if hasattr(wrapped, 'raw_env'):
      return wrapped.raw_env
    return wrapped

  def bsuite_info(self) -> Dict[str, Any]:
    return self._env.bsuite_info()

  def __getattr__(self, attr):
    """Delegate attribute access to underlying environment."""
    return getattr(self._env, attr)


class RewardNoise(_RewardWrapper):
  """Reward Noise environment wrapper (fused)."""

  def __init__(self, env: base.Environment, noise_scale: float, seed: Optional[int] = None):
    super().__init__(env)
    self._noise_scale = noise_scale
    # The reference wrapper owns a RandomState(seed) separate from the env's (wrappers.py:267);
    # here that is stream_id 1 of the draw stream, keyed by this seed.  In the MT19937-exact mode it
    # IS a second np.random.RandomState per lane (lane i: seed + i, or seed[i] for a sequence), whose
    # legacy randn the kernels reproduce bit for bit (include/bsx_stream.h bsx_mt_gauss).
    raw = self._raw
    if getattr(raw, '_rng_mode', 'philox') == 'mt19937':
      if seed is None:
        seeds = list(raw._mt_seeds)  # pylint: disable=protected-access
      elif isinstance(seed, (int, np.integer)):
        seeds = [(int(seed) + i) & 0xFFFFFFFF for i in range(raw.batch_size)]
      else:
        seeds = [int(x) for x in seed]
        if len(seeds) != raw.batch_size:
          raise ValueError('need one wrapper seed per lane')
      raw._set_wrap_mt_seeds(seeds)  # pylint: disable=protected-access
      wrap_seed = seeds[0]
    else:
      wrap_seed = raw.seed if seed is None else int(seed)
    wrap_seed &= (1 << 63) - 1
    kind, param, _, _ = raw._wrap  # pylint: disable=protected-access
    if kind == _native.WRAP_NONE:
      raw._wrap = (_native.WRAP_NOISE, float(noise_scale), wrap_seed, 0.0)  # pylint: disable=protected-access
    elif kind == _native.WRAP_SCALE:          # RewardNoise(RewardScale(env)): r*s + sigma*z
      raw._wrap = (_native.WRAP_SCALE_NOISE, param, wrap_seed, float(noise_scale))  # pylint: disable=protected-access
    else:
      raise NotImplementedError('the fused reward epilogue holds one RewardNoise and one RewardScale; '
                                'this environment already carries a RewardNoise')


class RewardScale(_RewardWrapper):
  """Reward Scale environment wrapper (fused)."""

  def __init__(self, env: base.Environment, reward_scale: float, seed: Optional[int] = None):
    super().__init__(env)
    self._reward_scale = reward_scale
    del seed  # the reference builds an unused RandomState (wrappers.py:330)
    raw = self._raw
    kind, param, wseed, _ = raw._wrap  # pylint: disable=protected-access
    if kind == _native.WRAP_NONE:
      raw._wrap = (_native.WRAP_SCALE, float(reward_scale), 0, 0.0)  # pylint: disable=protected-access
    elif kind == _native.WRAP_NOISE:          # RewardScale(RewardNoise(env)): (r + sigma*z)*s
      raw._wrap = (_native.WRAP_NOISE_SCALE, param, wseed, float(reward_scale))  # pylint: disable=protected-access
    else:
      raise NotImplementedError('the fused reward epilogue holds one RewardNoise and one RewardScale; '
                                'this environment already carries a RewardScale')


# Keys that are present for all experiments (wrappers.py:30-31).
STANDARD_KEYS = frozenset(['steps', 'episode', 'total_return', 'episode_len', 'episode_return'])


def _logarithmic_logging(episode: int, ratios: Optional[Sequence[float]] = None) -> bool:
  """Returns `True` only at specific ratios of 10**exponent (restated from wrappers.py:140-147)."""
  if ratios is None:
    ratios = [1., 1.2, 1.4, 1.7, 2., 2.5, 3., 4., 5., 6., 7., 8., 9., 10.]
  exponent = np.floor(np.log10(np.maximum(1, episode)))
  special_vals = [10**exponent * ratio for ratio in ratios]
  return any(episode == val for val in special_vals)


def logarithmic_logging_points(max_count: int) -> List[int]:
  """All counts in [0, max_count] at which `_logarithmic_logging` is true, ascending.  Candidates
  are the rounded ratio multiples of each decade; each is confirmed with the float test itself so
  the device table is exactly the reference predicate."""
  ratios = [1., 1.2, 1.4, 1.7, 2., 2.5, 3., 4., 5., 6., 7., 8., 9., 10.]
  cands = {0, 1}
  decade = 1
  while decade <= max_count:
    cands.update(int(round(decade * r)) for r in ratios)
    decade *= 10
  return sorted(c for c in cands if c <= max_count and _logarithmic_logging(c))


class Logging(_RewardWrapper):
  """Environment wrapper to track and log bsuite stats (batched counterpart of wrappers.py:34-137).

  Scalar view (`batch=None` env): every row the reference would write is passed to
  `logger.write(dict)` right after the step that produced it — a drop-in for
  `bsuite.utils.wrappers.Logging`.  Batched view: rows accumulate on the device per lane; read them
  with `rows(lane)` / `dataframe(lane)` (or all lanes with `all_rows()`); `logger` may be None.
  """

  def __init__(self, env, logger=None, log_by_step: bool = False, log_every: bool = False,
               max_rows: Optional[int] = None):
    raw = env.raw_env if hasattr(env, 'raw_env') else env
    super().__init__(raw)
    self._env = env
    self._raw = raw
    self._logger = logger
    self._log_by_step = log_by_step
    self._log_every = log_every
    self._lg = raw.enable_logging(log_by_step=log_by_step, log_every=log_every, max_rows=max_rows)
    self._columns = raw.logging_columns()

  def flush(self):
    self.check_overflow()
    if hasattr(self._logger, 'flush'):
      self._logger.flush()

  def check_overflow(self):
    """Raises as soon as some lane has logged more rows than its buffer holds (one 4-byte device-to-host read).  The
    batched row buffer is sized for the log points within 100 x bsuite_num_episodes episodes (enable_logging); a run
    beyond that horizon keeps counting rows but stores no more of them — call this (or flush(), or counters(check=True)) now
    and then rather than learning it from rows() after the run; Logging(..., max_rows=...) sizes the buffer."""
    cap = self._lg['rows'].shape[1]
    worst = int(self._lg['n_rows'].max().item())
    if worst > cap:
      raise RuntimeError(f'a lane has logged {worst} rows but the buffer holds {cap}: rows are being dropped; '
                         'construct Logging(..., max_rows=...) for the length of the run')

  def _forward_new_rows(self):
    if self._logger is None or not self._raw._scalar:  # pylint: disable=protected-access
      return
    # Rows are consumed right after the step that produced them (at most two per call), then the
    # lane's row buffer is rewound: a run of any length never fills it.
    n = int(self._lg['n_rows'][0].item())
    if n:
      if n > self._lg['rows'].shape[1]:
        raise RuntimeError(f'{n} log rows in one call exceed the row buffer ({self._lg["rows"].shape[1]})')
      rows = self._lg['rows'][0, :n].cpu().numpy()
      for r in rows:
        self._logger.write(self._row_dict(r))
      self._lg['n_rows'].zero_()
    if int(self._lg['episode'][0].item()) == self._raw.bsuite_num_episodes:
      self.flush()

  def _row_dict(self, r):
    out = {}
    for k, v in zip(self._columns, r):
      if k.startswith('_'):
        continue
      if k in ('steps', 'episode', 'episode_len') or k in self._raw._info_int_keys:  # pylint: disable=protected-access
        out[k] = int(v)
      else:
        out[k] = float(v)
    return out

  def reset(self):
    timestep = self._env.reset()
    self._forward_new_rows()
    return timestep

  def step(self, action):
    timestep = self._env.step(action)
    self._forward_new_rows()
    return timestep

  @property
  def raw_env(self):
    return self._raw

  # -- batched access ---------------------------------------------------------------------
  def num_rows(self):
    """int32 [B] device tensor: rows each lane has logged so far."""
    return self._lg['n_rows']

  def overflowed(self):
    """bool [B] device tensor: lanes that logged more rows than the buffer holds (rows past
    `max_rows` are counted but not stored; size `max_rows` for the run, see enable_logging)."""
    return self._lg['n_rows'] > self._lg['rows'].shape[1]

  def rows(self, lane: int = 0) -> List[Dict[str, Any]]:
    n = int(self._lg['n_rows'][lane].item())
    cap = self._lg['rows'].shape[1]
    if n > cap:
      raise RuntimeError(f'lane {lane} logged {n} rows but the buffer holds {cap}: rows were dropped; '
                         'construct Logging(..., max_rows=...) for the length of the run')
    return [self._row_dict(r) for r in self._lg['rows'][lane, :n].cpu().numpy()]

  def all_rows(self) -> List[List[Dict[str, Any]]]:
    """rows(lane) for every lane, with one device-to-host copy; raises if any lane overflowed."""
    n_rows = self._lg['n_rows'].cpu().numpy()
    cap = self._lg['rows'].shape[1]
    if (n_rows > cap).any():
      bad = int((n_rows > cap).sum())
      raise RuntimeError(f'{bad} lane(s) logged more rows than the buffer holds ({cap}): rows were dropped')
    rows = self._lg['rows'].cpu().numpy()
    return [[self._row_dict(r) for r in rows[l, :n_rows[l]]] for l in range(len(n_rows))]

  def dataframe(self, lane: int = 0):
    import pandas as pd  # pylint: disable=import-outside-toplevel
    return pd.DataFrame(self.rows(lane))

  def counters(self, check: bool = False) -> Dict[str, Any]:
    """The live per-lane accumulators as device tensors [B] (steps, episode, total_return, ...).  Never synchronises
    with the host (and is safe inside a HIP-graph capture) unless `check=True`, which first raises if a lane's row
    buffer has overflowed (check_overflow: one device-to-host read); `overflowed()` is the same fact as a device
    tensor."""
    if check:
      self.check_overflow()
    return {k: self._lg[k] for k in ('steps', 'episode', 'total_return', 'episode_len', 'episode_return')}


# ---------------------------------------------------------------------------------------------
# ImageObservation (wrappers.py:150-247)
def _image_cfg(shape: Sequence[int], obs_shape: Sequence[int]) -> '_native.ImageCfg':
  """Validates like `to_image` (wrappers.py:222-247) and picks the rule."""
  shape = tuple(int(s) for s in shape)
  assert len(shape) >= 2
  obs_shape = tuple(int(s) for s in obs_shape)
  size = int(np.prod(obs_shape)) if obs_shape else 1
  tail = int(np.prod(shape[2:])) if len(shape) > 2 else 1
  if size <= 4:                                   # _small_state_to_image (:178-204)
    return _native.ImageCfg(_native.IMAGE_SMALL, 1, size, shape[0], shape[1], tail)
  if len(obs_shape) > 2:
    raise ValueError('Cannot convert observation shape {} to desired shape {}'.format(obs_shape, shape))
  rows, cols = (1, obs_shape[0]) if len(obs_shape) == 1 else obs_shape   # np.expand_dims(obs, 0) (:212-213)
  cfg = _native.ImageCfg(_native.IMAGE_BILINEAR, rows, cols, shape[0], shape[1], tail)
  if shape[0] < rows or shape[1] < cols:
    # skimage.transform.resize(anti_aliasing=None): some output dimension is smaller than the input's
    # -> Gaussian pre-filter with sigma = max(0, (in/out - 1)/2) per axis (skimage/transform/_warps.py),
    # applied by scipy.ndimage.gaussian_filter; the kernel takes the half kernels built as scipy builds
    # them (scipy/ndimage/_filters.py _gaussian_kernel1d, truncate = 4)
    for axis, (n_in, n_out) in (('y', (rows, shape[0])), ('x', (cols, shape[1]))):
      half = _gaussian_half_kernel(n_in, n_out)
      if half is None:
        continue
      if len(half) - 1 > _native.IMAGE_MAX_RADIUS:
        raise ValueError(f'to_image: shrinking {n_in} -> {n_out} needs an anti-aliasing kernel of radius '
                         f'{len(half) - 1} > {_native.IMAGE_MAX_RADIUS}')
      setattr(cfg, 'radius_' + axis, len(half) - 1)
      arr = getattr(cfg, 'gauss_' + axis)
      for j, v in enumerate(half):
        arr[j] = float(v)
  return cfg


def _gaussian_half_kernel(n_in: int, n_out: int):
  """Weights at distance 0..radius of the anti-aliasing Gaussian along one axis, or None when the axis
  is not filtered (it does not shrink, or the kernel has a single tap)."""
  sigma = max(0.0, (np.float64(n_in) / np.float64(n_out) - 1) / 2)
  if not sigma > 1e-15:                               # scipy's gaussian_filter skips such axes
    return None
  radius = int(4.0 * float(sigma) + 0.5)
  if radius == 0:
    return None                                     # weights [1.0]: exact identity
  x = np.arange(-radius, radius + 1)
  phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
  phi = phi / phi.sum()
  return phi[radius:]


def to_image(shape: Sequence[int], observation, out: Optional[torch.Tensor] = None, batched=None):
  """Converts bsuite observations into an image-like format on the device (wrappers.py:222-247).

  observation: a device tensor `[B, *obs_shape]` (batched; returns a device tensor `[B, *shape]`)
  or a single numpy observation (returns numpy, like the reference; one H2D + D2H — compatibility
  path).  Values are tiled (size <= 4) or bilinearly interpolated (skimage >= 0.19 `resize` =
  scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True), after skimage's anti-aliasing Gaussian
  along every axis that shrinks) and broadcast over trailing dims."""
  shape = tuple(int(s) for s in shape)
  if batched is None:
    batched = torch.is_tensor(observation)
  if not batched:
    obs_np = np.ascontiguousarray(np.asarray(observation, dtype=np.float32))
    dev = torch.device('cuda', torch.cuda.current_device())
    res = to_image(shape, torch.from_numpy(obs_np).to(dev).unsqueeze(0), batched=True)
    return res[0].cpu().numpy().astype(np.asarray(observation).dtype, copy=False)
  obs = observation
  if obs.dtype != torch.float32 or not obs.is_cuda:
    raise TypeError('to_image: batched observations must be float32 device tensors')
  obs = obs.contiguous()
  B = int(obs.shape[0])
  cfg = _image_cfg(shape, obs.shape[1:])
  if out is None:
    out = torch.empty((B,) + shape, dtype=torch.float32, device=obs.device)
  elif out.shape != (B,) + shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != obs.device:
    raise ValueError('to_image: `out` must be a contiguous float32 tensor of shape (B, *shape) on the same device')
  rc = _native.lib.bsx_image_observation(ctypes.byref(cfg), B, obs.data_ptr(), out.data_ptr(),
                                         torch.cuda.current_stream(obs.device).cuda_stream)
  _native.check(rc, 'to_image')
  return out


class ImageObservation(dm_env.EnvironmentBase):
  """Environment wrapper to convert observations to an image-like format (wrappers.py:150-175).

  Batched environments keep `num_buffers` image buffers `[B, *shape]` on the device and fill one per
  call with a single kernel launch; the scalar view returns numpy images like the reference."""

  def __init__(self, env, shape: Sequence[int], num_buffers: int = 2):
    self._env = env
    self._shape = tuple(int(s) for s in shape)
    _image_cfg(self._shape, env.observation_spec().shape)       # validate once, like the first to_image call would
    self._num_buffers = max(1, int(num_buffers))
    self._images = None
    self._buf = 0

  def observation_spec(self):
    spec = self._env.observation_spec()
    return dm_env.specs.Array(shape=self._shape, dtype=spec.dtype, name=spec.name)

  def action_spec(self):
    return self._env.action_spec()

  def _convert(self, timestep):
    obs = timestep.observation
    if not torch.is_tensor(obs):
      return timestep._replace(observation=to_image(self._shape, obs))
    if self._images is None:
      self._images = [torch.empty((obs.shape[0],) + self._shape, dtype=torch.float32, device=obs.device)
                      for _ in range(self._num_buffers)]
    out = self._images[self._buf]
    self._buf = (self._buf + 1) % self._num_buffers
    return timestep._replace(observation=to_image(self._shape, obs, out=out))

  def reset(self):
    return self._convert(self._env.reset())

  def step(self, action):
    return self._convert(self._env.step(action))

  def __getattr__(self, attr):
    """Delegate attribute access to underlying environment."""
    return getattr(self._env, attr)
