"""Reward wrappers (counterpart of bsuite/utils/wrappers.py: RewardNoise :250-310, RewardScale
:313-373).

In the reference these are Python objects that post-process every TimeStep.  Here they configure a
*fused epilogue* of the wrapped environment's kernel (bsx_reward_wrap_t in include/bsuite_amd.h):
non-FIRST rewards become `r + noise_scale * randn()` or `r * reward_scale`, evaluated in f64 like
the reference; step_type / discount / observation are untouched and `bsuite_info()` passes through
un-perturbed (wrappers.py:305-306, :368-369).  The wrapper classes keep the reference's surface:
`reset/step/observation_spec/action_spec/raw_env/bsuite_info` and attribute delegation.

`Logging` and `ImageObservation` (wrappers.py:34-247) are host-side bookkeeping / adapters and are
out of scope of the hot path.
"""
from typing import Any, Dict, Optional

from bsuite_amd import _native
from bsuite_amd import dm_env_compat as dm_env
from bsuite_amd.environments import base


class _RewardWrapper(dm_env.EnvironmentBase):
  """Shared surface of the two reward wrappers."""

  def __init__(self, env: base.Environment):
    if not isinstance(env, base.Environment):
      raise TypeError('bsuite_amd reward wrappers fuse into a bsuite_amd environment kernel; got '
                      f'{type(env).__name__}')
    self._env = env

  def reset(self):
    return self._env.reset()

  def step(self, action):
    return self._env.step(action)

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
    # here that is stream_id 1 of the draw stream, keyed by this seed.
    wrap_seed = env.seed if seed is None else int(seed)
    env._wrap = (_native.WRAP_NOISE, float(noise_scale), wrap_seed & ((1 << 63) - 1))  # pylint: disable=protected-access


class RewardScale(_RewardWrapper):
  """Reward Scale environment wrapper (fused)."""

  def __init__(self, env: base.Environment, reward_scale: float, seed: Optional[int] = None):
    super().__init__(env)
    self._reward_scale = reward_scale
    del seed  # the reference builds an unused RandomState (wrappers.py:330)
    env._wrap = (_native.WRAP_SCALE, float(reward_scale), 0)  # pylint: disable=protected-access
