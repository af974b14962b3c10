"""OpenAI-gym adapters over the batched engine (counterpart of bsuite/utils/gym_wrapper.py:30-184,
SURVEY §8 f-4).

`GymFromDMEnv` follows the reference's gym 4-tuple protocol (`step -> (obs, reward, done, info)`,
`reset -> obs`, gym_wrapper.py:41-54).  Over a scalar environment (`batch=None`) it is the reference
adapter value for value (`reward or 0.`, `game_over`, rgb_array rendering of the last observation).
Over a batched environment it is a *vector* adapter with the same call pattern: `obs` f32
`[B, *shape]`, `reward` f32 `[B]`, `done` bool `[B]` device tensors and no host synchronisation —
lanes that are done auto-reset on their next `step` (their action is ignored, reward 0), exactly as
the underlying dm_env protocol prescribes (base.py:59-65).

gym itself is optional: if it is not importable (it is not in this image) the module provides the
two tiny space classes the adapter needs (`Discrete`, `Box`) with gym's attribute names.
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from bsuite_amd import dm_env_compat as dm_env

specs = dm_env.specs

try:  # pragma: no cover - gym is not part of the build image
  import gym  # type: ignore
  from gym import spaces  # type: ignore
  _EnvBase = gym.Env
except ImportError:
  gym = None

  class _Space:
    def __init__(self, shape, dtype):
      self.shape = tuple(shape)
      self.dtype = np.dtype(dtype)

  class _Discrete(_Space):
    def __init__(self, n):
      super().__init__((), np.int64)
      self.n = int(n)

    def contains(self, x):
      return 0 <= int(x) < self.n

    def __repr__(self):
      return f'Discrete({self.n})'

  class _Box(_Space):
    def __init__(self, low, high, shape, dtype=np.float32):
      super().__init__(shape, dtype)
      self.low = np.full(self.shape, low, dtype=self.dtype)
      self.high = np.full(self.shape, high, dtype=self.dtype)

    def contains(self, x):
      x = np.asarray(x)
      return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
      return f'Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})'

  class _Spaces:  # the subset of `gym.spaces` used here
    Discrete = _Discrete
    Box = _Box

  spaces = _Spaces()
  _EnvBase = object


class GymFromDMEnv(_EnvBase):
  """Presents a (scalar or batched) bsuite_amd environment through the gym step/reset protocol."""

  metadata = {'render.modes': ['human', 'rgb_array']}

  def __init__(self, env):
    self._env = env
    self._last_observation = None
    self.viewer = None
    self.game_over = False          # Dopamine reads this (gym_wrapper.py:39); a bool tensor [B] when batched

  def step(self, action):
    ts = self._env.step(action)
    self._last_observation = ts.observation
    if torch.is_tensor(ts.step_type):                           # batched: stay on the device
      done = ts.step_type == int(dm_env.StepType.LAST)
      self.game_over = done if self.game_over is False else (self.game_over | done)
      return ts.observation, ts.reward, done, {}                # reward is already 0 on FIRST lanes
    if ts.last():
      self.game_over = True
    return ts.observation, (ts.reward or 0.), ts.last(), {}

  def reset(self):
    self.game_over = False
    ts = self._env.reset()
    self._last_observation = ts.observation
    return ts.observation

  def render(self, mode: str = 'rgb_array'):
    if self._last_observation is None:
      raise ValueError('Environment not ready to render. Call reset() first.')
    if mode == 'rgb_array':
      return self._last_observation
    if mode == 'human':
      if gym is None:
        raise RuntimeError("render(mode='human') needs gym's SimpleImageViewer; gym is not installed")
      if self.viewer is None:
        from gym.envs.classic_control import rendering  # pylint: disable=import-outside-toplevel
        self.viewer = rendering.SimpleImageViewer()
      self.viewer.imshow(self._last_observation)
      return self.viewer.isopen
    return None

  @staticmethod
  def _bounds(spec):
    """(low, high) of a spec: its own bounds when it has them, else the whole real line."""
    if isinstance(spec, specs.BoundedArray):
      return spec.minimum, spec.maximum
    return -float('inf'), float('inf')

  @property
  def observation_space(self):
    spec = self._env.observation_spec()
    lo, hi = self._bounds(spec)
    return spaces.Box(low=float(lo), high=float(hi), shape=spec.shape, dtype=spec.dtype)

  @property
  def action_space(self):
    return spaces.Discrete(self._env.action_spec().num_values)

  @property
  def reward_range(self) -> Tuple[float, float]:
    return self._bounds(self._env.reward_spec())

  def __getattr__(self, attr):
    """Delegate attribute access to underlying environment."""
    return getattr(self._env, attr)


def _bounded(space, lo, hi, name):
  return specs.BoundedArray(shape=space.shape, dtype=space.dtype, minimum=lo, maximum=hi, name=name)


# gym space class name -> spec constructor (gym_wrapper.py:103-139); dispatch is by NAME so that gym's
# classes and the stand-ins above are treated alike.
_SPACE_TO_SPEC = {
    'Discrete': lambda sp, name: specs.DiscreteArray(num_values=sp.n, dtype=sp.dtype, name=name),
    'Box': lambda sp, name: _bounded(sp, sp.low, sp.high, name),
    'MultiBinary': lambda sp, name: _bounded(sp, 0.0, 1.0, name),
    'MultiDiscrete': lambda sp, name: _bounded(sp, np.zeros(sp.shape), sp.nvec, name),
    'Tuple': lambda sp, name: tuple(space2spec(child, name) for child in sp.spaces),
    'Dict': lambda sp, name: {key: space2spec(child, name) for key, child in sp.spaces.items()},
}


def space2spec(space, name: Optional[str] = None):
  """gym space -> dm_env spec; Tuple / Dict spaces map to nested tuples / dicts of specs."""
  convert = _SPACE_TO_SPEC.get(type(space).__name__.lstrip('_'))
  if convert is None:
    raise ValueError('Unexpected gym space: {}'.format(space))
  return convert(space, name)


class DMEnvFromGym(dm_env.EnvironmentBase):
  """Presents a gym environment (4-tuple API) as a dm_env.Environment (gym_wrapper.py:142-184)."""

  def __init__(self, gym_env):
    self.gym_env = gym_env
    self._specs = (space2spec(gym_env.observation_space, name='observations'),
                   space2spec(gym_env.action_space, name='actions'))
    self._reset_next_step = True      # like every bsuite environment, the first step() is a reset

  def observation_spec(self):
    return self._specs[0]

  def action_spec(self):
    return self._specs[1]

  def reset(self):
    self._reset_next_step = False
    return dm_env.restart(self.gym_env.reset())

  def step(self, action):
    if self._reset_next_step:
      return self.reset()
    observation, reward, done, info = self.gym_env.step(action)
    self._reset_next_step = bool(done)
    if not done:
      return dm_env.transition(reward, observation)
    # a time-limit cut keeps the discount at 1 (truncation), a real end of episode sets it to 0
    ended = dm_env.truncation if info.get('TimeLimit.truncated', False) else dm_env.termination
    return ended(reward, observation)

  def close(self):
    self.gym_env.close()
