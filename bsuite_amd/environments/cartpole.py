"""Batched Cartpole and CartpoleSwingup (counterparts of bsuite/environments/cartpole.py and
bsuite/experiments/cartpole_swingup/cartpole_swingup.py; kernel: csrc/small_obs.hip).

Device state is f32 (the reference holds Python floats); `time_elapsed` is kept as an integer step
count k, with the reference's f64 running sum `time_elapsed += timescale` (cartpole.py:63) replayed
on the host once to find the first k whose sum exceeds `max_time` (the reference terminates at step
1001, not 1000, for the defaults) and to tabulate the f32 `time_elapsed / max_time` observation.
"""
import collections
import ctypes
from typing import Optional

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd.dm_env_compat import specs
from bsuite_amd.environments import base

NUM_EPISODES = 1000  # bsuite/experiments/cartpole/sweep.py:19
_MAX_TABLE = 1 << 22

CartpoleConfig = collections.namedtuple(
    'CartpoleConfig', ['mass_cart', 'mass_pole', 'length', 'force_mag', 'gravity'])


def _time_table(timescale: float, max_time: float):
  """Replays `time_elapsed += timescale` in f64; returns (last_step, f32 table of t_k/max_time)."""
  sums = [0.0]
  t = 0.0
  while not t > max_time:
    t = t + timescale
    sums.append(t)
    if len(sums) > _MAX_TABLE:
      raise ValueError('max_time / timescale too large for the time table')
  last_step = len(sums) - 1
  return last_step, (np.asarray(sums, np.float64) / max_time).astype(np.float32)


class _CartpoleBase(base.Environment):
  """Shared host side of the two cart-pole variants."""

  def __init__(self, swingup, obs_shape, height_threshold, theta_dot_threshold,
               x_reward_threshold, move_cost, x_threshold, timescale, max_time, init_range, seed,
               engine_kwargs):
    super().__init__(obs_shape=obs_shape, num_actions=3, seed=seed, **engine_kwargs)
    self._height_threshold = height_threshold
    self._theta_dot_threshold = theta_dot_threshold
    self._x_reward_threshold = x_reward_threshold
    self._move_cost = move_cost
    self._x_threshold = x_threshold
    self._timescale = timescale
    self._max_time = max_time
    self._cartpole_config = CartpoleConfig(mass_cart=1., mass_pole=0.1, length=0.5, force_mag=10.,
                                           gravity=9.8)
    self._last_step, self._time_frac_host = _time_table(timescale, max_time)
    c = self._cartpole_config
    self._cfg = _native.CartpoleCfg(
        swingup=int(swingup), last_step=self._last_step, height_threshold=height_threshold,
        x_threshold=x_threshold, theta_dot_threshold=theta_dot_threshold,
        x_reward_threshold=x_reward_threshold, timescale=timescale, mass_cart=c.mass_cart,
        mass_pole=c.mass_pole, length=c.length, force_mag=c.force_mag, gravity=c.gravity,
        move_cost=move_cost, init_range=init_range, theta_offset=np.pi if swingup else 0.0,
        time_frac=None)
    self.bsuite_num_episodes = NUM_EPISODES

  def _state_tensors(self):
    self._time_frac = torch.from_numpy(self._time_frac_host).to(self._device)
    self._cfg.time_frac = self._time_frac.data_ptr()
    return dict(state=torch.zeros((4, self._batch), dtype=torch.float32, device=self._device),
                steps=torch.full((self._batch,), 1 << 30, dtype=torch.int32, device=self._device))

  _abi_name = 'cartpole'

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), self._state['steps'].data_ptr(), out, self._info.data_ptr())

  def action_spec(self):
    return specs.DiscreteArray(dtype=int, num_values=3, name='action')


class Cartpole(_CartpoleBase):
  """Classic cart-pole balancing task, 6-d observation (cartpole.py:68-116)."""

  _info_keys = ('raw_return', 'best_episode', '_episode_return', '_total_upright')
  _info_pending_column = 'steps'            # raw_return / episode_return of the running episode (bsx_bsuite_info)

  def _pending_info(self):
    # Rewards are 1 on every step that does not end the episode (cartpole.py:142-149), so a running
    # episode of k steps has earned exactly k; the kernel folds (k-1) + last reward into raw_return /
    # best_episode when the episode ends (csrc/small_obs.hip, cartpole_env).
    steps = self._state['steps']
    running = (steps & (1 << 30)) == 0
    k = torch.where(running, steps & 0x3FFFFFFF, torch.zeros_like(steps)).to(torch.float64)
    return {0: k, 2: k}

  def __init__(self,
               height_threshold: float = 0.8,
               x_threshold: float = 3.,
               timescale: float = 0.01,
               max_time: float = 10.,
               init_range: float = 0.05,
               seed: Optional[int] = None,
               **engine_kwargs):
    super().__init__(False, (1, 6), height_threshold, 1., 1., 0., x_threshold, timescale, max_time,
                     init_range, seed, engine_kwargs)


class CartpoleSwingup(_CartpoleBase):
  """Swing-up variant with a move cost, 8-d observation (cartpole_swingup.py:30-78)."""

  _info_keys = ('raw_return', 'best_episode', '_episode_return', 'total_upright')
  _info_variant = 1                         # accumulates per step like the reference: nothing pending

  def __init__(self,
               height_threshold: float = 0.5,
               theta_dot_threshold: float = 1.,
               x_reward_threshold: float = 1.,
               move_cost: float = 0.1,
               x_threshold: float = 3.,
               timescale: float = 0.01,
               max_time: float = 10.,
               init_range: float = 0.05,
               seed: Optional[int] = None,
               **engine_kwargs):
    super().__init__(True, (1, 8), height_threshold, theta_dot_threshold, x_reward_threshold,
                     move_cost, x_threshold, timescale, max_time, init_range, seed, engine_kwargs)

  def observation_spec(self):
    return specs.Array(shape=(1, 8), dtype=np.float32, name='state')
