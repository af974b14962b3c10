"""Batched MountainCar (counterpart of bsuite/environments/mountain_car.py; csrc/small_obs.hip)."""
import ctypes
from typing import Optional

import torch

from bsuite_amd import _native
from bsuite_amd.environments import base

NUM_EPISODES = 1000  # bsuite/experiments/mountain_car/sweep.py:19


class MountainCar(base.Environment):
  """Mountain Car, an underpowered car must power up a hill (mountain_car.py:29-57)."""

  _info_keys = ('raw_return',)
  _info_pending_column = 'steps'            # the running episode's -t (bsx_bsuite_info)

  def __init__(self, max_steps: int = 1000, seed: Optional[int] = None, **engine_kwargs):
    if max_steps < 1:
      raise ValueError('max_steps must be >= 1')
    super().__init__(obs_shape=(1, 3), num_actions=3, seed=seed, **engine_kwargs)
    self._max_steps = max_steps
    self._cfg = _native.MountainCarCfg(max_steps, 0)
    self.bsuite_num_episodes = NUM_EPISODES

  def _state_tensors(self):
    return dict(state=torch.zeros((2, self._batch), dtype=torch.float32, device=self._device),
                steps=torch.full((self._batch,), 1 << 30, dtype=torch.int32, device=self._device))

  _abi_name = 'mountain_car'

  def _pending_info(self):
    # every step pays -1 (mountain_car.py:75-76): a running episode of t steps has earned -t; the
    # kernel folds it into raw_return when the episode ends (csrc/small_obs.hip, mountain_car_env)
    steps = self._state['steps']
    running = (steps & (1 << 30)) == 0
    return {0: -torch.where(running, steps & 0x3FFFFFFF, torch.zeros_like(steps)).to(torch.float64)}

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), self._state['steps'].data_ptr(), out, self._info.data_ptr())
