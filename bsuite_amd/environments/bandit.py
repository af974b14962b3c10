"""Batched SimpleBandit (counterpart of bsuite/environments/bandit.py; kernel: csrc/small_obs.hip)."""
import ctypes
from typing import Optional

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd.environments import base

NUM_EPISODES = 10000  # bsuite/experiments/bandit/sweep.py:19


class SimpleBandit(base.Environment):
  """11-armed deterministic bandit with permuted linspace rewards (bandit.py:35-51)."""

  _info_keys = ('total_regret',)

  def __init__(self, mapping_seed: Optional[int] = None, num_actions: int = 11, **engine_kwargs):
    if not 1 <= num_actions <= _native.BANDIT_MAX_ACTIONS:
      raise ValueError(f'num_actions must be in [1, {_native.BANDIT_MAX_ACTIONS}]')
    super().__init__(obs_shape=(1, 1), num_actions=num_actions, **engine_kwargs)
    self._rng = np.random.RandomState(mapping_seed)
    self._num_actions = num_actions
    action_mask = self._rng.choice(range(self._num_actions), size=self._num_actions, replace=False)
    self._rewards = np.linspace(0, 1, self._num_actions)[action_mask]
    self._optimal_return = 1.
    cfg = _native.BanditCfg()
    cfg.num_actions = num_actions
    for k in range(num_actions):
      cfg.rewards[k] = float(self._rewards[k])
    self._cfg = cfg
    self.bsuite_num_episodes = NUM_EPISODES

  def _state_tensors(self):
    return dict(state=torch.ones(self._batch, dtype=torch.int32, device=self._device))

  _abi_name = 'bandit'

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), out, self._info.data_ptr())

  def _check_scalar_action(self, action):
    self._rewards[action]  # IndexError where bandit.py:61 raises it  pylint: disable=pointless-statement
