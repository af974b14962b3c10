"""Batched DiscountingChain (counterpart of bsuite/environments/discounting_chain.py)."""
import ctypes
from typing import Any, Dict, Optional

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd.environments import base

NUM_EPISODES = 1000  # bsuite/experiments/discounting_chain/sweep.py:19


class DiscountingChain(base.Environment):
  """Five chains paying at t in {1,3,10,30,100}; one pays 10% more (discounting_chain.py:37-61)."""

  def __init__(self, mapping_seed: Optional[int] = None, **engine_kwargs):
    super().__init__(obs_shape=(1, 2), num_actions=5, **engine_kwargs)
    self._episode_len = 100
    self._reward_timestep = [1, 3, 10, 30, 100]
    self._n_actions = len(self._reward_timestep)
    if mapping_seed is None:
      mapping_seed = np.random.randint(0, self._n_actions)
    else:
      mapping_seed = mapping_seed % self._n_actions
    self._rewards = np.ones(self._n_actions)
    self._rewards[mapping_seed] += 0.1
    self._cfg = _native.DiscountingChainCfg(int(mapping_seed), 0)
    self.bsuite_num_episodes = NUM_EPISODES

  def _state_tensors(self):
    return dict(state=torch.full((self._batch,), 1 << 12, dtype=torch.int32, device=self._device))

  _abi_name = 'discounting_chain'

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), out)

  @property
  def optimal_return(self):
    return 1.1

  def bsuite_info(self) -> Dict[str, Any]:
    return {}
