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
    pay_times = (1, 3, 10, 30, 100)                  # discounting_chain.py:49
    super().__init__(obs_shape=(1, 2), num_actions=len(pay_times), **engine_kwargs)
    # the chain that pays 10 % more: mapping_seed mod 5, a host-side random pick when unseeded (:50-58)
    bonus = int(np.random.randint(0, len(pay_times))) if mapping_seed is None else int(mapping_seed) % len(pay_times)
    rewards = np.ones(len(pay_times))
    rewards[bonus] += 0.1
    self._reward_timestep, self._n_actions, self._episode_len = list(pay_times), len(pay_times), pay_times[-1]
    self._rewards = rewards
    self._cfg = _native.DiscountingChainCfg(bonus, 0)
    self.bsuite_num_episodes = NUM_EPISODES

  def _state_tensors(self):
    return dict(state=torch.full((self._batch,), 1 << 12, dtype=torch.int32, device=self._device))

  _abi_name = 'discounting_chain'

  def _check_scalar_action(self, action):
    # discounting_chain.py:76-81: the episode's FIRST action becomes the context and indexes two Python lists on every step
    # of the episode: IndexError outside -5..4, raised on that first step (later actions are never looked at); -5..-1 wrap
    # like Python's negative indices while the observation shows the negative context — the kernels do the same, and
    # count lanes outside -5..4 in invalid_action_count() instead of raising.
    if self._scalar_last_type == _native.FIRST:
      self._reward_timestep[action]  # pylint: disable=pointless-statement

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), out)

  @property
  def optimal_return(self):
    return 1.1

  def bsuite_info(self) -> Dict[str, Any]:
    return {}
