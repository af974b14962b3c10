"""Batched UmbrellaChain (counterpart of bsuite/environments/umbrella_chain.py; csrc/small_obs.hip)."""
import ctypes
from typing import Optional

import torch

from bsuite_amd import _native
from bsuite_amd.environments import base

NUM_EPISODES = 10000  # bsuite/experiments/umbrella_length/sweep.py:19


class UmbrellaChain(base.Environment):
  """Credit assignment over a chain with Bernoulli distractors (umbrella_chain.py:34-58)."""

  _info_keys = ('total_regret',)

  def __init__(self, chain_length: int, n_distractor: int = 0, seed: Optional[int] = None,
               **engine_kwargs):
    if chain_length < 1 or not 0 <= n_distractor <= 253:
      raise ValueError('chain_length must be >= 1 and n_distractor in [0, 253]')
    super().__init__(obs_shape=(1, 3 + n_distractor), num_actions=2, seed=seed, **engine_kwargs)
    self._chain_length = chain_length
    self._n_distractor = n_distractor
    self._cfg = _native.UmbrellaChainCfg(chain_length, n_distractor)
    self.bsuite_num_episodes = NUM_EPISODES

  def _mt_constructor_draws(self, rs):
    rs.binomial(1, 0.5)                      # umbrella_chain.py:55 (need_umbrella)

  def _state_tensors(self):
    return dict(state=torch.full((self._batch,), 1 << 22, dtype=torch.int32, device=self._device))

  _abi_name = 'umbrella_chain'

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), out, self._info.data_ptr())

  @property
  def optimal_return(self):
    return 1
