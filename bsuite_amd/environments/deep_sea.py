"""Batched 'Deep Sea' exploration environment (counterpart of bsuite/environments/deep_sea.py).

Same constructor arguments and semantics as `DeepSea` (deep_sea.py:51-101); the dynamics
(`_step` :116-144, `_reset` :110-114, `_get_observation` :103-108) run in
bsuite_amd/csrc/deep_sea.hip.  The per-cell action mapping is drawn on the host with numpy's
RandomState exactly as the reference does (:76-85) and shipped to the kernel as N*N bits.
"""
import ctypes
import warnings
from typing import Optional

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd.environments import base

NUM_EPISODES = 10000  # bsuite/experiments/deep_sea/sweep.py:19


class DeepSea(base.Environment):
  """Deep Sea environment to test for deep exploration (batched)."""

  _info_keys = ('total_bad_episodes', 'denoised_return')
  _info_int_keys = ('total_bad_episodes',)

  def __init__(self,
               size: int,
               deterministic: bool = True,
               unscaled_move_cost: float = 0.01,
               randomize_actions: bool = True,
               seed: Optional[int] = None,
               mapping_seed: Optional[int] = None,
               **engine_kwargs):
    if not 1 <= size <= _native.DEEP_SEA_MAX_SIZE:
      raise ValueError(f'size must be in [1, {_native.DEEP_SEA_MAX_SIZE}]')
    super().__init__(obs_shape=(size, size), num_actions=2, seed=seed, **engine_kwargs)
    self._size = size
    self._deterministic = deterministic
    self._unscaled_move_cost = unscaled_move_cost

    if randomize_actions:
      self._mapping_rng = np.random.RandomState(mapping_seed)
      self._action_mapping = self._mapping_rng.binomial(1, 0.5, [size, size])
    else:
      warnings.warn('Environment is in debug mode (randomize_actions=False).'
                    'Only randomized_actions=True is the DeepSea environment.')
      self._action_mapping = np.ones([size, size])

    if not self._deterministic:  # action 'right' only succeeds (1 - 1/N)
      optimal_no_cost = (1 - 1 / self._size) ** (self._size - 1)
    else:
      optimal_no_cost = 1.
    self._optimal_return = optimal_no_cost - self._unscaled_move_cost

    cfg = _native.DeepSeaCfg()
    cfg.size = size
    cfg.deterministic = int(bool(deterministic))
    cfg.move_cost = float(unscaled_move_cost) / size     # f64, as deep_sea.py:132 evaluates it
    cfg.inv_size = 1 / size                              # f64, deep_sea.py:130
    flat = np.asarray(self._action_mapping).reshape(-1) == 1
    for idx in np.nonzero(flat)[0]:
      cfg.mapping_bits[int(idx) >> 5] |= (1 << (int(idx) & 31))
    self._cfg = cfg
    self.bsuite_num_episodes = NUM_EPISODES

  def _state_tensors(self):
    return dict(state=torch.full((self._batch,), 1 << 17, dtype=torch.int32, device=self._device))

  _abi_name = 'deep_sea'
  _supports_delta = True
  _pipelined_rollout = True

  # Bit 18 of the packed state word = the parity of the call index that reads it next (csrc/deep_sea_fam.h: every
  # advance writes it).  The base class keeps that true for words that arrive from elsewhere (load_state_dict) and sets
  # BSX_CALL_STATE_TAGGED where call indices are guaranteed consecutive (base.Environment._state_tag_bit): a
  # deterministic, un-wrapped step() is then ONE launch (deep_sea_step1_kernel) instead of lane advance + observation stream.
  _state_tag_bit = 1 << 18
  _state_lib_bits = 1 << 18

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), out, self._info.data_ptr())

  @property
  def optimal_return(self):
    return self._optimal_return
