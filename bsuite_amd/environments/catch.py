"""Batched Catch (counterpart of bsuite/environments/catch.py; kernel: csrc/catch.hip)."""
import ctypes
from typing import Optional

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd.dm_env_compat import specs
from bsuite_amd.environments import base

NUM_EPISODES = 10000  # bsuite/experiments/catch/sweep.py:19
_ACTIONS = (-1, 0, 1)  # Left, no-op, right (catch.py:27).


class Catch(base.Environment):
  """Falling-ball / paddle grid; observation is the rows x columns board (catch.py:30-66)."""

  _info_keys = ('total_regret',)
  _info_pending_column = 'state'            # misses counted in the state word (bsx_bsuite_info)

  def __init__(self, rows: int = 10, columns: int = 5, seed: Optional[int] = None,
               **engine_kwargs):
    if not (2 <= rows <= 64 and 1 <= columns <= 64):
      raise ValueError('rows must be in [2,64] and columns in [1,64]')
    super().__init__(obs_shape=(rows, columns), num_actions=len(_ACTIONS), seed=seed,
                     **engine_kwargs)
    self._rows, self._columns = rows, columns
    self._cfg = _native.CatchCfg(rows, columns)
    self.bsuite_num_episodes = NUM_EPISODES

  def _state_tensors(self):
    return dict(state=torch.full((self._batch,), 1 << 24, dtype=torch.int32, device=self._device))

  def _pending_info(self):
    # Without the Logging wrapper the kernel counts a lane's misses in bits 25..31 of its packed state and adds
    # 2 * 127 to the total_regret column once per 127 misses (csrc/catch_fam.h); a miss costs regret 2 (catch.py:92-94).
    return {0: 2.0 * ((self._state['state'] >> 25) & 0x7F).to(torch.float64)}

  def _clear_pending_info(self):
    self._state['state'] &= 0x01FFFFFF      # the misses just folded into the column (the Logging kernels never count here)

  _abi_name = 'catch'
  _supports_delta = True
  _pipelined_rollout = True

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), out, self._info.data_ptr())

  def _check_scalar_action(self, action):
    _ACTIONS[action]  # IndexError exactly where catch.py:84 raises it  pylint: disable=pointless-statement

  def observation_spec(self) -> specs.BoundedArray:
    return specs.BoundedArray(shape=self._obs_shape, dtype=np.float32, name='observation',
                              minimum=0, maximum=1)

  def action_spec(self) -> specs.DiscreteArray:
    return specs.DiscreteArray(dtype=int, num_values=len(_ACTIONS), name='action')
