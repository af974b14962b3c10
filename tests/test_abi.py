"""CPU: the C-ABI library loads and exports every symbol include/bsuite_amd.h declares (no compute
calls without a GPU), argument validation returns the documented codes, and the header's struct
layouts match the ctypes mirrors."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'bsuite_amd.h')


def declared_symbols():
  text = open(HEADER).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(bsx_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported():
  from bsuite_amd import _native
  names = declared_symbols()
  assert len(names) >= 12
  out = subprocess.check_output(['nm', '-D', '--defined-only', _native.SO_PATH], text=True)
  exported = set(l.split()[-1] for l in out.splitlines() if ' T ' in l)
  missing = [n for n in names if n not in exported]
  assert not missing, missing
  assert sorted(_native.EXPORTED) == names          # the ctypes binding covers the whole header


def test_struct_layouts_match_header():
  """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
  from bsuite_amd import _native
  src = r'''
#include <stdio.h>
#include <stddef.h>
#include "bsuite_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu ", sizeof(bsx_stream_t), sizeof(bsx_reward_wrap_t),
         sizeof(bsx_timestep_t), sizeof(bsx_call_t), sizeof(bsx_deep_sea_t), sizeof(bsx_catch_t),
         sizeof(bsx_bandit_t), sizeof(bsx_cartpole_t), offsetof(bsx_call_t, counters),
         offsetof(bsx_cartpole_t, move_cost), offsetof(bsx_cartpole_t, time_frac),
         sizeof(bsx_logging_t), sizeof(bsx_mnist_t), offsetof(bsx_call_t, logging),
         offsetof(bsx_stream_t, mt_pos), offsetof(bsx_logging_t, log_by_step));
  printf("%zu %zu\n", offsetof(bsx_call_t, action_ring), offsetof(bsx_call_t, row_scratch));
  return 0;
}'''
  import tempfile
  with tempfile.TemporaryDirectory() as d:
    c = os.path.join(d, 'l.c')
    open(c, 'w').write(src)
    exe = os.path.join(d, 'l')
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
    got = [int(x) for x in subprocess.check_output([exe], text=True).split()]
  want = [ctypes.sizeof(_native.Stream), ctypes.sizeof(_native.RewardWrap),
          ctypes.sizeof(_native.TimeStepPtrs), ctypes.sizeof(_native.Call),
          ctypes.sizeof(_native.DeepSeaCfg), ctypes.sizeof(_native.CatchCfg),
          ctypes.sizeof(_native.BanditCfg), ctypes.sizeof(_native.CartpoleCfg),
          _native.Call.counters.offset, _native.CartpoleCfg.move_cost.offset,
          _native.CartpoleCfg.time_frac.offset,
          ctypes.sizeof(_native.Logging), ctypes.sizeof(_native.MnistCfg), _native.Call.logging.offset,
          _native.Stream.mt_pos.offset, _native.Logging.log_by_step.offset, _native.Call.action_ring.offset, _native.Call.row_scratch.offset]
  assert got == want


def test_argument_errors_without_touching_the_gpu():
  from bsuite_amd import _native
  lib = _native.lib
  assert lib.bsx_abi_version() == 12
  assert lib.bsx_strerror(0) == b'ok'
  cfg = _native.DeepSeaCfg(size=10, deterministic=1, move_cost=0.001, inv_size=0.1)
  call = _native.Call(n_lanes=4)
  out = _native.TimeStepPtrs(0, 0, 0, 0)
  assert lib.bsx_deep_sea_step(ctypes.byref(cfg), ctypes.byref(call), 0, 0, out, 0) == -2   # BSX_ENULL
  out = _native.TimeStepPtrs(16, 16, 16, 24)
  assert lib.bsx_deep_sea_step(ctypes.byref(cfg), ctypes.byref(call), 16, 16, out, 16) == -3  # BSX_EALIGN
  out = _native.TimeStepPtrs(16, 16, 16, 32)
  cfg.size = 65
  assert lib.bsx_deep_sea_step(ctypes.byref(cfg), ctypes.byref(call), 16, 16, out, 16) == -4  # BSX_ERANGE
  call.n_lanes = -1
  assert lib.bsx_deep_sea_step(ctypes.byref(cfg), ctypes.byref(call), 16, 16, out, 16) == -1  # BSX_EINVAL
  call.n_lanes = 0
  cfg.size = 10
  assert lib.bsx_deep_sea_step(ctypes.byref(cfg), ctypes.byref(call), 0, 0, out, 0) == 0     # empty batch
  assert b'NULL' in lib.bsx_strerror(-2)
  # ABI v10: the action ring is a power of two of rows, single-step calls only
  call.n_lanes, call.action_ring = 4, 3
  assert lib.bsx_deep_sea_step(ctypes.byref(cfg), ctypes.byref(call), 16, 16, out, 16) == -1  # BSX_EINVAL
  call.action_ring, call.n_steps = 4, 2
  assert lib.bsx_deep_sea_step(ctypes.byref(cfg), ctypes.byref(call), 16, 16, out, 16) == -5  # BSX_EMODE
  call.action_ring, call.n_steps = 0, 0
  # ABI v12: the row scratch of the chains' wide rows must be 16-byte aligned; bsuite_info through the ABI checks its arguments
  uc, uout = _native.UmbrellaChainCfg(5, 20), _native.TimeStepPtrs(16, 16, 16, 32)
  call.row_scratch = 24
  assert lib.bsx_umbrella_chain_step(ctypes.byref(uc), ctypes.byref(call), 16, 16, uout, 16) == -3   # BSX_EALIGN
  call.row_scratch = None
  assert lib.bsx_row_scratch_bytes(_native.FAMILY_IDS['umbrella_chain'], 23, 64) == 4 * (2 * 23 + 64)
  assert lib.bsx_bsuite_info(_native.FAMILY_IDS['catch'], 0, 8, None, 16, 1, 1, 16, None) == -2      # pending part needs the state column
  assert lib.bsx_bsuite_info(_native.FAMILY_IDS['catch'], 0, 0, None, None, 1, 1, None, None) == 0  # empty batch
  assert lib.bsx_bsuite_info(12, 0, 8, 16, 16, 1, 1, 16, None) == -1
  # ABI v9: pipelined group step / phase-0 trace refuse what they cannot run (host-side checks only)
  assert lib.bsx_group_step_pipelined(None, None, None) == -2
  assert lib.bsx_group_trace(None, None, 0) == -2
  g = ctypes.c_void_p()
  assert lib.bsx_group_create(_native.FAMILY_IDS['bandit'], 1, ctypes.byref(g)) == 0
  assert lib.bsx_group_trace(g, None, 0) == -5                       # BSX_EMODE: whole-sweep groups only
  sw = ctypes.c_void_p()
  assert lib.bsx_group_create(_native.FAMILY_IDS['sweep_mixed'], 1, ctypes.byref(sw)) == 0
  assert lib.bsx_group_trace(sw, 16, 1 << 20) == -1                  # not committed yet: its size is unknown
  lib.bsx_group_destroy(sw)
  assert lib.bsx_group_step_pipelined(g, g, None) == -1              # not committed, not a whole-sweep group
  assert lib.bsx_group_small_class(3) == lib.bsx_group_small_class(200) == 256
  lib.bsx_group_destroy(g)


def test_graft_entry_build_runs():
  """The driver's build check: compiles (or finds up to date) every HIP source + the oracle checker."""
  import __graft_entry__ as g
  g.build()
