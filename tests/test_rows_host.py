"""CPU: the packed wide rows of memory_chain / umbrella_chain (bsuite_amd/csrc/bsx_rows.h, ABI v12 row scratch) through
gcc — the lane's thread packs its row (HEAD floats + bit planes), the store stream decodes every 16-byte chunk of the
[B x numel] observation array, chunks that run over a row boundary and ragged tails included.  The device compiles the
very same header; what the kernels add (global loads / stores, the per-lane environment step) is covered on the GPU
(tests/test_gpu_wide_rows.py)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UMBRELLA, MEMORY = 0, 1


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
  so = str(tmp_path_factory.mktemp('rows') / 'rows_shim.so')
  subprocess.check_call(['gcc', '-O2', '-std=gnu99', '-shared', '-fPIC',
                         os.path.join(ROOT, 'tests', 'csrc', 'rows_shim.c'), '-o', so])
  lib = ctypes.CDLL(so)
  lib.shim_row_stream.restype = ctypes.c_int64
  return lib


def _ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def div_magic(d):                       # bsx_div_magic (csrc/bsx_host.h)
  return (0x100000000 // d) + 1


def div64(d):                           # bsx_make_div64 (csrc/bsx_host.h)
  lg = d.bit_length() - 1
  s = lg - 1
  return ((1 << (64 + s)) // d + 1) & ((1 << 64) - 1), s


def expected(kind, head, bits0, bits1):
  """The dense rows the reference would emit: HEAD floats, then one float per bit element."""
  if kind == UMBRELLA:
    body = bits0.astype(np.float32)                                      # umbrella_chain.py:65 (0.0 / 1.0)
  else:
    body = np.where(bits0 != 0, 2.0 * bits1.astype(np.float32) - 1.0, 0.0).astype(np.float32)   # memory_chain.py:69-70
  return np.concatenate([head.view(np.float32), body], axis=1)


@pytest.mark.parametrize('kind,numel', [(UMBRELLA, n) for n in (9, 10, 23, 35, 36, 67, 103, 131, 256)] +
                         [(MEMORY, n) for n in (9, 10, 12, 19, 27, 34, 35, 42, 64)])
@pytest.mark.parametrize('lanes', [1, 3, 1025])
@pytest.mark.parametrize('K', [1, 2, 4])
def test_row_stream_decodes_what_the_lanes_packed(shim, kind, numel, lanes, K):
  rng = np.random.default_rng(kind * 100000 + numel * 100 + lanes)
  HEAD = 2 if kind == MEMORY else 3
  nbits = numel - HEAD
  RW = shim.shim_row_words(numel, kind)
  W = shim.shim_plane_words(numel, kind)
  assert RW % 4 == 0 and RW >= HEAD + (2 if kind == MEMORY else 1) * W and W == (nbits + 31) // 32
  head = rng.integers(0, 2 ** 32, size=(lanes, HEAD), dtype=np.uint64).astype(np.uint32)
  head &= np.uint32(0x7FFFFFFF)                                          # (any bit pattern but NaN payload noise is fine; keep them finite-ish)
  bits0 = rng.integers(0, 2, size=(lanes, nbits)).astype(np.uint8)
  bits1 = rng.integers(0, 2, size=(lanes, nbits)).astype(np.uint8)
  if kind == MEMORY:                                                     # rows of all-zero / all-set planes, as t != 0 / t == 0
    bits0[::3] = 1
    bits0[1::3] = 0
  rows = np.full((lanes + 1, RW), 0xDEADBEEF, np.uint32)                  # (+1: poison behind the last row — never read)
  for l in range(lanes):
    shim.shim_pack_row(_ptr(rows[l]), numel, kind, _ptr(head[l]), _ptr(bits0[l]), _ptr(bits1[l]))
  out = np.full(lanes * numel + 8, 0xABABABAB, np.uint32)
  m, s = div64(numel)
  blocks = shim.shim_row_stream(_ptr(rows), ctypes.c_int64(lanes), numel, kind, K, ctypes.c_uint32(div_magic(numel)),
                                ctypes.c_uint64(m), ctypes.c_uint32(s), _ptr(out))
  assert blocks == (lanes * numel + K * 1024 - 1) // (K * 1024)
  want = expected(kind, head, bits0, bits1).reshape(-1).view(np.uint32)
  np.testing.assert_array_equal(out[:lanes * numel], want)
  assert (out[lanes * numel:] == 0xABABABAB).all()                       # nothing beyond the array


def test_row_words_match_the_abi():
  """bsx_row_scratch_words (the C ABI) == the header's formula, 0 for short rows and other families."""
  from bsuite_amd import _native
  for numel in range(1, 257):
    for fam, kind in ((3, MEMORY), (4, UMBRELLA)):
      head, planes = (2, 2) if kind == MEMORY else (3, 1)
      want = 0 if numel <= 8 else (head + planes * ((numel - head + 31) // 32) + 3) // 4 * 4
      assert _native.lib.bsx_row_scratch_words(fam, numel) == want, (fam, numel)
    assert _native.lib.bsx_row_scratch_words(0, numel) == 0 and _native.lib.bsx_row_scratch_words(6, numel) == 0
  assert _native.lib.bsx_row_scratch_words(4, 257) == 0
