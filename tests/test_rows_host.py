"""CPU: the flat bit planes of the chains' wide rows (bsuite_amd/csrc/bsx_rows.h, ABI v12 row scratch) through gcc —
rows are laid into the scratch the way the lane advance lays them (bit e of a plane = element e of the observation
array, genuine floats in their own columns), the store stream decodes every 16-byte chunk of the [B x numel] observation
array, chunks that run over a row boundary and ragged tails included.  The device compiles the very same header; what
the kernels add (the wave-level assembly of the planes in LDS, global loads / stores, the per-lane environment step) is
covered on the GPU (tests/test_gpu_wide_rows.py)."""
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
  for f in ('shim_row_stream', 'shim_scratch_words', 'shim_plane_words'):
    getattr(lib, f).restype = ctypes.c_int64
  return lib


def _ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def div_magic(d):                       # bsx_div_magic (csrc/bsx_host.h)
  return (0x100000000 // d) + 1


def div64(d):                           # bsx_make_div64 (csrc/bsx_host.h)
  lg = d.bit_length() - 1
  s = lg - 1
  return ((1 << (64 + s)) // d + 1) & ((1 << 64) - 1), s


def lay_out(shim, kind, rows_f32):
  """The scratch of bsx_rows.h for dense rows [lanes, numel] (f32): every element that is not a float head must be a
  value the bit planes can hold — 0 / 1 (umbrella), 0 / +-1 (memory)."""
  lanes, numel = rows_f32.shape
  pw = shim.shim_plane_words(ctypes.c_int64(lanes), numel)
  planes, nf = shim.shim_planes(kind), shim.shim_nf(kind)
  fpos = [shim.shim_fpos(kind, k) for k in range(nf)]
  scratch = np.full(shim.shim_scratch_words(kind, ctypes.c_int64(lanes), numel), 0, np.uint32)
  assert scratch.size == planes * pw + nf * lanes and pw == (lanes + 63) // 64 * 2 * numel
  flat = rows_f32.copy()
  flat[:, fpos] = 0.0
  flat = flat.reshape(-1)
  bits0 = (flat != 0).astype(np.uint8)
  bits1 = (flat > 0).astype(np.uint8)
  for p, bits in enumerate((bits0, bits1)[:planes]):
    padded = np.zeros(pw * 32, np.uint8)
    padded[:bits.size] = bits
    scratch[p * pw:(p + 1) * pw] = np.packbits(padded.reshape(-1, 32), axis=1, bitorder='little').view(np.uint32).reshape(-1)
  for k, pos in enumerate(fpos):
    scratch[planes * pw + k * lanes: planes * pw + (k + 1) * lanes] = rows_f32[:, pos].view(np.uint32)
  return scratch


@pytest.mark.parametrize('kind,numel', [(UMBRELLA, n) for n in (9, 10, 23, 32, 35, 36, 67, 103, 131, 256)] +
                         [(MEMORY, n) for n in (9, 10, 12, 19, 27, 32, 34, 35, 42, 64)])
@pytest.mark.parametrize('lanes', [1, 3, 63, 64, 65, 1025, 2240])
@pytest.mark.parametrize('K', [1, 2, 4])
def test_row_stream_decodes_the_flat_planes(shim, kind, numel, lanes, K):
  rng = np.random.default_rng(kind * 100000 + numel * 100 + lanes)
  nf = shim.shim_nf(kind)
  fpos = [shim.shim_fpos(kind, k) for k in range(nf)]
  if kind == UMBRELLA:
    rows = rng.integers(0, 2, size=(lanes, numel)).astype(np.float32)          # need, has, distractors: 0.0 / 1.0
  else:
    rows = (rng.integers(0, 2, size=(lanes, numel)) * 2 - 1).astype(np.float32)  # +-1 at t == 0 ...
    rows[1::3] = 0.0                                                             # ... zeros otherwise
  heads = rng.integers(0, 2 ** 31, size=(lanes, nf), dtype=np.int64).astype(np.uint32)   # any finite-ish bit pattern
  rows[:, fpos] = heads.view(np.float32)
  scratch = lay_out(shim, kind, rows)
  out = np.full(lanes * numel + 8, 0xABABABAB, np.uint32)
  m, s = div64(numel)
  blocks = shim.shim_row_stream(_ptr(scratch), ctypes.c_int64(lanes), numel, kind, K, ctypes.c_uint32(div_magic(numel)),
                                ctypes.c_uint64(m), ctypes.c_uint32(s), _ptr(out))
  assert blocks == (lanes * numel + K * 1024 - 1) // (K * 1024)
  np.testing.assert_array_equal(out[:lanes * numel], rows.reshape(-1).view(np.uint32))
  assert (out[lanes * numel:] == 0xABABABAB).all()                       # nothing beyond the array


def test_scratch_size_matches_the_abi():
  """bsx_row_scratch_bytes (the C ABI) == the header's formula, 0 for short rows and other families."""
  from bsuite_amd import _native
  for lanes in (1, 64, 65, 2240, 1 << 20):
    for numel in (3, 8, 9, 23, 42, 103, 256, 257):
      for fam, planes, nf in ((3, 2, 2), (4, 1, 1)):
        want = 0 if (numel <= 8 or numel > 256) else 4 * (planes * ((lanes + 63) // 64) * 2 * numel + nf * lanes)
        assert _native.lib.bsx_row_scratch_bytes(fam, numel, lanes) == want, (fam, numel, lanes)
      assert _native.lib.bsx_row_scratch_bytes(0, numel, lanes) == 0 and _native.lib.bsx_row_scratch_bytes(6, numel, lanes) == 0
