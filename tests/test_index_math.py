"""CPU: the workgroup index arithmetic the kernels compile (bsuite_amd/csrc/bsx_index.h), through gcc —
the roles of a pipelined launch's workgroups and the flat bit planes of a wide-row tile."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
  so = str(tmp_path_factory.mktemp('ix') / 'index_shim.so')
  subprocess.check_call(['gcc', '-O2', '-std=gnu99', '-shared', '-fPIC',
                         os.path.join(ROOT, 'tests', 'csrc', 'index_shim.c'), '-o', so])
  return ctypes.CDLL(so)


def _ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize('place', [0, 1, 2])
@pytest.mark.parametrize('adv,stream', [(4096, 14400), (4213, 52000), (1, 1), (5, 3), (7, 0), (0, 9), (100, 101),
                                        (3, 1000), (4096, 230400), (17, 16)])
def test_pipelined_roles_partition_the_grid(shim, place, adv, stream):
  """Every placement hands out advance indices 0..A-1 and stream indices 0..S-1, each exactly once and in grid
  order (the store stream must stay in address order)."""
  grid = adv + stream
  is_adv = np.zeros(grid, np.int32)
  index = np.zeros(grid, np.uint32)
  shim.shim_pipe_roles(ctypes.c_uint32(grid), ctypes.c_uint32(adv), ctypes.c_uint32(place), _ptr(is_adv), _ptr(index))
  np.testing.assert_array_equal(index[is_adv != 0], np.arange(adv, dtype=np.uint32))
  np.testing.assert_array_equal(index[is_adv == 0], np.arange(stream, dtype=np.uint32))
  if place == 0:
    assert is_adv[:adv].all() and not is_adv[adv:].any()
  elif place == 1:
    assert is_adv[stream:].all() and not is_adv[:stream].any()
  elif adv > 1 and stream >= adv:
    gaps = np.diff(np.nonzero(is_adv)[0])
    assert gaps.min() == gaps.max() == grid // adv                  # evenly spread


def test_direct_shape_rule(shim):
  assert [n for n in range(1, 40) if shim.shim_direct_shape(n)] == [1, 2, 3, 4, 5, 6, 7, 8]


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 256), st.integers(0, 3), st.integers(1, 253), st.integers(0, 2 ** 31))
def test_bit_planes_hold_every_lanes_bits_where_the_store_loop_reads_them(shim, lanes, head, nbits, seed):
  """A lane's bit string lands at flat bits [l*numel + head, l*numel + head + nbits) whatever the alignment —
  pieces that straddle a 32-bit word are split, neighbouring lanes share words — and nothing else is set."""
  numel = head + nbits
  rng = np.random.default_rng(seed)
  wpl = (nbits + 31) // 32
  bits = rng.integers(0, 2 ** 32, size=(lanes, wpl), dtype=np.uint64).astype(np.uint32)   # garbage above nbits: masked
  plane_words = (256 * numel + 31) // 32
  plane = np.zeros(plane_words + 1, np.uint32)
  out = np.zeros(lanes * numel, np.uint8)
  shim.shim_plane_roundtrip(lanes, numel, head, nbits, _ptr(np.ascontiguousarray(bits)), wpl, _ptr(plane), plane_words, _ptr(out))
  want = np.zeros((lanes, numel), np.uint8)
  for b in range(nbits):
    want[:, head + b] = (bits[:, b >> 5] >> np.uint32(b & 31)) & 1
  np.testing.assert_array_equal(out.reshape(lanes, numel), want)
  assert plane[plane_words] == 0                                    # never past the tile
