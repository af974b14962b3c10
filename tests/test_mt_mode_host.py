"""CPU: the MT19937-exact samplers of include/bsx_stream.h against numpy's own RandomState, draw for
draw and state for state (the same header is compiled into the HIP kernels)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
  so = str(tmp_path_factory.mktemp('mt') / 'mt_shim.so')
  subprocess.check_call(['gcc', '-O2', '-std=gnu99', '-ffp-contract=off', '-shared', '-fPIC',
                         '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'csrc', 'mt_shim.c'),
                         '-o', so, '-lm'])
  return ctypes.CDLL(so)


def _run(shim, rs, ops, args):
  _, key, pos, _, _ = rs.get_state()
  state = np.ascontiguousarray(key, np.uint32).copy()
  p = ctypes.c_int32(int(pos))
  ops = np.ascontiguousarray(ops, np.int32)
  args = np.ascontiguousarray(args, np.uint32)
  out = np.zeros(len(ops), np.float64)
  shim.mt_run(state.ctypes.data_as(ctypes.c_void_p), ctypes.byref(p), len(ops),
              ops.ctypes.data_as(ctypes.c_void_p), args.ctypes.data_as(ctypes.c_void_p),
              out.ctypes.data_as(ctypes.c_void_p))
  return out, state, p.value


@pytest.mark.parametrize('seed', [0, 1, 42, 2**32 - 1, 123456789])
def test_samplers_match_numpy_randomstate(shim, seed):
  rng = np.random.default_rng(seed)
  n = 5000                                           # > 8 twists of the 624-word state
  ops = rng.integers(0, 3, size=n)
  args = rng.choice([1, 2, 3, 5, 7, 10, 40, 60000, 2**31 + 5], size=n)
  ref = np.random.RandomState(seed)
  got, state, pos = _run(shim, np.random.RandomState(seed), ops, args)
  want = np.zeros(n)
  for i in range(n):
    if ops[i] == 0:
      want[i] = ref.random_sample() if i % 2 else ref.rand()
    elif ops[i] == 1:
      want[i] = ref.binomial(1, 0.5)
    else:
      want[i] = ref.randint(int(args[i]))
  np.testing.assert_array_equal(got, want)
  _, key, rpos, _, _ = ref.get_state()
  # same generator state afterwards (positions may differ only by the lazy twist at 624)
  assert (pos % 624, list(state)) == (rpos % 624, list(key)) or (pos, list(state)) == (rpos, list(key))


def test_vector_binomial_and_uniform_forms(shim):
  ref = np.random.RandomState(7)
  ops = [1] * 23 + [0] * 4 + [2]
  args = [0] * 27 + [23]
  got, _, _ = _run(shim, np.random.RandomState(7), ops, args)
  want = list(ref.binomial(1, 0.5, size=23))                        # vector form == sequential draws
  want += [(ref.uniform(-0.05, 0.05) + 0.05) / 0.1 for _ in range(4)]   # lo + (hi-lo)*U
  want += [ref.randint(23)]
  np.testing.assert_allclose(got[:23], want[:23], rtol=0, atol=0)
  np.testing.assert_allclose(got[23:27], want[23:27], rtol=0, atol=1e-15)
  assert got[27] == want[27]


def _cpu_has_fma():
  try:
    with open('/proc/cpuinfo') as f:
      flags = next(l for l in f if l.startswith('flags')).split()
    return 'fma' in flags and 'avx2' in flags
  except (OSError, StopIteration):
    return False


def _run_gauss(shim, rs, ops, args=None):
  _, key, pos, has_gauss, gauss = rs.get_state()
  state = np.ascontiguousarray(key, np.uint32).copy()
  p, h, g = ctypes.c_int32(int(pos)), ctypes.c_int32(int(has_gauss)), ctypes.c_double(float(gauss))
  ops = np.ascontiguousarray(ops, np.int32)
  args = np.ascontiguousarray(np.zeros(len(ops)) if args is None else args, np.uint32)
  out = np.zeros(len(ops), np.float64)
  shim.mt_run_gauss(state.ctypes.data_as(ctypes.c_void_p), ctypes.byref(p), ctypes.byref(h), ctypes.byref(g),
                    len(ops), ops.ctypes.data_as(ctypes.c_void_p), args.ctypes.data_as(ctypes.c_void_p),
                    out.ctypes.data_as(ctypes.c_void_p))
  return out, state, p.value, h.value, g.value


@pytest.mark.skipif(not _cpu_has_fma(), reason='include/bsx_libm_log.h restates the FMA build of glibc log')
@pytest.mark.parametrize('seed', [0, 42, 2**32 - 1])
def test_randn_matches_numpy_randomstate_bit_for_bit(shim, seed):
  """>= 10^6 draws per seed: numpy's legacy polar Box-Muller (cached second value included) driven by
  the lane's MT19937, with the restated libm log — utils/wrappers.py:278, deep_sea.py:126."""
  n = 1_000_001                                      # odd: ends with a value in the cache
  got, state, pos, has_gauss, gauss = _run_gauss(shim, np.random.RandomState(seed), np.full(n, 4))
  ref = np.random.RandomState(seed)
  want = ref.randn(n)                                # vector form == n scalar randn() calls
  np.testing.assert_array_equal(got.view(np.uint64), want.view(np.uint64))
  _, key, rpos, rhas, rgauss = ref.get_state()
  assert (has_gauss, gauss) == (rhas, rgauss) and has_gauss == 1
  assert (pos % 624, list(state)) == (rpos % 624, list(key)) or (pos, list(state)) == (rpos, list(key))


@pytest.mark.skipif(not _cpu_has_fma(), reason='include/bsx_libm_log.h restates the FMA build of glibc log')
def test_randn_interleaved_with_the_other_samplers(shim):
  """deep_sea's stochastic step mixes randn() and rand() on one generator (deep_sea.py:126,130): the
  cached normal survives the interleaved uniform draws exactly as in numpy."""
  rng = np.random.default_rng(5)
  n = 200_000
  ops = rng.choice([0, 1, 2, 4], size=n, p=[0.3, 0.1, 0.1, 0.5])
  args = rng.choice([1, 2, 5, 30, 60000], size=n)
  ref = np.random.RandomState(99)
  got, _, _, _, _ = _run_gauss(shim, np.random.RandomState(99), ops, args)
  want = np.zeros(n)
  for i in range(n):
    if ops[i] == 0:
      want[i] = ref.rand()
    elif ops[i] == 1:
      want[i] = ref.binomial(1, 0.5)
    elif ops[i] == 2:
      want[i] = ref.randint(int(args[i]))
    else:
      want[i] = ref.randn()
  np.testing.assert_array_equal(got.view(np.uint64), want.view(np.uint64))
