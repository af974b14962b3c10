"""CPU: include/bsx_libm_log.h == this host's libm `log`, bit for bit, on 10^7 arguments — the pin that
lets the MT19937-exact mode reproduce np.random.RandomState.randn (whose polar Box-Muller calls libm log).
The header restates glibc's x86-64 FMA build of `log`; on a host whose libm resolves `log` to another
build (no FMA/AVX2) the last bit may differ, so the comparison is skipped there."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_has_fma():
  try:
    with open('/proc/cpuinfo') as f:
      flags = next(l for l in f if l.startswith('flags')).split()
    return 'fma' in flags and 'avx2' in flags
  except (OSError, StopIteration):
    return False


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
  so = str(tmp_path_factory.mktemp('lg') / 'libm_log_shim.so')
  subprocess.check_call(['gcc', '-O2', '-std=gnu99', '-ffp-contract=off', '-shared', '-fPIC',
                         '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'csrc', 'libm_log_shim.c'),
                         '-o', so, '-lm'])
  return ctypes.CDLL(so)


def _log(shim, x):
  x = np.ascontiguousarray(x, np.float64)
  y = np.empty_like(x)
  shim.shim_log(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(x.size), y.ctypes.data_as(ctypes.c_void_p))
  return y


@pytest.mark.skipif(not _cpu_has_fma(), reason='host libm does not run the FMA build of log')
def test_bit_exact_against_host_libm(shim):
  rng = np.random.default_rng(0)
  x = np.concatenate([
      rng.random(6_000_000),                                   # (0, 1): where r2 of the polar method lives
      1.0 - rng.random(1_500_000) * 2.0 ** -4,                 # the near-1 branch [1 - 2^-4, 1)
      1.0 + rng.random(500_000) * 0.07,                        # its upper half (1, 1 + 0x1.09p-4)
      np.exp(rng.uniform(-700, 700, 1_500_000)),               # the whole normal range
      2.0 ** -rng.integers(1, 1000, 400_000).astype(np.float64) * rng.random(400_000),
      np.array([1.0, 0.5, 2.0, 0.9375, 1.0647, np.nextafter(1.0, 0), np.nextafter(1.0, 2), 2.2250738585072014e-308,
                1.7976931348623157e308, 2.0 ** -104]),
  ])
  x = np.ascontiguousarray(x[(x > 2.3e-308) & np.isfinite(x)])
  assert x.size > 9_900_000
  got = _log(shim, x)
  # numpy's vectorised np.log may run its own SIMD kernel; legacy randn makes scalar libm calls, so the
  # reference here is libm's `log` called element by element from C on the same buffer
  ref = np.empty_like(x)
  shim.shim_host_log(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(x.size), ref.ctypes.data_as(ctypes.c_void_p))
  np.testing.assert_array_equal(got.view(np.uint64), ref.view(np.uint64))
  import math
  assert all(math.log(float(v)) == float(g) for v, g in zip(x[:20000], got[:20000]))       # and Python's view of it


def test_accuracy_everywhere(shim):
  """Independent of which libm build the host runs: within 1 ulp of the correctly rounded log."""
  rng = np.random.default_rng(1)
  x = np.concatenate([rng.random(500_000), 1.0 + (rng.random(200_000) - 0.5) * 0.12, np.exp(rng.uniform(-700, 700, 300_000))])
  got = _log(shim, x)
  ref = np.log(x)
  ulp = np.spacing(np.abs(ref))
  assert np.max(np.abs(got - ref) / np.maximum(ulp, 5e-324)) <= 1.0
