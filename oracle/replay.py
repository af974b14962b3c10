"""ORACLE (test infrastructure): drive the UNMODIFIED reference environments with a replay of the
engine's draw stream.

Only usable where /root/reference exists (this build container) — it is how tests/golden/*.npz are
produced (oracle/make_golden.py).  Nothing under tests -m gpu / bench.py / smoke() imports it.

`ReplayRNG` quacks like the np.random.RandomState members the reference calls
(deep_sea.py:126,130; catch.py:71; memory_chain.py:94-95; umbrella_chain.py:65,83,89-90;
cartpole.py:91-92; mountain_car.py:69; mnist.py:63; utils/wrappers.py:278) and maps each call to
the draw primitive of include/bsx_stream.h that the device kernels use at the same program point.
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get('BSX_REFERENCE_ROOT', '/root/reference')

from oracle import stream as S  # noqa: E402


def reference_available():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, 'bsuite'))


def import_reference():
  """Import the real `bsuite` package from /root/reference with the shim third-party modules."""
  if not reference_available():
    raise RuntimeError('reference tree not present (only exists in the build container)')
  shims = os.path.join(_HERE, 'ref_shims')
  for p in (REFERENCE_ROOT, shims):
    if p not in sys.path:
      sys.path.insert(0, p)
  import bsuite  # pylint: disable=import-outside-toplevel
  return bsuite


class ReplayRNG:
  """np.random.RandomState look-alike fed by one lane of the bsx stream."""

  def __init__(self, seed, lane, stream_id=S.STREAM_ENV):
    self._s = S.LaneStream(seed, lane, stream_id)

  def begin_step(self, step):
    self._s.begin_step(step)

  # -- members used by the reference -------------------------------------------------------
  def rand(self):
    return self._s.uniform01()

  def random_sample(self):
    return self._s.uniform01()

  def randn(self):
    return self._s.normal()

  def uniform(self, low=0.0, high=1.0):
    return low + (high - low) * self._s.uniform01()

  def randint(self, n):
    return self._s.randint(n)

  def binomial(self, n, p, size=None):
    assert n == 1 and p == 0.5, 'the reference only draws fair coins'
    if size is None:
      return self._s.bern()
    if isinstance(size, (list, tuple)):
      raise NotImplementedError('only host-side constants use shaped draws')
    return self._s.bern_vec(int(size))


def attach_replay(env, seed, lane, wrap_seed=None):
  """Swap the reference env's RandomState(s) for replays; returns the list to `begin_step` on."""
  rngs = []
  inner = env
  while hasattr(inner, '_env') and hasattr(inner, '_rng'):  # RewardNoise / RewardScale wrappers, possibly stacked
    # (utils/wrappers.py:250,313).  Only RewardNoise draws; every wrapper of a stack replays the one
    # wrapper stream (RewardScale's generator is never used, wrappers.py:330).
    w = ReplayRNG(seed if wrap_seed is None else wrap_seed, lane, S.STREAM_WRAP)
    inner._rng = w  # pylint: disable=protected-access
    rngs.append(w)
    inner = inner._env  # pylint: disable=protected-access
  r = ReplayRNG(seed, lane, S.STREAM_ENV)
  inner._rng = r  # pylint: disable=protected-access
  rngs.append(r)
  return rngs
