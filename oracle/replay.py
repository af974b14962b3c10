"""ORACLE (test infrastructure): drive the UNMODIFIED reference environments with a replay of the
engine's draw stream.

Where /root/reference exists (the build container) the reference is imported from there — that is how
tests/golden/*.npz are produced (oracle/make_golden.py).  On the GPU box it is imported from the
byte-code tree oracle/stage_reference.py compiled from those sources at build time (oracle/_ref,
git-ignored): tests/test_gpu_vs_reference_live.py and bench.py's cpu_baseline leg use it there.
The product (bsuite_amd/) never imports this module.

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


def reference_source_available():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, 'bsuite'))


def reference_available():
  """The reference can be imported: from its sources (build container) or from the staged byte-code."""
  from oracle import stage_reference  # pylint: disable=import-outside-toplevel
  return reference_source_available() or stage_reference.staged()


def reference_origin():
  """'source' (/root/reference), 'staged' (oracle/_ref byte-code) or None."""
  from oracle import stage_reference  # pylint: disable=import-outside-toplevel
  if os.environ.get('BSX_REFERENCE_STAGED') and stage_reference.staged():
    return 'staged'                      # test hook: exercise the staged tree where the sources exist too
  if reference_source_available():
    return 'source'
  return 'staged' if stage_reference.staged() else None


def import_reference():
  """Import the real `bsuite` package (sources, else the staged byte-code) with the shim third-party modules."""
  from oracle import stage_reference  # pylint: disable=import-outside-toplevel
  origin = reference_origin()
  if origin is None:
    raise RuntimeError('reference not present: neither /root/reference nor oracle/_ref (python -m oracle.stage_reference '
                       'in the build container stages it)')
  root = REFERENCE_ROOT if origin == 'source' else stage_reference.STAGE_DIR
  shims = os.path.join(_HERE, 'ref_shims')
  sys.dont_write_bytecode = True         # never write __pycache__ into the read-only reference tree
  for p in (root, shims):
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
