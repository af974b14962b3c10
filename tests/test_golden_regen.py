"""CPU, build container only: regenerate every fixture under tests/golden/ from the UNMODIFIED reference
(python -m oracle.make_golden into a scratch directory) and require the result to equal the committed
files array for array — the fixtures are outputs of the reference, not hand-edited, and the generator
is deterministic.  Skipped where /root/reference does not exist (the GPU box)."""
import filecmp
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.mark.timeout(600)
@pytest.mark.skipif(not replay.reference_available(), reason='needs the reference tree (build container only)')
def test_fixtures_regenerate_identically(tmp_path):
  out = str(tmp_path / 'golden')
  os.makedirs(out)
  env = dict(os.environ, BSX_GOLDEN_OUT=out)
  subprocess.run([sys.executable, '-m', 'oracle.make_golden'], cwd=ROOT, env=env, check=True,
                 stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=500)
  new = sorted(os.path.relpath(p, out) for p in glob.glob(os.path.join(out, '**', '*'), recursive=True)
               if os.path.isfile(p))
  old = sorted(os.path.relpath(p, GOLDEN) for p in glob.glob(os.path.join(GOLDEN, '**', '*'), recursive=True)
               if os.path.isfile(p))
  assert new == old
  for rel in new:
    a, b = os.path.join(out, rel), os.path.join(GOLDEN, rel)
    if rel.endswith('.npz'):
      x, y = np.load(a), np.load(b)
      assert sorted(x.files) == sorted(y.files), rel
      for k in x.files:
        assert x[k].dtype == y[k].dtype and x[k].shape == y[k].shape, (rel, k)
        np.testing.assert_array_equal(x[k], y[k], err_msg=f'{rel}:{k}')     # NaN == NaN here
    else:
      assert filecmp.cmp(a, b, shallow=False), rel
