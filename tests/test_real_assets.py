"""Self-activating checks against the REAL assets this image lacks (VERDICT r05 missing #2): the real MNIST idx files
(bsuite/utils/datasets.py:42-69 downloads them; no network here) and a real scikit-image (bsuite/utils/wrappers.py:216
calls skimage.transform.resize; not installed).  Today 60 of the 468 bsuite_ids and the ImageObservation adapter are pinned
on stand-ins: a synthetic dataset in the same idx wire format (tests/golden/mnist_synthetic_dataset.npz) and the published
algorithm of `resize` over scipy.ndimage (oracle/ref_shims/skimage/transform.py).  Every test here SKIPS while the asset
is absent and runs, unchanged, the day it appears:

  * MNIST: the four *-ubyte.gz files in $BSX_REAL_MNIST_DIR, /opt/datasets/mnist, ~/.cache/mnist or /tmp/mnist, recognised by
    what they hold (60 000 28x28 training images whose label histogram is MNIST's), not by where they are — the test
    suite itself writes its synthetic stand-in into /tmp/mnist, the directory the reference hard-wires;
  * scikit-image: an importable `skimage` that is not the stand-in under oracle/ref_shims.
"""
import gzip
import hashlib
import os
import struct
import sys

import numpy as np
import pytest

from bsuite_amd.utils import datasets
from tests import golden_util as gu

# label counts of the 60 000 MNIST training images (digits 0..9) and the md5 digests torchvision publishes for the files
MNIST_TRAIN_LABEL_COUNTS = (5923, 6742, 5958, 6131, 5842, 5421, 5918, 6265, 5851, 5949)
MNIST_MD5 = {'train-images-idx3-ubyte.gz': 'f68b3c2dcbeaaa9fbdd348bbdeb94873', 'train-labels-idx1-ubyte.gz': 'd53e105ee54ea40749a09fcbcd1e9432',
             't10k-images-idx3-ubyte.gz': '9fb629c4189551a2d022fa330f9573f3', 't10k-labels-idx1-ubyte.gz': 'ec29112dd5afa0611ce80d1b7f02629c'}


def _holds_real_mnist(directory):
  try:
    p = lambda f: os.path.join(directory, f)  # noqa: E731
    if not all(os.path.isfile(p(f)) for f in datasets.FILES):
      return False
    with gzip.open(p(datasets.FILES[0]), 'rb') as fh:
      magic, n, rows, cols = struct.unpack('>IIII', fh.read(16))
    if (magic, n, rows, cols) != (2051, 60000, 28, 28):
      return False
    with gzip.open(p(datasets.FILES[1]), 'rb') as fh:
      magic, n = struct.unpack('>II', fh.read(8))
      labels = np.frombuffer(fh.read(), dtype=np.uint8)
    return (magic, n) == (2049, 60000) and tuple(np.bincount(labels, minlength=10)) == MNIST_TRAIN_LABEL_COUNTS
  except (OSError, struct.error, ValueError):
    return False


def real_mnist_dir():
  for d in (os.environ.get('BSX_REAL_MNIST_DIR'), '/opt/datasets/mnist', os.path.expanduser('~/.cache/mnist'), '/tmp/mnist'):
    if d and _holds_real_mnist(d):
      return d
  return None


def real_skimage():
  """The installed scikit-image, or None (the stand-in under oracle/ref_shims is NOT one)."""
  saved_path, saved_mods = list(sys.path), {k: v for k, v in sys.modules.items() if k == 'skimage' or k.startswith('skimage.')}
  try:
    sys.path[:] = [p for p in sys.path if 'ref_shims' not in p]
    for k in saved_mods:
      del sys.modules[k]
    import importlib
    try:
      mod = importlib.import_module('skimage.transform')
    except ImportError:
      return None
    if 'ref_shims' in (getattr(mod, '__file__', '') or ''):
      return None
    return mod
  finally:
    sys.path[:] = saved_path
    for k in [k for k in sys.modules if k == 'skimage' or k.startswith('skimage.')]:
      if k not in saved_mods:
        sys.modules.pop(k, None)
    sys.modules.update(saved_mods)


# ------------------------------------------------------------------------------------------------------------ MNIST
def test_detector_rejects_the_synthetic_stand_in(tmp_path):
  """(runs always) the stand-in in the real wire format is not mistaken for the real files, nor is an empty directory."""
  imgs, labels = gu.mnist_dataset()
  datasets.write_idx_files(str(tmp_path), imgs.view(np.uint8), labels)
  assert not _holds_real_mnist(str(tmp_path)) and not _holds_real_mnist(str(tmp_path / 'nothing'))


def test_real_mnist_parses_like_the_reference():
  """CPU: bsuite_amd.utils.datasets.load_mnist on the real files == the reference's loader on them (int8 quirk of
  datasets.py:55-56 included); digests reported against the published ones."""
  d = real_mnist_dir()
  if d is None:
    pytest.skip('real MNIST files are not on this machine (no network): the mnist ids stay pinned on the synthetic idx files')
  (tr_x, tr_y), (te_x, te_y) = datasets.load_mnist(d)
  assert tr_x.shape == (60000, 28, 28) and tr_x.dtype == np.int8 and te_x.shape == (10000, 28, 28)
  assert tr_y.shape == (60000,) and te_y.shape == (10000,) and tr_x.min() < 0          # bright pixels are negative
  for f, want in MNIST_MD5.items():
    with open(os.path.join(d, f), 'rb') as fh:
      got = hashlib.md5(fh.read()).hexdigest()
    assert got == want, f'{f}: md5 {got}, published {want} (a re-compressed copy? the contents matched MNIST)'
  from oracle import replay
  if replay.reference_available():
    replay.import_reference()
    from bsuite.utils import datasets as ref_datasets
    (rx, ry), (sx, sy) = ref_datasets.load_mnist(d)
    for a, b in ((tr_x, rx), (tr_y, ry), (te_x, sx), (te_y, sy)):
      assert a.dtype == b.dtype and np.array_equal(a, b)


@pytest.mark.gpu
def test_real_mnist_bandit_matches_the_live_reference():
  """GPU: mnist/0 on the REAL dataset, 4096 lanes, through `load_from_id` on both sides — the unmodified reference (its
  hard-wired /tmp/mnist pointed at the real files, one parse shared by the lanes) lane by lane with its RandomState replaced
  by the replay of the engine's draw stream, against the batched engine: every TimeStep field and total_regret bit for bit."""
  import functools
  import torch
  import bsuite_amd
  from oracle import replay
  d = real_mnist_dir()
  if d is None:
    pytest.skip('real MNIST files are not on this machine')
  if not replay.reference_available():
    pytest.skip('no reference on this box')
  bs = replay.import_reference()
  from bsuite.utils import datasets as ref_datasets
  B, T, seed = 4096, 6, 20260930
  original = ref_datasets.load_mnist
  ref_datasets.load_mnist = functools.lru_cache(maxsize=1)(lambda directory=d: original(directory))
  try:
    refs = [bs.load_from_id('mnist/0') for _ in range(B)]
  finally:
    ref_datasets.load_mnist = original
  rngs = [replay.attach_replay(e, seed, lane) for lane, e in enumerate(refs)]
  env = bsuite_amd.load_from_id('mnist/0', batch=B, seed=seed, data_dir=d, num_buffers=1)
  rnd = np.random.default_rng(3)
  for t in range(T):
    a = rnd.integers(0, 10, size=B).astype(np.int32)
    ts = env.step(torch.from_numpy(a).cuda())
    st, rw, dc, ob = (x.cpu().numpy() for x in (ts.step_type, ts.reward, ts.discount, ts.observation))
    for lane, e in enumerate(refs):
      for r in rngs[lane]:
        r.begin_step(t)
      want = e.step(int(a[lane]))
      assert int(want.step_type) == st[lane], (t, lane)
      assert np.array_equal(np.asarray(want.observation), ob[lane]), (t, lane)
      if not want.first():
        assert np.float32(want.reward) == rw[lane] and np.float32(want.discount) == dc[lane], (t, lane)
  got = env.bsuite_info()['total_regret'].cpu().numpy()
  assert np.array_equal(got, np.array([e.bsuite_info()['total_regret'] for e in refs], np.float64))


# ------------------------------------------------------------------------------------------------------ scikit-image
def test_image_adapter_fixtures_against_a_real_skimage():
  """tests/golden/image_adapter.npz was written by the reference's own `to_image` over the stand-in `resize`; with a real
  scikit-image the same observations go through `skimage.transform.resize(obs, shape[:2], preserve_range=True)` and must give
  the stored images bit for bit (the interpolated cases: more than four elements, wrappers.py:207-219)."""
  tr = real_skimage()
  if tr is None:
    pytest.skip('scikit-image is not installed: `resize` stays pinned on its published algorithm over scipy.ndimage')
  checked = 0
  for name, shape, obs, image in gu.image_adapter_cases():
    for o, want in zip(obs, image):
      if o.size <= 4:
        continue                                   # hand-crafted tiling, no resize (wrappers.py:166-204)
      plane = tr.resize(o if o.ndim == 2 else o[None], tuple(shape[:2]), preserve_range=True)
      got = np.empty(shape, dtype=o.dtype)
      got[:, :] = plane.reshape(plane.shape + (1,) * (len(shape) - 2))
      np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg=name)
      checked += 1
  assert checked > 0
