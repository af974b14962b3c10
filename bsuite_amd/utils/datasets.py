"""MNIST idx-file loader (counterpart of bsuite/utils/datasets.py:42-69, minus the download).

Parses the four `*-ubyte.gz` files exactly as the reference does — including its quirk of reading
the image bytes as **int8** (datasets.py:55-56), which makes bright pixels negative before the
`/255` in the environment.  There is no network on the target systems, so nothing is downloaded:
the files must already be in `directory` (the reference's default `/tmp/mnist`).
"""
import gzip
import os
import struct

import numpy as np

FILES = ('train-images-idx3-ubyte.gz', 'train-labels-idx1-ubyte.gz',
         't10k-images-idx3-ubyte.gz', 't10k-labels-idx1-ubyte.gz')


def _labels(path):
  with gzip.open(path, 'rb') as fh:
    struct.unpack('>II', fh.read(8))
    return np.frombuffer(fh.read(), dtype=np.uint8).copy()


def _images(path):
  with gzip.open(path, 'rb') as fh:
    _, num_data, rows, cols = struct.unpack('>IIII', fh.read(16))
    return np.frombuffer(fh.read(), dtype=np.int8).reshape((num_data, rows, cols)).copy()


def load_mnist(directory='/tmp/mnist'):
  """Returns ((train_images int8, train_labels uint8), (test_images, test_labels))."""
  missing = [f for f in FILES if not os.path.isfile(os.path.join(directory, f))]
  if missing:
    raise FileNotFoundError(
        f'MNIST files {missing} not found in {directory}; the reference would download them '
        '(bsuite/utils/datasets.py:58-60) but this system has no network — place the idx .gz files '
        'there first.')
  p = lambda f: os.path.join(directory, f)  # noqa: E731
  # One parse per (directory, file stamps): a sweep builds 60 mnist environments on the same files.
  key = (os.path.abspath(directory),) + tuple((os.path.getmtime(p(f)), os.path.getsize(p(f))) for f in FILES)
  if key not in _PARSED:
    _PARSED.clear()
    out = ((_images(p(FILES[0])), _labels(p(FILES[1]))), (_images(p(FILES[2])), _labels(p(FILES[3]))))
    for pair in out:
      for arr in pair:
        arr.setflags(write=False)
    _PARSED[key] = out
  return _PARSED[key]


_PARSED = {}


def write_idx_files(directory, train_images_u8, train_labels, test_images_u8=None, test_labels=None):
  """Writes idx .gz files in the MNIST wire format (used to stage synthetic data offline)."""
  os.makedirs(directory, exist_ok=True)
  if test_images_u8 is None:
    test_images_u8, test_labels = train_images_u8[:1], train_labels[:1]
  for name, arr in ((FILES[0], train_images_u8), (FILES[2], test_images_u8)):
    arr = np.ascontiguousarray(arr, np.uint8)
    with gzip.open(os.path.join(directory, name), 'wb') as fh:
      fh.write(struct.pack('>IIII', 2051, arr.shape[0], arr.shape[1], arr.shape[2]))
      fh.write(arr.tobytes())
  for name, arr in ((FILES[1], train_labels), (FILES[3], test_labels)):
    arr = np.ascontiguousarray(arr, np.uint8)
    with gzip.open(os.path.join(directory, name), 'wb') as fh:
      fh.write(struct.pack('>II', 2049, arr.shape[0]))
      fh.write(arr.tobytes())
