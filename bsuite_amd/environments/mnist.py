"""Batched MNIST contextual bandit (counterpart of bsuite/environments/mnist.py; csrc/mnist.hip)."""
import ctypes
import warnings
import weakref
from typing import Optional

import numpy as np
import torch

from bsuite_amd import _native
from bsuite_amd.environments import base
from bsuite_amd.utils import datasets

NUM_EPISODES = 10000  # bsuite/experiments/mnist/sweep.py:19


_DEVICE_TABLES = {}   # key -> (weakref images_dev, weakref labels_dev, host array kept alive for the key)


class MNISTBandit(base.Environment):
  """MNIST classification as a bandit environment (mnist.py:33-59).

  `images` (int8 [n, rows, cols], parsed like the reference) / `labels` may be passed directly;
  otherwise the idx files are read from `data_dir` (default: the reference's /tmp/mnist).
  """

  _info_keys = ('total_regret',)

  def __init__(self, fraction: float = 1., seed: Optional[int] = None, *, data_dir='/tmp/mnist',
               images=None, labels=None, **engine_kwargs):
    if images is None:
      (images, labels), _ = datasets.load_mnist(data_dir)
    images = np.asarray(images)
    if images.dtype != np.int8:
      raise TypeError('images must be int8, as bsuite/utils/datasets.py:55-56 parses them')
    labels = np.asarray(labels, np.uint8)
    num_data = len(labels)
    self._num_data = int(fraction * num_data)
    self._image_shape = tuple(images.shape[1:])
    n_pix = int(np.prod(self._image_shape))
    if self._num_data < 1 or self._num_data > (1 << 24) or n_pix % 4 or n_pix > 4096:
      raise ValueError('unsupported dataset geometry')
    super().__init__(obs_shape=self._image_shape, num_actions=10, seed=seed, **engine_kwargs)
    self._images = images[:self._num_data]
    self._labels = labels[:self._num_data]
    self._optimal_return = 1.
    cfg = _native.MnistCfg()
    cfg.num_data, cfg.num_pixels = self._num_data, n_pix
    # byte b of the file -> np.int8 value -> astype(float32) / 255 (mnist.py:64), by numpy itself
    lut = (np.arange(256, dtype=np.uint8).view(np.int8).astype(np.float32) / 255)
    assert lut.dtype == np.float32
    for b in range(256):
      cfg.pixel_lut[b] = float(lut[b])
    self._cfg = cfg
    self.bsuite_num_episodes = NUM_EPISODES

  def _state_tensors(self):
    # One device copy of the image table per (host array, device): environments built on the same
    # dataset (a sweep has 60) share it — real MNIST is 47 MB per copy.
    key = (self._images.__array_interface__['data'][0], self._images.shape,
           self._labels.__array_interface__['data'][0], str(self._device))
    hit = _DEVICE_TABLES.get(key)
    if hit is None or hit[0]() is None:
      host = np.ascontiguousarray(self._images).reshape(self._num_data, -1)
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')            # read-only numpy view -> tensor (never written)
        images_dev = torch.from_numpy(host).to(self._device)
        labels_dev = torch.from_numpy(np.ascontiguousarray(self._labels)).to(self._device)
      _DEVICE_TABLES[key] = hit = (weakref.ref(images_dev), weakref.ref(labels_dev), self._images)
      self._images_dev, self._labels_dev = images_dev, labels_dev
    else:
      self._images_dev, self._labels_dev = hit[0](), hit[1]()
    self._cfg.images = self._images_dev.data_ptr()
    self._cfg.labels = self._labels_dev.data_ptr()
    return dict(state=torch.full((self._batch,), 1 << 28, dtype=torch.int32, device=self._device))

  _abi_name = 'mnist'

  def _native_args(self, call, action_ptr, out):
    return (ctypes.byref(self._cfg), ctypes.byref(call), action_ptr, self._state['state'].data_ptr(), out, self._info.data_ptr())
