"""The `gym.spaces` attributes bsuite/utils/gym_wrapper.py reads (Discrete.n, Box.low/high/shape/dtype)."""
import numpy as np


class Discrete:
  def __init__(self, n):
    self.n = n
    self.shape = ()
    self.dtype = np.dtype(np.int64)


class Box:
  def __init__(self, low, high, shape=None, dtype=np.float32):
    self.shape = tuple(shape)
    self.dtype = np.dtype(dtype)
    self.low = np.full(self.shape, low, dtype=self.dtype)
    self.high = np.full(self.shape, high, dtype=self.dtype)


class MultiBinary:
  pass


class MultiDiscrete:
  pass


class Tuple:
  pass


class Dict:
  pass
