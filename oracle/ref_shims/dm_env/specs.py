"""ORACLE shim for dm_env.specs (Array / BoundedArray / DiscreteArray)."""
import numpy as np


class Array:

  def __init__(self, shape, dtype, name=None):
    self._shape = tuple(int(d) for d in shape)
    self._dtype = np.dtype(dtype)
    self._name = name

  shape = property(lambda self: self._shape)
  dtype = property(lambda self: self._dtype)
  name = property(lambda self: self._name)

  def validate(self, value):
    value = np.asarray(value)
    if value.shape != self._shape:
      raise ValueError(f'{self._name}: shape {value.shape} != {self._shape}')
    if value.dtype != self._dtype:
      raise ValueError(f'{self._name}: dtype {value.dtype} != {self._dtype}')
    return value

  def generate_value(self):
    return np.zeros(self._shape, self._dtype)

  def __repr__(self):
    return f'Array(shape={self._shape}, dtype={self._dtype}, name={self._name!r})'


class BoundedArray(Array):

  def __init__(self, shape, dtype, minimum, maximum, name=None):
    super().__init__(shape, dtype, name)
    self._minimum = np.array(minimum, dtype=self._dtype)
    self._maximum = np.array(maximum, dtype=self._dtype)

  minimum = property(lambda self: self._minimum)
  maximum = property(lambda self: self._maximum)

  def validate(self, value):
    value = super().validate(value)
    if (value < self._minimum).any() or (value > self._maximum).any():
      raise ValueError(f'{self._name}: value out of bounds')
    return value

  def generate_value(self):
    return (np.ones(self._shape, self._dtype) * self._dtype.type(self._minimum))


class DiscreteArray(BoundedArray):

  def __init__(self, num_values, dtype=np.int32, name=None):
    if num_values <= 0:
      raise ValueError('num_values must be positive')
    super().__init__((), dtype, 0, num_values - 1, name)
    self._num_values = int(num_values)

  num_values = property(lambda self: self._num_values)
