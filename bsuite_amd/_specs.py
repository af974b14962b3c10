"""dm_env.specs equivalent (Array, BoundedArray, DiscreteArray) used when dm_env is absent."""
import numpy as np


class Array:
  """Shape + dtype description of an array-like value."""

  def __init__(self, shape, dtype, name=None):
    self._shape = tuple(int(d) for d in shape)
    self._dtype = np.dtype(dtype)
    self._name = name

  @property
  def shape(self):
    return self._shape

  @property
  def dtype(self):
    return self._dtype

  @property
  def name(self):
    return self._name

  def validate(self, value):
    value = np.asarray(value)
    if value.shape != self._shape:
      raise ValueError(f'Expected shape {self._shape} but found {value.shape} ({self._name})')
    if value.dtype != self._dtype:
      raise ValueError(f'Expected dtype {self._dtype} but found {value.dtype} ({self._name})')
    return value

  def generate_value(self):
    return np.zeros(shape=self._shape, dtype=self._dtype)

  def __repr__(self):
    return f'{type(self).__name__}(shape={self._shape}, dtype={self._dtype!r}, name={self._name!r})'


class BoundedArray(Array):
  """An Array with inclusive element-wise bounds."""

  def __init__(self, shape, dtype, minimum, maximum, name=None):
    super().__init__(shape, dtype, name)
    self._minimum = np.array(minimum, dtype=self._dtype)
    self._maximum = np.array(maximum, dtype=self._dtype)
    if (self._minimum > self._maximum).any():
      raise ValueError('minimum must not exceed maximum')

  @property
  def minimum(self):
    return self._minimum

  @property
  def maximum(self):
    return self._maximum

  def validate(self, value):
    value = super().validate(value)
    if (value < self._minimum).any() or (value > self._maximum).any():
      raise ValueError(f'Value out of bounds for {self._name}')
    return value

  def generate_value(self):
    return (np.ones(shape=self.shape, dtype=self.dtype) * self.dtype.type(self.minimum))


class DiscreteArray(BoundedArray):
  """A scalar integer in [0, num_values)."""

  def __init__(self, num_values, dtype=np.int32, name=None):
    if num_values <= 0 or not np.issubdtype(type(num_values), np.integer):
      raise ValueError(f'`num_values` must be a positive integer, got {num_values}')
    if not np.issubdtype(dtype, np.integer):
      raise ValueError(f'`dtype` must be integral, got {dtype}')
    super().__init__(shape=(), dtype=dtype, minimum=0, maximum=num_values - 1, name=name)
    self._num_values = int(num_values)

  @property
  def num_values(self):
    return self._num_values
