"""ORACLE shim: minimal stand-in for the third-party `dm_env` package (not installed here, not in
/root/reference; setup.py:82 lists it unpinned) so the unmodified reference can be imported to
generate golden vectors.  Semantics restated from dm_env 1.x public API."""
import abc
import enum
from typing import Any, NamedTuple

from dm_env import specs  # noqa: F401  (re-export)


class StepType(enum.IntEnum):
  FIRST = 0
  MID = 1
  LAST = 2

  def first(self):
    return self is StepType.FIRST

  def mid(self):
    return self is StepType.MID

  def last(self):
    return self is StepType.LAST


class TimeStep(NamedTuple):
  step_type: Any
  reward: Any
  discount: Any
  observation: Any

  def first(self):
    return self.step_type == StepType.FIRST

  def mid(self):
    return self.step_type == StepType.MID

  def last(self):
    return self.step_type == StepType.LAST


def restart(observation):
  return TimeStep(StepType.FIRST, None, None, observation)


def transition(reward, observation, discount=1.0):
  return TimeStep(StepType.MID, reward, discount, observation)


def termination(reward, observation):
  return TimeStep(StepType.LAST, reward, 0.0, observation)


def truncation(reward, observation, discount=1.0):
  return TimeStep(StepType.LAST, reward, discount, observation)


class Environment(abc.ABC):

  @abc.abstractmethod
  def reset(self):
    pass

  @abc.abstractmethod
  def step(self, action):
    pass

  @abc.abstractmethod
  def observation_spec(self):
    pass

  @abc.abstractmethod
  def action_spec(self):
    pass

  def reward_spec(self):
    return specs.Array(shape=(), dtype=float, name='reward')

  def discount_spec(self):
    return specs.BoundedArray(shape=(), dtype=float, minimum=0., maximum=1., name='discount')

  def close(self):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *a):
    self.close()
