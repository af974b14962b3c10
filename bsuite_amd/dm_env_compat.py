"""The slice of the third-party `dm_env` API that bsuite's environment path touches.

bsuite's boundary is the dm_env protocol (bsuite/environments/base.py:34-77 subclasses
dm_env.Environment and every `_step/_reset` returns dm_env.TimeStep via restart / transition /
termination).  dm_env is not part of the reference tree (setup.py:82) and is not installed here, so
the engine carries this small equivalent; when the real package is importable it is used instead,
so TimeSteps interoperate with user code that imports dm_env.
"""
# pylint: disable=unused-import,g-import-not-at-top
try:
  import dm_env as _real
  from dm_env import specs
  StepType = _real.StepType
  TimeStep = _real.TimeStep
  restart, transition = _real.restart, _real.transition
  termination, truncation = _real.termination, _real.truncation
  EnvironmentBase = _real.Environment
  HAVE_DM_ENV = True
except ImportError:
  import abc
  import enum
  from typing import Any, NamedTuple

  from bsuite_amd import _specs as specs

  HAVE_DM_ENV = False

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

  class EnvironmentBase(abc.ABC):
    """dm_env.Environment: abstract reset/step/specs + default reward/discount specs."""

    @abc.abstractmethod
    def reset(self):
      """Starts a new episode and returns the first TimeStep."""

    @abc.abstractmethod
    def step(self, action):
      """Advances the environment by one step."""

    @abc.abstractmethod
    def observation_spec(self):
      """Spec of the observation."""

    @abc.abstractmethod
    def action_spec(self):
      """Spec of the action."""

    def reward_spec(self):
      return specs.Array(shape=(), dtype=float, name='reward')

    def discount_spec(self):
      return specs.BoundedArray(shape=(), dtype=float, minimum=0., maximum=1., name='discount')

    def close(self):
      pass

    def __enter__(self):
      return self

    def __exit__(self, exc_type, exc_value, traceback):
      self.close()
