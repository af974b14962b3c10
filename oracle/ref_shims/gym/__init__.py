"""Minimal `gym` stand-in so the reference's bsuite/utils/gym_wrapper.py can be imported to write
fixtures (oracle/make_golden.py).  TEST INFRASTRUCTURE ONLY; only what that module touches."""
from gym import spaces  # noqa: F401


class Space:
  pass


class Env:
  metadata = {}

  def close(self):
    pass
