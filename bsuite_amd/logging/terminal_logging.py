"""Terminal logging (counterpart of bsuite/logging/terminal_logging.py)."""
import logging as std_logging
import numbers
from typing import Any, Mapping

from bsuite_amd.logging import base
from bsuite_amd.utils import wrappers


def wrap_environment(env, pretty_print: bool = True, log_every: bool = False, log_by_step: bool = False):
  """Returns a wrapped environment that logs to terminal."""
  logger = Logger(pretty_print)
  return wrappers.Logging(env, logger, log_by_step=log_by_step, log_every=log_every)


class Logger(base.Logger):
  """Writes data to terminal."""

  def __init__(self, pretty_print: bool = True):
    self._pretty_print = pretty_print
    self._log = std_logging.getLogger('bsuite_amd')
    if not self._log.handlers:
      self._log.addHandler(std_logging.StreamHandler())
    self._log.setLevel(std_logging.INFO)

  def write(self, data: Mapping[str, Any]):
    if self._pretty_print:
      data = ' | '.join(
          f'{k} = {v:0.3f}' if isinstance(v, numbers.Real) and not isinstance(v, numbers.Integral)
          else f'{k} = {v}' for k, v in sorted(data.items()))
    self._log.info(data)
