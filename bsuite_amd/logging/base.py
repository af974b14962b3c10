"""Logger interface (counterpart of bsuite/logging/base.py): an object with a `write(dict)` method."""
import abc
from typing import Any, Mapping


class Logger(abc.ABC):
  """A logger has a `write` method."""

  @abc.abstractmethod
  def write(self, data: Mapping[str, Any]):
    """Writes `data` to destination (file, terminal, database, etc)."""
