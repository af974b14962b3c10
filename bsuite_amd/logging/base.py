"""What the engine's `Logging` wrapper needs from a logger (cf. bsuite/logging/base.py): one method,
`write(row)`, called with a flat mapping of column name -> value for every logged row; an optional
`flush()` is called when the last scheduled episode has been logged."""
import abc
from typing import Any, Mapping


class Logger(abc.ABC):
  """Sink for logged rows: CSV file, terminal, anything with `write`."""

  @abc.abstractmethod
  def write(self, data: Mapping[str, Any]):
    """Receives one row."""

  @classmethod
  def __subclasshook__(cls, other):
    # duck typing: third-party loggers written against the reference interface qualify as well
    if cls is Logger and callable(getattr(other, 'write', None)):
      return True
    return NotImplemented
