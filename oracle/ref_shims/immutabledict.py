"""ORACLE shim for the third-party `immutabledict` package (bsuite/sweep.py:62): a read-only dict
that still copies/deep-copies like the real one (bsuite/logging/logging_utils.py:31 deep-copies
sweep.SETTINGS and then indexes into the copy's *values*, which are plain dicts)."""
import copy


class immutabledict(dict):

  def _ro(self, *a, **k):
    raise TypeError('immutabledict is read-only')

  __setitem__ = __delitem__ = clear = pop = popitem = setdefault = update = _ro

  def __copy__(self):
    return self

  def __deepcopy__(self, memo):
    return immutabledict({copy.deepcopy(k, memo): copy.deepcopy(v, memo) for k, v in self.items()})

  def __reduce__(self):
    return (immutabledict, (dict(self),))
