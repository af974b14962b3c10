"""ORACLE shim for the third-party `immutabledict` package (bsuite/sweep.py:62)."""


class immutabledict(dict):

  def _ro(self, *a, **k):
    raise TypeError('immutabledict is read-only')

  __setitem__ = __delitem__ = clear = pop = popitem = setdefault = update = _ro
