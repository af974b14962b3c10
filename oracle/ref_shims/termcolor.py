"""ORACLE shim for `termcolor` (bsuite/bsuite.py:50): printing is irrelevant to parity."""


def cprint(*a, **k):
  pass


def colored(text, *a, **k):
  return text
