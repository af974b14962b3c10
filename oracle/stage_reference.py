"""ORACLE (test infrastructure): stage the UNMODIFIED reference for the GPU box.

    python -m oracle.stage_reference            (also run by __graft_entry__.build())

`/root/reference` exists only in the build container.  The reference is pure Python, so "building it
from the sources where they lie" is byte-compiling them: every non-test module under
`/root/reference/bsuite/` is compiled by CPython's own `py_compile` into a SOURCELESS `.pyc` tree
under `oracle/_ref/bsuite/` (legacy layout, `foo.pyc` next to where `foo.py` would be, which the
import system loads without the source).  No reference source text enters the repository:
`oracle/_ref/` is git-ignored (it is NOT gpurun-ignored, so it travels to the GPU box exactly like
the built `.so` files) and holds build outputs only, plus a manifest with the sha256 of every source
it was compiled from.

Consumers — all of them checkers, never the product (bsuite_amd/ imports nothing under oracle/):
  * tests/test_gpu_vs_reference_live.py : engine vs the reference itself, live on the GPU box;
  * bench.py's `cpu_baseline` leg      : the reference's numpy step timed on the GPU box's host cores;
  * oracle/replay.import_reference()   : falls back to this tree where /root/reference is absent.
"""
import hashlib
import json
import os
import py_compile
import shutil
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get('BSX_REFERENCE_ROOT', '/root/reference')
STAGE_DIR = os.path.join(_HERE, '_ref')
MANIFEST = os.path.join(STAGE_DIR, 'MANIFEST.json')
PACKAGE = 'bsuite'
# baselines other than the random agent / run loop / pool need jax, tf, … and are not on the path
_SKIP_DIRS = ('baselines/jax', 'baselines/tf', 'baselines/third_party', 'scripts', 'tests', 'analysis')


def _sources():
  top = os.path.join(REFERENCE_ROOT, PACKAGE)
  out = []
  for d, _, files in os.walk(top):
    rel_d = os.path.relpath(d, top)
    if any(rel_d == s or rel_d.startswith(s + '/') for s in _SKIP_DIRS):
      continue
    for f in files:
      if f.endswith('.py') and not f.endswith('_test.py'):
        out.append(os.path.normpath(os.path.join(rel_d, f)))
  return sorted(out)


def _sha(path):
  with open(path, 'rb') as f:
    return hashlib.sha256(f.read()).hexdigest()


def staged():
  """True when oracle/_ref holds a tree compiled by THIS interpreter version."""
  try:
    with open(MANIFEST) as f:
      m = json.load(f)
  except (OSError, ValueError):
    return False
  return m.get('python') == list(sys.version_info[:2]) and os.path.exists(
      os.path.join(STAGE_DIR, PACKAGE, '__init__.pyc'))


def reference_present():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, PACKAGE))


def stage(verbose=False):
  """Compile the reference into oracle/_ref.  No-op (returns False) where /root/reference is absent."""
  if not reference_present():
    return False
  top = os.path.join(REFERENCE_ROOT, PACKAGE)
  srcs = _sources()
  want = {s: _sha(os.path.join(top, s)) for s in srcs}
  if staged():
    with open(MANIFEST) as f:
      if json.load(f).get('sources') == want:
        return True
  tmp = STAGE_DIR + f'.{os.getpid()}.tmp'
  shutil.rmtree(tmp, ignore_errors=True)
  for s in srcs:
    dst = os.path.join(tmp, PACKAGE, s[:-3] + '.pyc')
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    # dfile: the path tracebacks show; UNCHECKED_HASH: never looks for the (absent) source
    py_compile.compile(os.path.join(top, s), cfile=dst, dfile=os.path.join('reference', PACKAGE, s), doraise=True,
                       invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
  with open(os.path.join(tmp, 'MANIFEST.json'), 'w') as f:
    json.dump(dict(what='sourceless byte-code of the unmodified reference package (py_compile); test infrastructure',
                   reference_root=REFERENCE_ROOT, python=list(sys.version_info[:2]), sources=want), f, indent=1, sort_keys=True)
  shutil.rmtree(STAGE_DIR, ignore_errors=True)
  os.replace(tmp, STAGE_DIR)
  if verbose:
    print(f'staged {len(srcs)} reference modules as byte-code under {os.path.relpath(STAGE_DIR)}')
  return True


if __name__ == '__main__':
  if not stage(verbose=True):
    raise SystemExit(f'{REFERENCE_ROOT}/{PACKAGE} not present: nothing staged')
