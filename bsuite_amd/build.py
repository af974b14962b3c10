"""Build bsuite_amd/_lib/libbsuite_amd.so (hand-written HIP for gfx950 + the C ABI) with hipcc.

    python -m bsuite_amd.build [--force] [--tuning]

hipcc cross-compiles for gfx950 without a GPU.  The .so is built in-tree so that it travels with
the repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).

--tuning additionally builds libbsuite_amd_tuning.so: the same sources compiled with -DBSX_TUNING, the
only build in which the A/B knobs of DESIGN §8 (BSX_STREAM_K, BSX_PIPELINED_PLACE, ...) are read from
the environment.  The product library reads no environment variable; the A/B scripts under tools/ and the
tests of the non-default settings load the tuning build through BSX_NATIVE_LIB.
"""
import concurrent.futures
import contextlib
import fcntl
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, '_lib')
SO_PATH = os.path.join(LIB_DIR, 'libbsuite_amd.so')
TUNING_SO_PATH = os.path.join(LIB_DIR, 'libbsuite_amd_tuning.so')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')

# -ffp-contract=off: rewards / draws are specified as sequences of IEEE ops without fused
# multiply-add so that the device agrees bit-for-bit with the reference's f64 arithmetic.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-fno-fast-math', '-Wall', '-Wno-unused-function']


def sources():
  return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps():
  hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
  hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith('.h')]
  return hdrs


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj, extra=()):
  tmp = f'{obj}.{os.getpid()}.tmp'
  subprocess.check_call(['hipcc'] + FLAGS + list(extra) + ['-c', src, '-o', tmp])
  os.replace(tmp, obj)                      # atomic: a concurrent reader never sees a half-written file
  return obj


@contextlib.contextmanager
def _locked():
  """One builder at a time: several ranks / helper processes import the package concurrently."""
  os.makedirs(LIB_DIR, exist_ok=True)
  with open(os.path.join(LIB_DIR, '.build.lock'), 'w') as f:
    fcntl.flock(f, fcntl.LOCK_EX)
    try:
      yield
    finally:
      fcntl.flock(f, fcntl.LOCK_UN)


HASH_PATH = os.path.join(LIB_DIR, '.build_hash')


def _digest(paths):
  import hashlib  # pylint: disable=import-outside-toplevel
  h = hashlib.sha256(' '.join(FLAGS).encode())
  for p in sorted(paths):
    h.update(os.path.basename(p).encode())
    with open(p, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()


def source_hashes():
  """{object file name: sha256 over the flags, its source and every header}."""
  hdrs = _deps()
  return {os.path.basename(s)[:-4] + '.o': _digest([s] + hdrs) for s in sources()}


def _variant(tuning):
  """(library path, object directory, hash file, extra flags) of the product / the tuning build."""
  if tuning:
    return TUNING_SO_PATH, os.path.join(LIB_DIR, 'tuning'), HASH_PATH + '_tuning', ('-DBSX_TUNING',)
  return SO_PATH, LIB_DIR, HASH_PATH, ()


def _recorded(tuning=False):
  import json  # pylint: disable=import-outside-toplevel
  try:
    with open(_variant(tuning)[2]) as f:
      return json.load(f)
  except (OSError, ValueError):
    return {}


def needs_build(tuning=False):
  """True when the library does not match the sources.  Decided by CONTENT (hashes recorded next to
  the library at build time), not by mtimes: a snapshot copied to another machine may carry any
  timestamps, and must neither rebuild needlessly nor run stale kernels."""
  return not os.path.exists(_variant(tuning)[0]) or _recorded(tuning) != source_hashes()


def build(force=False, verbose=False, tuning=False):
  if not force and not needs_build(tuning):  # the common import: nothing to do, no lock traffic
    return _variant(tuning)[0]
  with _locked():
    return _build_locked(force, verbose, (tuning,))[0]


def build_all(force=False, verbose=False):
  """The product library AND the tuning build, their translation units compiled in ONE pool (a clean build of both is two
  rounds of ~14 hipcc runs; side by side the cores never wait for one variant's slowest file).  Returns both paths."""
  if not force and not needs_build(False) and not needs_build(True):
    return _variant(False)[0], _variant(True)[0]
  with _locked():
    return tuple(_build_locked(force, verbose, (False, True)))


def _build_locked(force, verbose, variants):
  import json  # pylint: disable=import-outside-toplevel
  want = source_hashes()
  plans, jobs = [], []
  for tuning in variants:
    so_path, obj_dir, hash_path, extra = _variant(tuning)
    have = _recorded(tuning)
    if not force and os.path.exists(so_path) and want == have:
      plans.append((so_path, None, None))      # another process built it while we waited for the lock
      continue
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    for s in sources():
      name = os.path.basename(s)[:-4] + '.o'
      o = os.path.join(obj_dir, name)
      objs.append(o)
      if force or not os.path.exists(o) or have.get(name) != want[name]:
        jobs.append((s, o, extra))
    plans.append((so_path, objs, hash_path))
  if jobs:
    if verbose:
      print('hipcc:', ' '.join(os.path.basename(s) + (' (-DBSX_TUNING)' if extra else '') for s, _, extra in jobs), flush=True)
    # the slowest files first (sweep_mixed.hip holds every family's body, the physics families the most instantiations)
    slow = ('sweep_mixed', 'cartpole', 'umbrella_chain', 'memory_chain', 'mountain_car', 'deep_sea')
    jobs.sort(key=lambda j: next((i for i, n in enumerate(slow) if os.path.basename(j[0]).startswith(n)), len(slow)))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(jobs))) as ex:
      list(ex.map(lambda j: _compile(j[0], j[1], extra=j[2]), jobs))
  for so_path, objs, hash_path in plans:
    if objs is None:
      continue
    tmp = f'{so_path}.{os.getpid()}.tmp'
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs)
    os.replace(tmp, so_path)
    # objects of sources that no longer exist (a file split or renamed) must not linger next to the library
    keep = {os.path.basename(o) for o in objs}
    for f in os.listdir(os.path.dirname(objs[0])):
      if f.endswith('.o') and f not in keep:
        os.remove(os.path.join(os.path.dirname(objs[0]), f))
    with open(hash_path + '.tmp', 'w') as f:
      json.dump(want, f)
    os.replace(hash_path + '.tmp', hash_path)
  return [p[0] for p in plans]


if __name__ == '__main__':
  if '--tuning' in sys.argv:
    print(*build_all(force='--force' in sys.argv, verbose=True), sep='\n')
  else:
    print(build(force='--force' in sys.argv, verbose=True))
