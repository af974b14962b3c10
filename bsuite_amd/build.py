"""Build bsuite_amd/_lib/libbsuite_amd.so (hand-written HIP for gfx950 + the C ABI) with hipcc.

    python -m bsuite_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is built in-tree so that it travels with
the repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, '_lib')
SO_PATH = os.path.join(LIB_DIR, 'libbsuite_amd.so')
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


def _compile(src, obj):
  subprocess.check_call(['hipcc'] + FLAGS + ['-c', src, '-o', obj])
  return obj


def build(force=False, verbose=False):
  os.makedirs(LIB_DIR, exist_ok=True)
  srcs, hdrs = sources(), _deps()
  objs, jobs = [], []
  for s in srcs:
    o = os.path.join(LIB_DIR, os.path.basename(s)[:-4] + '.o')
    objs.append(o)
    if force or _stale(o, [s] + hdrs):
      jobs.append((s, o))
  if jobs:
    if verbose:
      print('hipcc:', ' '.join(os.path.basename(s) for s, _ in jobs), flush=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
      list(ex.map(lambda so: _compile(*so), jobs))
  if jobs or force or _stale(SO_PATH, objs):
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', SO_PATH] + objs)
  return SO_PATH


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, verbose=True))
