"""Builds the library with extra compiler flags into tools/ab/ (git-ignored; travels to the GPU box), for a same-call A/B of a
compile-time constant:   python tools/ab_flag_lib.py pk4 -DPAIR_MNIST_K=4   ->  tools/ab/libbsuite_amd_pk4.so
(load it through BSX_NATIVE_LIB)."""
import concurrent.futures
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsuite_amd import build  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
objdir = os.path.join(root, 'tools', 'ab', 'obj_' + name)
os.makedirs(objdir, exist_ok=True)
jobs = [(s, os.path.join(objdir, os.path.basename(s)[:-4] + '.o')) for s in build.sources()]
with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
  list(ex.map(lambda so: build._compile(*so, extra=flags), jobs))  # pylint: disable=protected-access
so = os.path.join(root, 'tools', 'ab', f'libbsuite_amd_{name}.so')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + [o for _, o in jobs])
print(so)
