"""The boundary is a C ABI: a plain-C99 program (tests/csrc/abi_host_demo.c — no C++, no Python, no
torch) includes include/bsuite_amd.h, links libbsuite_amd.so and the HIP runtime, and checks
deep_sea's known answers.  CPU: it compiles and links with gcc -std=c99.  GPU: it runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'csrc', 'abi_host_demo.c')
LIB_DIR = os.path.join(ROOT, 'bsuite_amd', '_lib')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def _build(tmp_path):
  from bsuite_amd import build
  build.build()
  exe = str(tmp_path / 'abi_host_demo')
  cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', f'-I{ROCM}/include',
         f'-I{ROOT}/include', SRC, '-o', exe, f'-L{LIB_DIR}', '-lbsuite_amd', f'-L{ROCM}/lib', '-lamdhip64',
         f'-Wl,-rpath,{LIB_DIR}', f'-Wl,-rpath,{ROCM}/lib', '-lm']
  subprocess.run(cmd, check=True, capture_output=True, text=True)
  return exe


def test_plain_c_program_compiles_and_links(tmp_path):
  exe = _build(tmp_path)
  assert os.path.exists(exe)
  syms = subprocess.run(['nm', '-D', '--undefined-only', exe], check=True, capture_output=True, text=True).stdout
  assert 'bsx_deep_sea_step' in syms and 'bsx_abi_version' in syms and 'bsx_strerror' in syms


@pytest.mark.gpu
def test_plain_c_program_runs(tmp_path):
  exe = _build(tmp_path)
  r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
  assert r.returncode == 0, r.stdout + r.stderr
  assert 'abi_host_demo: ok' in r.stdout
