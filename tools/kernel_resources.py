"""Register / LDS / scratch budget of every kernel in a built library, read from the code objects' metadata (no GPU):

  python tools/kernel_resources.py [libbsuite_amd.so] [--filter SUBSTR] [--json]

The .hip_fatbin section holds one clang offload bundle per translation unit; each is unbundled with
clang-offload-bundler and its NT_AMDGPU_METADATA note read with llvm-readelf.  waves_per_simd is what the VGPR count
alone allows on gfx950 (512 VGPRs per SIMD lane, allocation granule 8, at most 8 waves); LDS limits come on top
(160 KiB per CU).  tests/test_kernel_resources.py keeps the hot kernels inside their budgets.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'
FIELDS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'group_segment_fixed_size',
          'private_segment_fixed_size', 'kernarg_segment_size', 'max_flat_workgroup_size')


def demangle(names):
  p = subprocess.run(['c++filt'], input='\n'.join(names), stdout=subprocess.PIPE, text=True, check=True)
  return p.stdout.split('\n')[:len(names)]


def waves_per_simd(vgprs):
  return max(1, min(8, 512 // max(8, (vgprs + 7) // 8 * 8)))


def kernels(lib):
  """[{name (demangled), symbol, vgpr_count, ..., waves_per_simd}] for every kernel of `lib`."""
  out = []
  with tempfile.TemporaryDirectory(prefix='bsx_co_', dir='/tmp') as d:
    fat = os.path.join(d, 'fat.bin')
    subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', lib, os.path.join(d, 'copy.so')], check=True)
    data = open(fat, 'rb').read()
    starts = [m.start() for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', data)]
    for k, p in enumerate(starts):
      bundle, co = os.path.join(d, f'b{k}.bin'), os.path.join(d, f'b{k}.co')
      with open(bundle, 'wb') as f:
        f.write(data[p:starts[k + 1] if k + 1 < len(starts) else len(data)])
      subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', f'--input={bundle}',
                      f'--targets={TARGET}', f'--output={co}'], check=True)
      notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], stdout=subprocess.PIPE, text=True, check=True).stdout
      for block in notes.split('  - .agpr_count:')[1:]:
        block = '.agpr_count:' + block
        rec = {}
        for fld in FIELDS:
          m = re.search(r'\.' + fld + r':\s+(\d+)', block)
          rec[fld] = int(m.group(1)) if m else None
        rec['symbol'] = re.search(r'\.name:\s+(\S+)', block).group(1)
        out.append(rec)
  for rec, name in zip(out, demangle([r['symbol'] for r in out])):
    rec['name'] = re.sub(r'^void ', '', name)
    rec['waves_per_simd'] = waves_per_simd(rec['vgpr_count'])
  return out


def main():
  argv = sys.argv[1:]
  flt = ''
  if '--filter' in argv:
    k = argv.index('--filter')
    flt = argv[k + 1]
    del argv[k:k + 2]
  args = [a for a in argv if not a.startswith('--')]
  lib = args[0] if args else os.path.join(ROOT, 'bsuite_amd', '_lib', 'libbsuite_amd.so')
  ks = [k for k in kernels(lib) if flt in k['name']]
  if '--json' in sys.argv:
    print(json.dumps(ks, indent=1))
    return
  print(f'{len(ks)} kernels in {lib}')
  print(f'{"vgpr":>5} {"w/simd":>6} {"sgpr":>5} {"lds":>6} {"scratch":>7} {"spill":>5}  kernel')
  for k in sorted(ks, key=lambda r: r['name']):
    print(f'{k["vgpr_count"]:5d} {k["waves_per_simd"]:6d} {k["sgpr_count"]:5d} {k["group_segment_fixed_size"]:6d} '
          f'{k["private_segment_fixed_size"]:7d} {k["vgpr_spill_count"] + k["sgpr_spill_count"]:5d}  {k["name"].split("(")[0][:150]}')


if __name__ == '__main__':
  main()
