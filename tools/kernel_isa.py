"""ISA of one kernel out of a .hip translation unit, with a per-basic-block instruction census (no GPU):

  python tools/kernel_isa.py bsuite_amd/csrc/cartpole.hip 'small_obs_kernel<cartpole_env, true, 0, 0, 0, true, true>' [-o out.s] [-D...]

Compiles the file to gfx950 assembly (cached under /tmp/bsx_isa by content hash), finds the kernel by its demangled
name (substring), writes its text to -o and prints, per basic block, the number of VALU / SALU / VMEM / LDS
instructions, spill traffic (v_readlane / v_writelane) and waits — what tests/test_kernel_resources.py and
profiles/r04/*_loop_isa.txt are made from.
"""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = '/tmp/bsx_isa'


def assembly(src, defines=()):
  import importlib.util  # pylint: disable=import-outside-toplevel
  spec = importlib.util.spec_from_file_location('bsx_build', os.path.join(ROOT, 'bsuite_amd', 'build.py'))   # (not the package:
  build = importlib.util.module_from_spec(spec)                                                                #  importing it builds)
  spec.loader.exec_module(build)
  os.makedirs(CACHE, exist_ok=True)
  key = hashlib.sha256((build._digest([src] + build._deps()) + ' '.join(defines)).encode()).hexdigest()[:16]
  out = os.path.join(CACHE, f'{os.path.basename(src)}.{key}.s')
  if not os.path.exists(out):
    flags = [f for f in build.FLAGS if f not in ('-Wall',)]
    subprocess.run(['hipcc'] + flags + list(defines) + ['-S', '--cuda-device-only', '-o', out + '.tmp', src],
                   check=True, stderr=subprocess.DEVNULL)
    os.replace(out + '.tmp', out)
  return open(out).read().split('\n')


def functions(lines):
  """{demangled name: (first line, last line)} of the kernels / functions in the assembly."""
  starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
  names = subprocess.run(['c++filt'], input='\n'.join(n for _, n in starts), capture_output=True, text=True, check=True).stdout.split('\n')
  out = {}
  for (i, _), dn in zip(starts, names):
    j = i
    while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
      j += 1
    out[re.sub(r'^void ', '', dn).split('(')[0]] = (i, j)
  return out


def classify(ins):
  op = ins.split()[0]
  if op in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'):
    return op
  if op.startswith('v_'):
    return 'valu'
  if op.startswith('s_waitcnt'):
    return 'wait'
  if op.startswith('s_barrier'):
    return 'barrier'
  if op.startswith('s_cbranch') or op.startswith('s_branch'):
    return 'branch'
  if op.startswith('s_load') or op.startswith('s_buffer_load'):
    return 'smem'
  if op.startswith('s_'):
    return 'salu'
  if op.startswith('global_load') or op.startswith('flat_load') or op.startswith('buffer_load') or op.startswith('scratch_load'):
    return 'vload'
  if op.startswith('global_store') or op.startswith('flat_store') or op.startswith('buffer_store') or op.startswith('scratch_store'):
    return 'vstore'
  if op.startswith('global_atomic') or op.startswith('flat_atomic'):
    return 'vatomic'
  if op.startswith('ds_'):
    return 'lds'
  return 'other'


def census(text):
  """[(label, {class: count})] per basic block, in layout order."""
  blocks, cur, name = [], {}, 'entry'
  for l in text:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
      blocks.append((name, cur))
      cur, name = {}, m.group(1)
      continue
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
      continue
    c = classify(s)
    cur[c] = cur.get(c, 0) + 1
  blocks.append((name, cur))
  return blocks


def loop_spill_reloads(text, min_depth=2):
  """SGPR spill reloads (v_readlane_b32 sN, vM, <constant lane>) inside loops nested at least `min_depth` deep — the
  step loop of a fused rollout is the inner of two.  The block labels carry the compiler's loop comments."""
  depth, n = 0, 0
  for l in text:
    m = re.match(r'^\.LBB\d+_\d+:\s*;(.*)$', l)
    if m:
      d = re.search(r'Depth=(\d+)', m.group(1))
      depth = int(d.group(1)) if d else 0
      continue
    if re.match(r'^\.LBB\d+_\d+:', l) or l.startswith('; %bb.'):
      d = re.search(r'Depth=(\d+)', l)
      depth = int(d.group(1)) if d else 0
      continue
    if depth >= min_depth and re.search(r'v_readlane_b32 s\d+, v\d+, \d+', l):
      n += 1
  return n


def kernel_text(src, want, defines=()):
  """(demangled name, assembly lines) of the one kernel of `src` whose name contains `want`."""
  lines = assembly(os.path.abspath(src), defines)
  fns = functions(lines)
  hits = [n for n in fns if want in n]
  if len(hits) != 1:
    hits = [n for n in hits if n == want]
  if len(hits) != 1:
    raise KeyError(f'{want!r} matches {len(hits)} kernels')
  i, j = fns[hits[0]]
  return hits[0], lines[i:j]


def main():
  argv = sys.argv[1:]
  defines = [a for a in argv if a.startswith('-D')]
  argv = [a for a in argv if not a.startswith('-D')]
  out = None
  if '-o' in argv:
    out = argv[argv.index('-o') + 1]
    del argv[argv.index('-o'):argv.index('-o') + 2]
  src, want = argv[0], argv[1]
  lines = assembly(os.path.abspath(src), defines)
  fns = functions(lines)
  hits = [n for n in fns if want in n]
  if len(hits) != 1:
    exact = [n for n in hits if n == want]
    if len(exact) == 1:
      hits = exact
    else:
      print(f'{len(hits)} kernels match {want!r}:', *hits, sep='\n  ')
      sys.exit(1)
  i, j = fns[hits[0]]
  text = lines[i:j]
  regs = {}
  for l in lines[j:j + 16]:                      # the .set <symbol>.num_vgpr, N lines that follow the function
    m = re.search(r'\.(num_vgpr|numbered_sgpr|private_seg_size), (\d+)', l)
    if m:
      regs[m.group(1)] = int(m.group(2))
  if out:
    with open(out, 'w') as f:
      f.write('\n'.join(text) + '\n')
  total = {}
  print(hits[0], regs)
  cols = ('valu', 'v_readlane_b32', 'v_writelane_b32', 'salu', 'smem', 'vload', 'vstore', 'lds', 'wait', 'barrier', 'branch')
  print(f'{"block":>12} ' + ' '.join(f'{c.replace("_b32", "").replace("v_", ""):>9}' for c in cols))
  for name, cnt in census(text):
    if not cnt:
      continue
    for k, v in cnt.items():
      total[k] = total.get(k, 0) + v
    print(f'{name:>12} ' + ' '.join(f'{cnt.get(c, 0):9d}' for c in cols))
  print(f'{"total":>12} ' + ' '.join(f'{total.get(c, 0):9d}' for c in cols))
  print('SGPR spill reloads inside loops of depth >= 2:', loop_spill_reloads(text))


if __name__ == '__main__':
  main()
