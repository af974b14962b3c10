"""Dynamic instruction counts per wavefront of a workload's kernels -> JSON (rocprofv3 PMC, own pass).

  python tools/pmc_insts.py <out.json> <kernel substring> -- <bench args>
Counters: SQ_WAVES, SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS, SQ_INSTS_VMEM_RD, SQ_INSTS_VMEM_WR (one rocprofv3
run with --kernel-trace only, as gpurun requires)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR']


def main():
  sep = sys.argv.index('--')
  out_path, wanted, bench_args = sys.argv[1], sys.argv[2:sep], sys.argv[sep + 1:]
  out = tempfile.mkdtemp(prefix='pmc_insts_', dir='/tmp')
  cmd = ['timeout', '300', 'rocprofv3', '--pmc'] + COUNTERS + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--',
         sys.executable, os.path.join(ROOT, 'bench.py')] + bench_args
  subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL,
                 stderr=subprocess.DEVNULL, check=False)
  per = collections.defaultdict(lambda: collections.defaultdict(list))
  for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
      if any(w in r['Kernel_Name'] for w in wanted):
        per[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
  doc = dict(command='rocprofv3 --pmc ' + ' '.join(COUNTERS) + ' --kernel-trace -- python bench.py ' + ' '.join(bench_args), kernels={})
  for k, c in per.items():
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    waves = mean.get('SQ_WAVES', 0) or 1
    doc['kernels'][k] = dict(launches=len(c.get('SQ_WAVES', [])), mean=mean,
                             per_wave={n: v / waves for n, v in mean.items() if n != 'SQ_WAVES'})
  os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
  with open(out_path, 'w') as f:
    json.dump(doc, f, indent=1)
  for k, v in doc['kernels'].items():
    print(k[:70], json.dumps({n: round(x, 1) for n, x in v['per_wave'].items()}))


if __name__ == '__main__':
  main()
