"""HBM traffic per launch from rocprofv3 PMC counters -> profiles/rNN/<workload>_pmc_traffic.json.

Runs on the GPU box.  WRITE_SIZE and FETCH_SIZE are collected in SEPARATE rocprofv3 passes (no trace
domains besides --kernel-trace), per /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KiB;
WRITE_SIZE is exact for 16-byte stores (calibrated on the fill kernel, whose byte count is known);
gfx950 reports half of the coalesced read bytes in FETCH_SIZE, so it is doubled.

  python tools/pmc_traffic.py <workload> <out.json> <algorithmic bytes per launch> <kernel substring>... -- <bench args>
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, bench_args):
  out = tempfile.mkdtemp(prefix=f'pmc_{counter}_', dir='/tmp')
  cmd = ['timeout', '300', 'rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '--',
         sys.executable, os.path.join(ROOT, 'bench.py')] + bench_args
  env = dict(os.environ, TMPDIR='/tmp')
  subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
  files = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
  per = collections.defaultdict(list)
  for f in files:
    for r in csv.DictReader(open(f)):
      if r['Counter_Name'] == counter:
        per[r['Kernel_Name']].append(float(r['Counter_Value']))
  return {k: dict(n=len(v), mean_KiB=sum(v) / len(v), min_KiB=min(v), max_KiB=max(v)) for k, v in per.items()}


def main():
  sep = sys.argv.index('--')
  workload, out_path, algorithmic, wanted = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4:sep]
  bench_args = sys.argv[sep + 1:]
  counters = {c: one_pass(c, bench_args) for c in ('WRITE_SIZE', 'FETCH_SIZE')}
  fill = [k for k in counters['WRITE_SIZE'] if 'calib_fill' in k]
  calib = None
  if fill:
    got = counters['WRITE_SIZE'][fill[0]]['mean_KiB'] * 1024
    calib = dict(kernel=fill[0], known_bytes=float(1 << 31), counter_bytes=got, factor=(1 << 31) / got)
  write = fetch = 0.0
  used = []
  for k in counters['WRITE_SIZE']:
    if any(w in k for w in wanted):
      used.append(k)
      write += counters['WRITE_SIZE'][k]['mean_KiB'] * 1024
      fetch += 2 * counters['FETCH_SIZE'].get(k, dict(mean_KiB=0.0))['mean_KiB'] * 1024
  doc = dict(
      command='rocprofv3 --pmc WRITE_SIZE|FETCH_SIZE --kernel-trace --output-format csv -- python bench.py '
              + ' '.join(bench_args) + '  (separate passes, tools/pmc_traffic.py)',
      workload=workload, kernels=used, counters=counters, write_size_calibration=calib,
      fetch_note='FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of coalesced reads)',
      per_launch=dict(note='sum over the kernels of one step() call', write_bytes=write, fetch_bytes_x2=fetch,
                      hbm_bytes=write + fetch, algorithmic_bytes=algorithmic,
                      ratio=(write + fetch) / algorithmic if algorithmic else None))
  os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
  with open(out_path, 'w') as f:
    json.dump(doc, f, indent=1)
  print(json.dumps(doc['per_launch']), 'kernels:', used)


if __name__ == '__main__':
  main()
