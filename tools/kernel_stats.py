"""Per-kernel durations of the TIMED launches of a bench.py command, from a rocprofv3 kernel trace (run on the GPU box):

  python tools/kernel_stats.py <out.csv> [--last K] -- <bench args>

runs `rocprofv3 --kernel-trace --output-format csv -- python bench.py <bench args>` and, for each of OUR kernels
(bsx_* / small_obs* / sweep_* / pair_mixed* / mnist_*), averages the LAST K dispatches (default: the command's
--steps; with --no-also --no-cpu-baseline those are exactly the timed region — not the ~10^3 phase-stagger pre-roll
launches that rocprofv3's own --stats table mixes in).  The bench's JSON line is written next to
the csv (<out>.bench.json) so that the HIP-event time and the trace average of the same run can be compared.
"""
import collections
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ('bsx_', 'small_obs', 'sweep_', 'pair_mixed', 'mnist_')


def main():
  sep = sys.argv.index('--')
  head, bench_args = sys.argv[1:sep], sys.argv[sep + 1:]
  out_csv = head[0]
  last = int(head[head.index('--last') + 1]) if '--last' in head else int(bench_args[bench_args.index('--steps') + 1])
  out = tempfile.mkdtemp(prefix='kstats_', dir='/tmp')
  cmd = ['timeout', '900', 'rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable,
         os.path.join(ROOT, 'bench.py')] + bench_args
  p = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
  line = [l for l in p.stdout.splitlines() if l.startswith('{')]
  os.makedirs(os.path.dirname(os.path.abspath(out_csv)), exist_ok=True)
  if line:
    open(out_csv + '.bench.json', 'w').write(line[-1] + '\n')
  per = collections.defaultdict(list)
  for f in glob.glob(os.path.join(out, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
      if any(o in r['Kernel_Name'] for o in OURS):
        per[r['Kernel_Name']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']),
                                      r.get('Grid_Size', ''), r.get('VGPR_Count', ''), r.get('LDS_Block_Size', '')))
  if not per:
    raise SystemExit(f'no kernels traced: rc={p.returncode}\n{p.stderr[-2000:]}')
  with open(out_csv, 'w') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'dispatches_total', 'timed_dispatches', 'timed_avg_us', 'timed_min_us', 'timed_max_us', 'all_avg_us',
                'grid_size', 'vgpr', 'lds_bytes'])
    for k, v in sorted(per.items(), key=lambda kv: -sum(d for _, d, *_ in kv[1][-last:])):
      v.sort()
      t = [d for _, d, *_ in v[-last:]]
      a = [d for _, d, *_ in v]
      w.writerow([k.split('(')[0].replace('void ', ''), len(v), len(t), f'{sum(t) / len(t) / 1e3:.3f}', f'{min(t) / 1e3:.3f}',
                  f'{max(t) / 1e3:.3f}', f'{sum(a) / len(a) / 1e3:.3f}', v[-1][2], v[-1][3], v[-1][4]])
  print(open(out_csv).read())


if __name__ == '__main__':
  main()
