"""rocprofv3 PMC passes over a bench.py command -> profiles/rNN/<name>_pmc_{traffic,sq}.json (run on the GPU box).

  python tools/pmc.py traffic <name> <out.json> --kernels SUB [SUB...] --alg-bytes N [--lanes L] [--last K] -- <bench args>
  python tools/pmc.py sq      <name> <out.json> --kernels SUB [SUB...] [--lanes L] [--last K] -- <bench args>

Every counter group is collected in its OWN rocprofv3 run with --kernel-trace only (gpurun refuses --pmc next to
other trace domains), as /opt/skills/guides/MI355X_MICROARCH.md prescribes: WRITE_SIZE and FETCH_SIZE cannot share a
pass (TCC has 4 slots: 2 + 3), both are in KiB, WRITE_SIZE is calibrated on bench.py's 2 GiB fill kernel
(`bsx_calib_fill`, known byte count), gfx950 reports half of the coalesced read bytes in FETCH_SIZE, so it is doubled.
Only OUR kernels are kept (names containing one of --kernels), and of each only its LAST K dispatches (default: the
bench command's --steps) — with --no-also --no-cpu-baseline those are exactly the timed launches, not the
phase-stagger pre-roll.  A kernel that is missing from a pass is an error, never a zero.

sq: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
in one pass, SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
GRBM_GUI_ACTIVE in a second.  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (guide, "tick vs SQ PMC
units"); valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8
XCDs when that is plausible, else duration * 2.4 GHz.
"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ_PASSES = (['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES',
              'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY'],
             ['SQ_INSTS_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_INSTS_VMEM_WR',
              'SQ_INSTS_VMEM_RD', 'GRBM_GUI_ACTIVE'])


SCRIPT = [os.path.join(ROOT, 'bench.py')]


def one_pass(counters, bench_args, wanted, last):
  """{kernel: {counter: mean over its last `last` dispatches}, '_n': dispatches used, '_dur_ns': mean duration}."""
  out = tempfile.mkdtemp(prefix='pmc_', dir='/tmp')
  cmd = ['timeout', '600', 'rocprofv3', '--pmc'] + list(counters) + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--',
         sys.executable] + SCRIPT + bench_args
  p = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                     text=True, check=False)
  rows = collections.defaultdict(lambda: collections.defaultdict(dict))          # kernel -> dispatch -> counter -> value
  dur = collections.defaultdict(dict)
  for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
      k = r['Kernel_Name']
      if any(w in k for w in wanted) or 'calib_fill' in k:
        rows[k][int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        if r.get('Start_Timestamp') and r.get('End_Timestamp'):
          dur[k][int(r['Dispatch_Id'])] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
  for f in glob.glob(os.path.join(out, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
      k = r['Kernel_Name']
      if k in rows:
        dur[k][int(r['Dispatch_Id'])] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
  res = {}
  for k, per in rows.items():
    ids = sorted(per)[-last:]
    res[k] = {c: sum(per[i].get(c, 0.0) for i in ids) / len(ids) for c in counters}
    res[k]['_n'] = len(ids)
    d = [dur[k][i] for i in ids if i in dur[k]]
    res[k]['_dur_ns'] = sum(d) / len(d) if d else None
  if not res:
    raise SystemExit(f'rocprofv3 pass {counters} produced no rows for {wanted}: rc={p.returncode}\n{p.stderr[-2000:]}')
  return res


def last_total(counter, bench_args, wanted, n):
  """SUM of `counter` (KiB) over the LAST n dispatches of the kernels whose names contain one of `wanted`, in dispatch
  order — the timed region of a multi-launch rollout (1 advance + T-1 pipelined launches + 1 stream per call) — and
  how many dispatches of each kernel that was."""
  rows = raw_pass([counter], bench_args, wanted)
  flat = sorted((i, k, v.get(counter, 0.0)) for k, per in rows.items() for i, v in per.items())
  if len(flat) < n:
    raise SystemExit(f'{len(flat)} dispatches of {wanted}, {n} expected')
  by_kernel = collections.Counter(short(k) for _, k, _ in flat[-n:])
  return sum(v for _, _, v in flat[-n:]), dict(by_kernel)


def raw_pass(counters, bench_args, wanted):
  out = tempfile.mkdtemp(prefix='pmc_', dir='/tmp')
  cmd = ['timeout', '600', 'rocprofv3', '--pmc'] + list(counters) + ['--kernel-trace', '--output-format', 'csv', '-d', out, '--',
         sys.executable] + SCRIPT + bench_args
  p = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                     text=True, check=False)
  rows = collections.defaultdict(lambda: collections.defaultdict(dict))
  for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
      k = r['Kernel_Name']
      if any(w in k for w in wanted):
        rows[k][int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
  if not rows:
    raise SystemExit(f'rocprofv3 pass {counters} produced no rows for {wanted}: rc={p.returncode}\n{p.stderr[-2000:]}')
  return rows


def short(k):
  return k.split('(')[0].replace('void ', '')[:90]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('what', choices=['traffic', 'sq'])
  ap.add_argument('name')
  ap.add_argument('out')
  ap.add_argument('--kernels', nargs='+', required=True)
  ap.add_argument('--alg-bytes', type=float, default=0.0)
  ap.add_argument('--lanes', type=int, default=1 << 20)
  ap.add_argument('--last', type=int, default=0)
  ap.add_argument('--last-total', type=int, default=0,
                  help='the LAST N dispatches of all --kernels together are the timed region (a pipelined rollout is 1 advance + '
                       'T-1 pipelined launches + 1 stream per call); the per-launch figures become per STEP: their sum / --steps-total')
  ap.add_argument('--steps-total', type=int, default=0, help='env steps the counted dispatches cover (see --last-total)')
  ap.add_argument('--steps-per-launch', type=int, default=1, help='a fused rollout launch runs T steps: per-launch bytes / T')
  ap.add_argument('--script', default=None, help='profile this python script instead of bench.py (needs --last)')
  sep = sys.argv.index('--', 2)
  args = ap.parse_args(sys.argv[1:sep])
  bench_args = sys.argv[sep + 1:]
  if args.script:
    SCRIPT[0] = os.path.abspath(args.script)
  last = args.last or int(bench_args[bench_args.index('--steps') + 1])
  doc = dict(name=args.name, lanes=args.lanes, last_dispatches_per_kernel=last,
             command='rocprofv3 --pmc <one group per run> --kernel-trace --output-format csv -- python bench.py ' + ' '.join(bench_args)
                     + '   (tools/pmc.py)')
  if args.what == 'traffic':
    passes = {c: one_pass([c], bench_args, args.kernels, last) for c in ('WRITE_SIZE', 'FETCH_SIZE')}
    fill = [k for k in passes['WRITE_SIZE'] if 'calib_fill' in k]
    calib = None
    if fill:
      got = passes['WRITE_SIZE'][fill[0]]['WRITE_SIZE'] * 1024
      calib = dict(kernel=short(fill[0]), known_bytes=float(1 << 31), counter_bytes=got, factor=(1 << 31) / got)
    kernels = sorted(k for k in passes['WRITE_SIZE'] if any(w in k for w in args.kernels))
    missing = [k for k in kernels if k not in passes['FETCH_SIZE']]
    if missing or not kernels:
      raise SystemExit(f'kernels missing from the FETCH_SIZE pass: {missing or args.kernels}')
    if args.last_total:
      # per STEP of a multi-launch rollout: everything the timed region's dispatches moved / the steps they cover
      w_kib, by_kernel = last_total('WRITE_SIZE', bench_args, args.kernels, args.last_total)
      f_kib, _ = last_total('FETCH_SIZE', bench_args, args.kernels, args.last_total)
      write, fetch = w_kib * 1024 / args.steps_total, 2 * f_kib * 1024 / args.steps_total
      doc['timed_region'] = dict(dispatches=args.last_total, steps=args.steps_total, dispatches_by_kernel=by_kernel,
                                 write_KiB_total=w_kib, fetch_KiB_total=f_kib)
    else:
      write = sum(passes['WRITE_SIZE'][k]['WRITE_SIZE'] * 1024 for k in kernels) / args.steps_per_launch
      fetch = sum(2 * passes['FETCH_SIZE'][k]['FETCH_SIZE'] * 1024 for k in kernels) / args.steps_per_launch
    doc.update(kernels={short(k): dict(write_KiB=passes['WRITE_SIZE'][k]['WRITE_SIZE'], fetch_KiB=passes['FETCH_SIZE'][k]['FETCH_SIZE'],
                                       dispatches=passes['WRITE_SIZE'][k]['_n'], dur_us=(passes['WRITE_SIZE'][k]['_dur_ns'] or 0) / 1e3)
                        for k in kernels},
               write_size_calibration=calib,
               fetch_note='FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of coalesced reads)',
               per_launch=dict(note='sum over the kernels of one step' + (f' (a launch runs {args.steps_per_launch} steps: per-launch bytes / {args.steps_per_launch})' if args.steps_per_launch > 1 else '')
                                    + (' (multi-launch rollout: the timed dispatches summed / their steps)' if args.last_total else ''), write_bytes=write, fetch_bytes_x2=fetch,
                               hbm_bytes=write + fetch, algorithmic_bytes=args.alg_bytes,
                               ratio=(write + fetch) / args.alg_bytes if args.alg_bytes else None))
  else:
    merged = collections.defaultdict(dict)
    errors = []
    for counters in SQ_PASSES:
      try:
        for k, v in one_pass(counters, bench_args, args.kernels, last).items():
          if any(w in k for w in args.kernels):
            merged[k].update(v)
      except SystemExit as e:
        errors.append(str(e)[:500])
    if not merged:
      raise SystemExit('\n'.join(errors))
    doc['kernels'] = {}
    total = collections.defaultdict(float)
    for k, v in merged.items():
      waves = v.get('SQ_WAVES') or 1.0
      dur_ns = v.get('_dur_ns') or 0.0
      cyc_grbm = (v.get('GRBM_GUI_ACTIVE') or 0.0) / 8.0
      cyc_wall = dur_ns * 2.4
      cyc = cyc_grbm if 0.3 * cyc_wall < cyc_grbm < 1.5 * cyc_wall else cyc_wall
      rec = dict(mean={c: x for c, x in v.items() if not c.startswith('_')}, dispatches=v['_n'], dur_us=dur_ns / 1e3,
                 per_wave={c: x / waves for c, x in v.items() if c.startswith('SQ_') and c != 'SQ_WAVES'},
                 kernel_cycles=cyc, kernel_cycles_source='GRBM_GUI_ACTIVE/8' if cyc == cyc_grbm else 'duration*2.4GHz')
      if cyc and 'SQ_ACTIVE_INST_VALU' in v:
        rec['valu_busy_frac'] = 4.0 * v['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc)
        rec['valu_issue_frac_2cyc'] = 2.0 * v['SQ_INSTS_VALU'] / (1024.0 * cyc)      # every VALU instruction priced at 2 cycles
      if v.get('SQ_WAVE_CYCLES'):
        rec['wave_cycle_shares'] = {c: v[c] / v['SQ_WAVE_CYCLES'] for c in ('SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
                                                                             'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS') if c in v}
        rec['mean_waves_per_simd'] = 4.0 * v['SQ_WAVE_CYCLES'] / (1024.0 * cyc) if cyc else None
      doc['kernels'][short(k)] = rec
      for c in ('SQ_INSTS_VALU', 'SQ_WAVES'):
        total[c] += v.get(c, 0.0)
    first = next(iter(doc['kernels'].values()))
    doc['per_launch'] = dict(SQ_INSTS_VALU=total['SQ_INSTS_VALU'], SQ_WAVES=total['SQ_WAVES'],
                             valu_busy_frac=first.get('valu_busy_frac'), note='per kernel launch (a fused rollout launch runs T steps)')
    if errors:
      doc['errors'] = errors
  os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
  with open(args.out, 'w') as f:
    json.dump(doc, f, indent=1)
  print(args.name, json.dumps(doc['per_launch']))


if __name__ == '__main__':
  main()
