"""Where phase 0 of a config-5 sweep step (one launch: every lane's advance) spends its time: per-workgroup
start / end stamps (bsx_group_trace, 100 MHz wall clock) of one step, summarised per family.

  python tools/sweep_phase0_trace.py [--lanes 1048576] [--out gpurun_out/sweep_phase0_trace.json]
"""
import argparse
import collections
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bsuite_amd import _native  # noqa: E402
from bsuite_amd import sweep_batch as sb  # noqa: E402
from bsuite_amd.utils import datasets  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--lanes', type=int, default=1 << 20)
  ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'sweep_phase0_trace.json'))
  ap.add_argument('--life', action='store_true',
                  help='a -DBSX_TRACE_LIFE build (BSX_NATIVE_LIB): 8 stamps inside every workgroup, medians per family')
  ap.add_argument('--phase0-only', action='store_true',
                  help='step only the lane-advance launch, back to back: no store stream in between to sweep the caches')
  args = ap.parse_args()
  d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
  tmp = tempfile.mkdtemp(prefix='bsx_mnist_')
  datasets.write_idx_files(tmp, d['images_u8'], d['labels'])
  mn = dict(data_dir=tmp)
  batch = sb.SweepBatch(None, args.lanes, seed=42, env_kwargs=dict(mnist=mn, mnist_noise=mn, mnist_scale=mn))
  acts = batch.random_actions(seed=1)
  batch.prepare_groups(acts)
  stream = torch.cuda.current_stream().cuda_stream

  def one_step():
    if args.phase0_only:
      _native.check(_native.lib.bsx_group_step_phase(batch._groups[0], 0, stream), 'bsx_group_step_phase')
    else:
      batch.step_grouped()
  for _ in range(30):
    one_step()
  torch.cuda.synchronize()
  n_blocks = sum((lanes + 255) // 256 for _, _, lanes in batch.segments)
  buf = torch.zeros((11 if args.life else 3) * n_blocks, dtype=torch.int64, device='cuda')
  _native.check(_native.lib.bsx_group_trace(batch._groups[0], buf.data_ptr(), buf.numel()), 'bsx_group_trace')
  names = {v: k for k, v in _native.FAMILY_IDS.items()}
  summary = []
  for rep in range(3):
    one_step()
    torch.cuda.synchronize()
    raw = buf.cpu().numpy()
    t = raw[:3 * n_blocks].reshape(-1, 3)
    t0 = t[:, 0].min()
    start, end, tag = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2]      # us
    rec = {'rep': rep, 'workgroups': int(n_blocks), 'span_us': float(end.max()),
           'last_start_us': float(start.max()), 'families': {}}
    for fam in sorted(set(tag.tolist())):
      m = tag == fam
      life = end[m] - start[m]
      rec['families'][names.get(int(fam), str(fam))] = {
          'workgroups': int(m.sum()), 'life_us_median': float(np.median(life)), 'life_us_p95': float(np.percentile(life, 95)),
          'life_us_max': float(life.max()), 'first_start_us': float(start[m].min()), 'last_end_us': float(end[m].max())}
    # how many workgroups are resident over time
    ev = sorted([(s, 1) for s in start] + [(e, -1) for e in end])
    cur, peak = 0, 0
    for _, dlt in ev:
      cur += dlt
      peak = max(peak, cur)
    rec['peak_resident_workgroups'] = peak
    hist = collections.Counter(int(s) for s in start)
    rec['starts_per_us'] = [hist.get(u, 0) for u in range(int(end.max()) + 1)]
    if args.life:
      # stamps: 0 entry, 1 map entry arrived, 2 argument slot + call counter arrived, 3 state + action arrived (advance
      # bodies only), 4 computed, 5 stores issued, 6 past the final barrier, 7 past the retirement ticket
      life = raw[3 * n_blocks:].reshape(-1, 8).astype(np.float64)
      rec['life_stamps_us_median'] = {}
      for fam in sorted(set(tag.tolist())):
        m = tag == fam
        rows = life[m]
        base = rows[:, 0:1]
        rel = np.where(rows > 0, (rows - base) / 100.0, np.nan)
        rec['life_stamps_us_median'][names.get(int(fam), str(fam))] = [None if np.isnan(x) else round(float(x), 2) for x in np.nanmedian(rel, axis=0)]
      print(json.dumps(rec['life_stamps_us_median']))
    summary.append(rec)
    print(json.dumps({k: v for k, v in rec.items() if k != 'starts_per_us'}))
    print('starts per us:', rec['starts_per_us'])
  _native.check(_native.lib.bsx_group_trace(batch._groups[0], None, 0), 'bsx_group_trace')
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  json.dump(summary, open(args.out, 'w'), indent=1)
  batch.release_groups()


if __name__ == '__main__':
  main()
