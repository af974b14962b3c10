"""Where a config-5 sweep step spends its time: the grouped sweep restricted to subsets of bsuite_ids
(each subset keeps the lane count its ids have in the full 2^20-lane sweep), timed serially with HIP events.

  python tools/sweep_breakdown.py [--steps 100]
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bsuite_amd import sweep  # noqa: E402
from bsuite_amd import sweep_batch as sb  # noqa: E402
from bsuite_amd.utils import datasets  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--graph', type=int, default=0)
  args = ap.parse_args()
  d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
  tmp = tempfile.mkdtemp(prefix='bsx_mnist_')
  datasets.write_idx_files(tmp, d['images_u8'], d['labels'])
  mn = dict(data_dir=tmp)
  per = (1 << 20) // len(sweep.SWEEP)
  names = []
  for b in sweep.SWEEP:
    n = b.split('/')[0]
    if n not in names:
      names.append(n)
  subsets = {'ALL': list(sweep.SWEEP)}
  big = ('deep_sea', 'deep_sea_stochastic', 'catch', 'catch_noise', 'catch_scale', 'mnist', 'mnist_noise', 'mnist_scale')
  subsets['small families together'] = [b for b in sweep.SWEEP if b.split('/')[0] not in big]
  for n in names:
    subsets[n] = [b for b in sweep.SWEEP if b.split('/')[0] == n]
  for label, ids in list(subsets.items())[:int(os.environ.get("BSX_BREAKDOWN_N", "99"))]:
    batch = sb.SweepBatch(ids, per * len(ids), device='cuda:0', seed=42,
                          env_kwargs=dict(mnist=mn, mnist_noise=mn, mnist_scale=mn))
    acts = batch.random_actions(seed=1)
    batch.prepare_groups(acts)
    run = batch.step_grouped
    if args.graph:
      batch.capture_grouped(2, phased=os.environ.get('BSX_SWEEP_PHASED', '1') != '0')
      run = batch.replay_grouped
    for _ in range(10):
      run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
      run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    nbytes = sum(l * sb.bytes_per_step(int(np.prod(e.observation_spec().shape))) for e, (_, _, l) in zip(batch.envs, batch.segments))
    print(json.dumps(dict(subset=label, ids=len(ids), lanes=batch.lanes(), us_per_step=round(ms * 1e3, 2),
                          MB=round(nbytes / 1e6, 2), GBps=round(nbytes / ms / 1e6, 1), groups=len(batch._groups))), flush=True)
    batch.release_groups()
    del batch, acts
    torch.cuda.empty_cache()


if __name__ == '__main__':
  main()
