"""Per-rank proxy of the strong-scaling curve, measured on ONE GPU -> profiles/rNN/strong_scaling_proxy.json.

  python tools/strong_scaling_proxy.py <out.json>

The path has no data-path communication (lanes never interact; the only collective is the end-of-rollout all-gather of
a few dozen bytes), so rank r of an N-GPU strong-scaled run does exactly what one GPU does at 2^20 / N lanes.  This
script times deep_sea/10 and catch/0 (BASELINE.json's metric) at 2^20, 2^19, 2^18 and 2^17 lanes — eager step(), the
pipelined rollout, catch also as a HIP graph of 16 launches — and rank 0's / rank N-1's bin-packed share of the 468-id
sweep (BASELINE config 5), and reports the PREDICTED N-GPU speed-up  N * v(2^20 / N) / v(2^20)  (sweep: t(1) / slowest rank).  It is a proxy: no multi-GPU
timing exists in this repository (gpurun exposes one GPU); the curve itself is the driver's to measure.
Reference fan-out being replaced: bsuite/baselines/utils/pool.py:28-54.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
  out_path = sys.argv[1]
  args = argparse.Namespace(gpus=1, no_stagger=False)
  r = bench.Rank(args)
  doc = dict(what='single-GPU time per step at the per-rank lane count of an N-GPU strong-scaled run of 2^20 global lanes; '
                  'predicted speed-up = N * v(2^20/N) / v(2^20) (no communication on the path)',
             note='a proxy, NOT a measured scaling curve', script='tools/strong_scaling_proxy.py', records={})
  for workload, T in (('deep_sea', 16), ('catch', 32)):
    for mode in ('eager', 'rollout'):
      base = None
      for n in (1, 2, 4, 8):
        lanes = (1 << 20) // n
        steps = 96 if workload == 'deep_sea' and n == 1 else 256
        warm = 32
        if mode == 'rollout':
          steps, warm = (steps + T - 1) // T * T, T
        m = r.measure(workload, lanes, steps, warm, mode, T if mode == 'rollout' else 0)
        v = m['value']
        base = base or v
        doc['records'][f'{workload} {mode} N={n}'] = dict(lanes_per_gpu=lanes, us_per_step=m['kernel_ms'] * 1e3, env_steps_per_s_per_gpu=v,
                                                        frac_hbm=m['achieved'] / bench.HBM_PEAK_GBPS, predicted_speedup=n * v / base)
        print(workload, mode, n, json.dumps(bench.sig(doc['records'][f'{workload} {mode} N={n}'])), flush=True)
  # catch under a HIP graph of 16 step() launches: what a rank's share costs once the host is out of the way
  base = None
  for n in (1, 2, 4, 8):
    lanes = (1 << 20) // n
    m = r.measure('catch', lanes, 256, 32, 'graph', 16)
    base = base or m['value']
    doc['records'][f'catch graph16 N={n}'] = dict(lanes_per_gpu=lanes, us_per_step=m['kernel_ms'] * 1e3, env_steps_per_s_per_gpu=m['value'],
                                                  frac_hbm=m['achieved'] / bench.HBM_PEAK_GBPS, predicted_speedup=n * m['value'] / base)
    print('catch graph16', n, json.dumps(bench.sig(doc['records'][f'catch graph16 N={n}'])), flush=True)
  # BASELINE config 5: the busiest and the lightest rank's bin-packed share of the 468-id sweep (whole segments, 2^20 / 468
  # lanes each), closed-loop and pipelined; the N-GPU time is the slowest rank's: predicted speed-up = t(1) / max_r t_N(r)
  # over the two ranks timed.  Phase 0 does not shrink with the lanes (dispatch + one workgroup life): the workload
  # that scales worst.
  import tempfile
  import numpy as np
  import torch
  from bsuite_amd import sweep_batch as sb
  from bsuite_amd.utils import datasets
  d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
  tmp = tempfile.mkdtemp(prefix='bsx_mnist_')
  datasets.write_idx_files(tmp, d['images_u8'], d['labels'])
  mn = dict(data_dir=tmp)
  t1 = {}
  for n in (1, 2, 4, 8):
    worst = {}
    for rank in sorted({0, n - 1}):
      batch = sb.SweepBatch(None, 1 << 20, device=r.dev, seed=42, rank=rank, world_size=n, env_kwargs=dict(mnist=mn, mnist_noise=mn, mnist_scale=mn))
      acts = batch.random_actions(seed=1, ring=16)
      nbytes = float(sum(l * sb.bytes_per_step(int(np.prod(e.observation_spec().shape))) for e, (_, _, l) in zip(batch.envs, batch.segments)))
      for name, pipelined in (('closed', False), ('pipelined', True)):
        batch.prepare_groups(acts, pipelined=pipelined)
        for _ in range(30):
          batch.step_grouped()
        torch.cuda.synchronize(r.dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
          batch.step_grouped()
        e1.record()
        torch.cuda.synchronize(r.dev)
        us = e0.elapsed_time(e1) / 200 * 1e3
        batch.release_groups()
        worst[name] = max(worst.get(name, 0.0), us)
        doc['records'][f'sweep {name} N={n} rank={rank}'] = dict(segments=len(batch.envs), lanes=batch.lanes(), alg_bytes=nbytes, us_per_step=us,
                                                               frac_hbm=nbytes / (us * 1e-6) / 1e9 / bench.HBM_PEAK_GBPS)
        print('sweep', name, n, rank, json.dumps(bench.sig(doc['records'][f'sweep {name} N={n} rank={rank}'])), flush=True)
      del batch, acts
      torch.cuda.empty_cache()
    for name, us in worst.items():
      t1.setdefault(name, us)
      doc['records'][f'sweep {name} N={n}'] = dict(us_per_step_slowest_rank=us, predicted_speedup=t1[name] / us)
      print('sweep', name, n, json.dumps(bench.sig(doc['records'][f'sweep {name} N={n}'])), flush=True)
  os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
  with open(out_path, 'w') as f:
    json.dump(bench.sig(doc, 6), f, indent=1)
  r.close()


if __name__ == '__main__':
  main()
