"""Per-rank proxy of the strong-scaling curve, measured on ONE GPU -> profiles/rNN/strong_scaling_proxy.json.

  python tools/strong_scaling_proxy.py <out.json>

The path has no data-path communication (lanes never interact; the only collective is the end-of-rollout all-gather of
a few dozen bytes), so rank r of an N-GPU strong-scaled run does exactly what one GPU does at 2^20 / N lanes.  This
script times deep_sea/10 and catch/0 (BASELINE.json's metric) at 2^20, 2^19, 2^18 and 2^17 lanes — eager step() and the
pipelined rollout — and reports the PREDICTED N-GPU speed-up  N * v(2^20 / N) / v(2^20).  It is a proxy: no multi-GPU
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
  os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
  with open(out_path, 'w') as f:
    json.dump(bench.sig(doc, 6), f, indent=1)
  r.close()


if __name__ == '__main__':
  main()
