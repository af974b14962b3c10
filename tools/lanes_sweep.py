"""us per step of bench workloads at several batch sizes, one process (for A/B runs of a tuning knob: run it twice
under BSX_NATIVE_LIB=<tuning build> with the knob set differently).

  python tools/lanes_sweep.py [--mode eager|rollout] [--T 16] [--steps 300] <workload>... -- <lanes>...
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
  sep = sys.argv.index('--')
  ap = argparse.ArgumentParser()
  ap.add_argument('--mode', default='eager')
  ap.add_argument('--T', type=int, default=16)
  ap.add_argument('--steps', type=int, default=300)
  ap.add_argument('workloads', nargs='+')
  a = ap.parse_args(sys.argv[1:sep])
  lanes = [int(eval(x)) for x in sys.argv[sep + 1:]]   # pylint: disable=eval-used  ("2**17" is allowed)
  r = bench.Rank(argparse.Namespace(gpus=1, no_stagger=False))
  for w in a.workloads:
    for n in lanes:
      steps = a.steps if a.mode == 'eager' else (a.steps + a.T - 1) // a.T * a.T
      m = r.measure(w, n, steps, 32 if a.mode == 'eager' else a.T * 2, a.mode, a.T if a.mode != 'eager' else 0)
      print(json.dumps(bench.sig(dict(workload=w, mode=a.mode, lanes=n, us_per_step=m['kernel_ms'] * 1e3,
                                      frac_hbm=m['achieved'] / bench.HBM_PEAK_GBPS, env_steps_per_s=m['value']))), flush=True)
  r.close()


if __name__ == '__main__':
  main()
