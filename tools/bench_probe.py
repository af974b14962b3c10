import os, sys, time, argparse
sys.path.insert(0, os.getcwd())
import bench
ap_args = argparse.Namespace(gpus=1, steps=400, warmup=50, lanes=1 << 20, workload='discounting_chain', graph=0, rollout=0,
                             observation_mode='dense', logging=False, no_also=True, no_stagger=False, no_cpu_baseline=True, strong=False)
r = bench.Rank(ap_args)
torch = r.torch
orig = r.timed
def timed(run, steps, warmup):
  run(warmup); r.sync_all()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter(); e0.record(); run(steps); th = time.perf_counter() - t0; e1.record(); torch.cuda.synchronize(r.dev)
  wall = time.perf_counter() - t0
  print('  host enqueue %.2f us/step, GPU %.2f us/step, wall %.2f' % (th / steps * 1e6, e0.elapsed_time(e1) / steps * 1e3, wall / steps * 1e6))
  return wall, e0.elapsed_time(e1) / steps
r.timed = timed
for w in ('discounting_chain', 'bandit', 'mountain_car'):
  for stag in (True, False):
    print(w, 'stagger' if stag else 'lock-step')
    r.measure(w, 1 << 20, 400, 50, 'eager', 0, 'dense', False, stagger=stag)
