"""Times the UNMODIFIED reference (google-deepmind/bsuite under /root/reference, imported with the
third-party stand-ins of oracle/ref_shims) on this machine's host cores -> profiles/rNN/cpu_reference_numpy.json.

  python tools/cpu_reference_numpy.py [out.json] [--seconds S]

Only runs where /root/reference exists (the build container): the reference cannot travel to the GPU
box, so bench.py reports this file's numbers as `cpu_baseline` (kind "reference", provenance stated)
next to the C port it times live on the GPU box's own cores.

Loop shape = bsuite/baselines/experiment.py:43-57 with the random agent of
bsuite/baselines/random/agent.py:35-37 inlined (`rng.randint(num_actions)`); env-steps count every
environment call, reset() included (they are API calls: bsuite/environments/base.py:54-65).  The
all-cores number runs one process per core, each on its own environment — the shape of the
reference's own fan-out (bsuite/baselines/utils/pool.py:35,48).
"""
import json
import multiprocessing
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

IDS = ('deep_sea/10', 'catch/0', 'cartpole/0', 'mountain_car/0')


def _loop(bsuite_id, seconds, seed=0):
  import numpy as np
  from oracle import replay
  bs = replay.import_reference()
  env = bs.load_from_id(bsuite_id)
  num_actions = env.action_spec().num_values
  rng = np.random.RandomState(seed)
  calls = 0
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < seconds:
    timestep = env.reset()
    calls += 1
    while not timestep.last():
      timestep = env.step(rng.randint(num_actions))
      calls += 1
  return calls, time.perf_counter() - t0


def _worker(args):
  return _loop(*args)


def main():
  out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else None
  seconds = float(sys.argv[sys.argv.index('--seconds') + 1]) if '--seconds' in sys.argv else 10.0
  cores = len(os.sched_getaffinity(0))
  cpu = ''
  try:
    with open('/proc/cpuinfo') as f:
      cpu = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
  except (OSError, StopIteration):
    pass
  import numpy as np
  doc = dict(
      what='unmodified reference bsuite (/root/reference + oracle/ref_shims) stepped through load_from_id with '
           'uniform random actions; env-steps/s counts every reset()/step() call',
      script='tools/cpu_reference_numpy.py', host=dict(cpu=cpu, cores=cores, machine=platform.machine(),
                                                        python=platform.python_version(), numpy=np.__version__,
                                                        where='build container (the GPU box has no /root/reference)'),
      seconds_per_measurement=seconds, results={})
  for bid in IDS:
    calls, dt = _loop(bid, seconds)
    rec = dict(single_core=dict(value=calls / dt, unit='env-steps/s', cores=1, calls=calls, seconds=dt))
    with multiprocessing.get_context('spawn').Pool(cores) as pool:
      t0 = time.perf_counter()
      res = pool.map(_worker, [(bid, seconds / 2, 1 + j) for j in range(cores)])
      wall = time.perf_counter() - t0
    rec['all_cores'] = dict(value=sum(c for c, _ in res) / max(d for _, d in res), unit='env-steps/s', cores=cores,
                            calls=sum(c for c, _ in res), seconds=max(d for _, d in res), wall_incl_startup=wall)
    doc['results'][bid] = rec
    print(bid, json.dumps(rec), flush=True)
  if out:
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, 'w') as f:
      json.dump(doc, f, indent=1)


if __name__ == '__main__':
  main()
