"""BASELINE configs[0]: the scalar dm_env view (`load_from_id('catch/0')`, B=1) driven like the
reference run loop — a compatibility path (one host round trip per step), timed for the record."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bsuite_amd

for bid in ('catch/0', 'deep_sea/10', 'cartpole/0'):
  env = bsuite_amd.load_from_id(bid)
  n = env.action_spec().num_values
  rng = np.random.RandomState(0)
  acts = rng.randint(n, size=4096)
  ts = env.reset()
  for a in acts[:200]:
    ts = env.step(int(a))
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for a in acts:
    ts = env.step(int(a))
  dt = time.perf_counter() - t0
  print(json.dumps(dict(bsuite_id=bid, view='scalar (batch=None)', steps=len(acts), steps_per_s=round(len(acts) / dt),
                        us_per_step=round(dt / len(acts) * 1e6, 1))), flush=True)

# the way most bsuite users run it: load_and_record_to_csv + the reference run loop shape
import tempfile
tmp = tempfile.mkdtemp(prefix='bsx_csv_')
env = bsuite_amd.load_and_record_to_csv('catch/0', results_dir=tmp, overwrite=True)
rng = np.random.RandomState(0)
acts = rng.randint(3, size=4096)
ts = env.reset()
for a in acts[:200]:
  ts = env.step(int(a))
t0 = time.perf_counter()
for a in acts:
  ts = env.step(int(a))
dt = time.perf_counter() - t0
print(json.dumps(dict(bsuite_id='catch/0', view='scalar + load_and_record_to_csv', steps=len(acts),
                      steps_per_s=round(len(acts) / dt), us_per_step=round(dt / len(acts) * 1e6, 1))), flush=True)
