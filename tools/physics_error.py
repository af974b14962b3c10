"""Measured f32-vs-f64 error of the physics families on the GPU -> gpurun_out/physics_error.json
(copy to profiles/rNN/).  Runs on the GPU box; the C oracle (f64, the reference's arithmetic) is the checker.

1. teacher-forced (the contract, BASELINE north_star): every call the device state is set to
   f32(reference f64 state); reported: max over lanes and calls of |a-b|/max(1,|b|) per observation
   component and for the reward, and the number of threshold ties (tests/engine_util.PhysicsChecker).
2. free-running (SURVEY §7 hard part 4, information only): engine and oracle start from the same
   draws and see the same actions but are never re-synchronised; reported: median / max over lanes
   of the state difference after 10 / 100 / 300 / 1000 calls among lanes whose step_type sequences
   still agree, and the fraction of lanes whose sequences have diverged (a threshold crossed on a
   different call).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import coracle          # noqa: E402
from tests import engine_util as eu  # noqa: E402

CASES = [('cartpole', {}), ('cartpole_swingup', {}), ('mountain_car', {})]


def teacher_forced(family, kwargs, batch=16384, T=400, seed=11):
  env = eu.make_env(family, kwargs, batch=batch, lane_offset=0, seed=seed)
  orc = coracle.OracleEnv(family, kwargs, np.arange(batch, dtype=np.uint64), seed=seed)
  rng = np.random.default_rng(3)
  raw = eu.raw(env)
  chk = eu.PhysicsChecker(family, kwargs, batch)
  comp = None
  for t in range(T):
    a = rng.integers(0, 3, size=batch).astype(np.int32)
    if t > 0:
      eu.teacher_force(raw, orc, family)
    ts = env.step(torch.from_numpy(a).cuda())
    want = orc.call(a, t)
    got = eu.to_np(ts)
    chk.check(got, want, eu.oracle_physics_state(orc, family), msg=f'{family} t={t}')
    _, err = eu.within_tol(got[3], want[3])
    same = got[0] == want[0]
    e = err.reshape(batch, -1)[same].max(axis=0)
    comp = e if comp is None else np.maximum(comp, e)
  return dict(lanes=batch, calls=T, tolerance=eu.PHYS_TOL, max_scaled_error_observation=chk.max_err['observation'],
              max_scaled_error_reward=chk.max_err['reward'],
              max_scaled_error_per_observation_component=[float(x) for x in comp],
              threshold_ties=chk.ties, lane_calls=batch * T)


def free_running(family, kwargs, batch=8192, T=1000, seed=11):
  env = eu.make_env(family, kwargs, batch=batch, lane_offset=0, seed=seed)
  orc = coracle.OracleEnv(family, kwargs, np.arange(batch, dtype=np.uint64), seed=seed)
  rng = np.random.default_rng(3)
  raw = eu.raw(env)
  agree = np.ones(batch, bool)
  out = {}
  for t in range(T + 1):
    a = rng.integers(0, 3, size=batch).astype(np.int32)
    ts = env.step(torch.from_numpy(a).cuda())
    st = orc.call(a, t)[0]
    agree &= ts.step_type.cpu().numpy() == st
    if t in (10, 100, 300, 1000):
      dev = raw._state['state'].double().cpu().numpy()          # [n_state, B]
      if family == 'mountain_car':
        ref = np.stack([orc.s['position'], orc.s['velocity']])
        names = ['position', 'velocity']
      else:
        ref = orc.s['state'][:, :4].T
        names = ['x', 'x_dot', 'theta', 'theta_dot']
      d = np.abs(dev - ref)
      if family != 'mountain_car':
        d[2] = np.minimum(d[2], 2 * np.pi - d[2])               # angles wrap
      rec = dict(lanes_still_in_step=int(agree.sum()), fraction_diverged=float(1 - agree.mean()))
      if agree.any():
        for j, n in enumerate(names):
          rec[n] = dict(median=float(np.median(d[j][agree])), max=float(d[j][agree].max()))
      out[f'after_{t}_calls'] = rec
  return out


def main():
  doc = dict(device=torch.cuda.get_device_name(0),
             what='f32 engine vs f64 C oracle (the reference arithmetic), random actions; see tools/physics_error.py')
  for family, kwargs in CASES:
    doc[family] = dict(teacher_forced=teacher_forced(family, kwargs), free_running=free_running(family, kwargs))
    print(family, json.dumps(doc[family]['teacher_forced']), flush=True)
  out = os.path.join(ROOT, 'gpurun_out', 'physics_error.json')
  os.makedirs(os.path.dirname(out), exist_ok=True)
  with open(out, 'w') as f:
    json.dump(doc, f, indent=1)
  print('wrote', out)


if __name__ == '__main__':
  main()
