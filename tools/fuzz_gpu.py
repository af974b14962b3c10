"""Randomised differential test: HIP engine vs the C oracle over random families, settings, batch
sizes, lane offsets, wrappers, explicit resets, rollouts and Logging — run on the GPU box:
    python tools/fuzz_gpu.py --seconds 120 --seed 0
Integer / grid families are compared bit for bit; physics families teacher-forced at 1e-6."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bsuite_amd.utils import wrappers  # noqa: E402
from oracle import coracle, logging_oracle  # noqa: E402
from tests import engine_util as eu  # noqa: E402


def random_case(rng):
  fam = rng.choice(['deep_sea', 'catch', 'bandit', 'memory_chain', 'umbrella_chain', 'discounting_chain',
                    'cartpole', 'cartpole_swingup', 'mountain_car'])
  if fam == 'deep_sea':
    kw = dict(size=int(rng.integers(1, 41)), deterministic=bool(rng.integers(2)), mapping_seed=int(rng.integers(100)),
              unscaled_move_cost=float(rng.choice([0.01, 0.05, 0.0])))
  elif fam == 'catch':
    kw = dict(rows=int(rng.integers(2, 20)), columns=int(rng.integers(1, 12)))
  elif fam == 'bandit':
    kw = dict(mapping_seed=int(rng.integers(50)), num_actions=int(rng.integers(1, 20)))
  elif fam == 'memory_chain':
    kw = dict(memory_length=int(rng.integers(1, 12)), num_bits=int(rng.integers(1, 50)))
  elif fam == 'umbrella_chain':
    kw = dict(chain_length=int(rng.integers(1, 10)), n_distractor=int(rng.integers(0, 120)))
  elif fam == 'discounting_chain':
    kw = dict(mapping_seed=int(rng.integers(20)))
  elif fam == 'cartpole':
    kw = dict(max_time=float(rng.choice([10., 0.2])))
  elif fam == 'cartpole_swingup':
    kw = dict(height_threshold=float(rng.random()), x_reward_threshold=float(rng.random()),
              init_range=float(rng.choice([0.05, 3.0])))
  else:
    kw = dict(max_steps=int(rng.integers(2, 60)))
  wrap = None
  r = rng.random()
  if r < 0.25:
    wrap = ('noise', float(rng.choice([0.1, 1.0, 10.0])))
  elif r < 0.5:
    wrap = ('scale', float(rng.choice([0.001, 30.0])))
  return fam, kw, wrap


def run_case(rng, case_id):
  fam, kw, wrap = random_case(rng)
  B = int(rng.choice([1, 3, 64, 255, 256, 257, 1000, 4100]))
  off = int(rng.choice([0, 5, (1 << 32) - 100, (1 << 40) + 3]))
  seed = int(rng.integers(1 << 40))
  step0 = int(rng.choice([0, (1 << 32) - 3, (1 << 33) + 1]))
  phys = fam in ('cartpole', 'cartpole_swingup', 'mountain_car')
  by_step = bool(rng.integers(2))
  # deep_sea / catch: half of the cases run the delta observation mode (persistent buffers patched in
  # place, no rollouts there), with 1-3 observation buffers
  delta = fam in ('deep_sea', 'catch') and bool(rng.integers(2))
  env = eu.make_env(fam, kw, batch=B, lane_offset=off, seed=seed, wrap=wrap, num_buffers=int(rng.integers(1, 4)),
                    observation_mode='delta' if delta else 'dense')
  raw = eu.raw(env)
  raw._step_index = step0
  # half of the cases run WITHOUT the Logging wrapper: the lean kernel instantiations, per-thread
  # observation stores and episode-end info folding of cartpole / mountain_car are then the ones tested
  use_log = bool(rng.integers(2))
  log = wrappers.Logging(env, None, log_by_step=by_step, max_rows=400) if use_log else env
  chk = eu.PhysicsChecker(fam, kw, B) if phys else None
  orc = coracle.OracleEnv(fam, kw, np.arange(off, off + B, dtype=np.uint64), seed=seed, wrap=wrap)
  trk = logging_oracle.TrackOracle(B, list(orc.bsuite_info()), log_by_step=by_step)
  T = int(rng.integers(5, 60))
  t = 0
  while t < T:
    mode = rng.random()
    n = 1
    if mode < 0.1:
      ts = log.reset(); force = True; acts = np.zeros((1, B), np.int32)
    elif mode < 0.3 and not phys and not delta:
      n = int(rng.integers(2, 7))
      acts = rng.integers(0, orc.num_actions, size=(n, B)).astype(np.int32)
      ro = log.rollout(torch.from_numpy(acts).cuda()); force = False
    else:
      acts = rng.integers(0, orc.num_actions, size=(1, B)).astype(np.int32)
      force = False
      if phys and t > 0:
        if fam == 'mountain_car':
          st32 = np.stack([orc.s['position'], orc.s['velocity']]).astype(np.float32); k = orc.s['timestep'].astype(np.int32)
        else:
          st32 = orc.s['state'][:, :4].T.astype(np.float32)
          k = np.rint(orc.s['state'][:, 4] / orc.cfg.timescale).astype(np.int32)
        raw._state['state'].copy_(torch.from_numpy(np.ascontiguousarray(st32)).cuda())
        raw._state['steps'].copy_(torch.from_numpy(k | (orc.reset_next.astype(np.int32) << 30)).cuda())
      ts = log.step(torch.from_numpy(acts[0]).cuda())
    for j in range(n):
      st, r, d, o = orc.call(acts[j], step0 + t + j, force_reset=force)
      trk.track(st, r, orc.bsuite_info())
      if n > 1:
        g = (ro.step_type[j], ro.reward[j], ro.discount[j], ro.observation[j])
      else:
        g = (ts.step_type, ts.reward, ts.discount, ts.observation)
      gst, gr, gd, go = [x.cpu().numpy() for x in g]
      live = st != 0
      if phys:   # |a-b| <= 1e-6*max(1,|b|); a lane may differ in step_type / reward / sign flag only on a verified tie
        chk.check((gst, gr, gd, go), (st, r, d, o), eu.oracle_physics_state(orc, fam), msg=str((case_id, fam, kw, wrap, B, t + j)))
      else:
        np.testing.assert_array_equal(gst, st, err_msg=str((case_id, fam, kw, wrap, B, off, t + j)))
        np.testing.assert_array_equal(eu.f32_bits(gr[live]), eu.f32_bits(r[live].astype(np.float32)),
                                      err_msg=str((case_id, fam, kw, wrap, B, t + j)))
        np.testing.assert_array_equal(eu.f32_bits(go), eu.f32_bits(o), err_msg=str((case_id, fam, kw, B, t + j)))
    t += n
  clean = ~chk.tainted if phys else np.ones(B, bool)
  for k_, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(raw.bsuite_info()[k_].cpu().numpy()[clean], v[clean], err_msg=str((case_id, fam, kw, k_, use_log)))
  if phys:
    return 'tie' if chk.ties else 'ok'
  if use_log:
    c = log.counters()
    np.testing.assert_array_equal(c['steps'].cpu().numpy(), trk.steps)
    np.testing.assert_array_equal(c['total_return'].cpu().numpy(), trk.total_return)
    np.testing.assert_array_equal(log.num_rows().cpu().numpy(), [len(x) for x in trk.rows])
  return 'ok'


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--seconds', type=float, default=60)
  ap.add_argument('--seed', type=int, default=0)
  args = ap.parse_args()
  rng = np.random.default_rng(args.seed)
  t0, n, ties = time.time(), 0, 0
  while time.time() - t0 < args.seconds:
    res = run_case(rng, n)
    ties += res == 'tie'
    n += 1
  print(f'fuzz: {n} random cases passed in {time.time() - t0:.0f} s ({ties} ended early on a physics threshold tie)')


if __name__ == '__main__':
  main()
