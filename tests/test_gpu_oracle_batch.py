"""GPU: the HIP kernels against the C oracle on seeded random-action rollouts at batch sizes the
golden fixtures do not reach — ragged (not a multiple of the tile), one lane, several thousand
lanes, long lane offsets — and through explicit reset() calls.  Integer / grid families bit-exact;
physics families teacher-forced from the oracle's f64 state at 1e-6."""
import numpy as np
import pytest
import torch

from oracle import coracle
from tests import engine_util as eu

pytestmark = pytest.mark.gpu

CASES = [
    ('deep_sea', dict(size=30, mapping_seed=42), None),
    ('deep_sea', dict(size=10, mapping_seed=42), None),
    ('deep_sea', dict(size=9, mapping_seed=1), None),                      # odd: straddling chunks
    ('deep_sea', dict(size=13, deterministic=False, mapping_seed=42), None),
    ('deep_sea', dict(size=1, mapping_seed=0), None),
    ('deep_sea', dict(size=64, mapping_seed=5), None),
    ('deep_sea', dict(size=10, deterministic=False, mapping_seed=42), ('noise', 0.5)),
    ('catch', dict(), None),
    ('catch', dict(rows=7, columns=3), None),
    ('catch', dict(rows=2, columns=1), None),
    ('catch', dict(rows=64, columns=64), None),
    ('catch', dict(), ('scale', 30.0)),
    ('bandit', dict(mapping_seed=3), None),
    ('bandit', dict(mapping_seed=3), ('noise', 1.0)),
    ('memory_chain', dict(memory_length=3, num_bits=5), None),
    ('memory_chain', dict(memory_length=1, num_bits=62), None),
    ('memory_chain', dict(memory_length=30, num_bits=1), None),
    ('umbrella_chain', dict(chain_length=4, n_distractor=20), None),
    ('umbrella_chain', dict(chain_length=2, n_distractor=253), None),
    ('umbrella_chain', dict(chain_length=6, n_distractor=64), ('noise', 0.1)),
    # row shapes either side of the per-thread-store / packed-record split and of the 32-bit word boundaries
    ('umbrella_chain', dict(chain_length=3, n_distractor=0), None),     # 3 floats: one 12-byte store
    ('umbrella_chain', dict(chain_length=3, n_distractor=2), None),     # 5 floats: packed, rows not 16-byte aligned
    ('umbrella_chain', dict(chain_length=3, n_distractor=5), None),     # 8 floats: four 8-byte stores
    ('umbrella_chain', dict(chain_length=3, n_distractor=32), None),    # exactly one bit word
    ('umbrella_chain', dict(chain_length=3, n_distractor=33), None),
    ('memory_chain', dict(memory_length=2, num_bits=2), None),          # 4 floats: per-thread stores
    ('memory_chain', dict(memory_length=2, num_bits=32), None),
    ('memory_chain', dict(memory_length=2, num_bits=33), None),
    ('discounting_chain', dict(mapping_seed=2), None),
]


@pytest.mark.parametrize('family,kwargs,wrap', CASES)
@pytest.mark.parametrize('batch,lane_offset', [(1, 0), (1000, 0), (4099, (1 << 32) - 17)])
def test_random_rollout_bit_exact(family, kwargs, wrap, batch, lane_offset):
  if family == 'catch' and kwargs.get('rows') == 64 and batch > 1000:
    pytest.skip('big board at big batch adds nothing')
  seed = 1234
  env = eu.make_env(family, kwargs, batch=batch, lane_offset=lane_offset, seed=seed, wrap=wrap)
  orc = coracle.OracleEnv(family, kwargs, np.arange(lane_offset, lane_offset + batch, dtype=np.uint64),
                          seed=seed, wrap=wrap)
  rng = np.random.default_rng(batch)
  T = 45 if family != 'discounting_chain' else 105
  for t in range(T):
    a = rng.integers(0, orc.num_actions, size=batch).astype(np.int32)
    force = t in (11, 12)
    ts = env.reset() if force else env.step(torch.from_numpy(a).cuda())
    st, r, d, o = orc.call(a, t, force_reset=force)
    gst, gr, gd, go = eu.to_np(ts)
    np.testing.assert_array_equal(gst, st, err_msg=f't={t}')
    live = st != 0
    np.testing.assert_array_equal(eu.f32_bits(gr[live]), eu.f32_bits(r[live].astype(np.float32)), err_msg=f'reward t={t}')
    np.testing.assert_array_equal(gd[live], d[live].astype(np.float32))
    assert (gr[~live] == 0).all() and (gd[~live] == 1).all()
    np.testing.assert_array_equal(eu.f32_bits(go), eu.f32_bits(o), err_msg=f'obs t={t}')
  info = env.bsuite_info()
  for k, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(info[k].cpu().numpy(), v, err_msg=k)
  c = eu.raw(env).episode_counters().cpu().numpy()
  assert c[0] >= 0 and c[1] >= batch


@pytest.mark.parametrize('family,kwargs', [
    ('cartpole', dict()), ('cartpole_swingup', dict(height_threshold=0.25, x_reward_threshold=0.75)),
    ('cartpole_swingup', dict(init_range=3.2)), ('mountain_car', dict()), ('mountain_car', dict(max_steps=30))])
def test_physics_teacher_forced(family, kwargs):
  """Every call compared at |a-b| <= 1e-6*max(1,|b|); a lane may differ in step_type / reward / sign
  flag only on a verified threshold tie (tests/engine_util.PhysicsChecker); bsuite_info (raw_return,
  best_episode, total_upright) exact on every lane that never tied — episodes end, auto-reset and
  restart inside the horizon."""
  batch, seed, T = 3001, 99, 260
  env = eu.make_env(family, kwargs, batch=batch, lane_offset=5, seed=seed)
  orc = coracle.OracleEnv(family, kwargs, np.arange(5, 5 + batch, dtype=np.uint64), seed=seed)
  rng = np.random.default_rng(7)
  r_env = eu.raw(env)
  chk = eu.PhysicsChecker(family, kwargs, batch)
  for t in range(T):
    a = rng.integers(0, 3, size=batch).astype(np.int32)
    if t > 0:   # teacher forcing: device f32 state := f32(reference-precision f64 state)
      eu.teacher_force(r_env, orc, family)
    ts = env.step(torch.from_numpy(a).cuda())
    want = tuple(x.copy() for x in orc.call(a, t))
    chk.check(eu.to_np(ts), want, eu.oracle_physics_state(orc, family), msg=f'{family} t={t}')
    if t % 20 == 19 or t == T - 1:
      info = env.bsuite_info()
      for k, v in orc.bsuite_info().items():
        np.testing.assert_array_equal(info[k].cpu().numpy()[~chk.tainted], v[~chk.tainted], err_msg=f'{k} t={t}')
  chk.assert_few_ties()
  finished = eu.raw(env).episode_counters().cpu().numpy()[0]
  if family == 'cartpole' or kwargs.get('max_steps') or kwargs.get('init_range'):
    assert finished > batch          # episodes end, auto-reset and restart inside the horizon
  assert chk.max_err['observation'] <= eu.PHYS_TOL


@pytest.mark.parametrize('family,kwargs', [('catch', dict()), ('deep_sea', dict(size=10, deterministic=False, mapping_seed=42)),
                                           ('umbrella_chain', dict(chain_length=5, n_distractor=20)),
                                           ('cartpole', dict())])
def test_sharded_run_equals_unsharded_run(family, kwargs):
  """Multi-GPU oracle: lanes keyed by GLOBAL id => a G-way sharded run reproduces the 1-shard run
  bit for bit (here the shards run one after another on the one GPU gpurun provides)."""
  from bsuite_amd.distributed import shard_lanes
  total, world, T, seed = 3001, 3, 40, 5
  rng = np.random.default_rng(1)
  full = eu.make_env(family, kwargs, batch=total, lane_offset=0, seed=seed)
  shards = [eu.make_env(family, kwargs, batch=n, lane_offset=o, seed=seed)
            for o, n in (shard_lanes(total, r, world) for r in range(world))]
  na = full.action_spec().num_values
  for t in range(T):
    a = torch.from_numpy(rng.integers(0, na, size=total).astype(np.int32)).cuda()
    ref = eu.to_np(full.step(a))
    parts = [eu.to_np(s.step(a[o:o + n])) for s, (o, n) in
             zip(shards, (shard_lanes(total, r, world) for r in range(world)))]
    for j in range(4):
      np.testing.assert_array_equal(eu.f32_bits(ref[j]) if ref[j].dtype == np.float32 else ref[j],
                                    np.concatenate([eu.f32_bits(p[j]) if p[j].dtype == np.float32 else p[j]
                                                    for p in parts]))
  c_full = full.episode_counters().cpu().numpy()
  c_sh = sum(s.episode_counters().cpu().numpy() for s in shards)
  np.testing.assert_array_equal(c_full, c_sh)
