"""Helpers for the GPU parity tests: build bsuite_amd environments from golden-case metadata."""
import numpy as np
import torch

from bsuite_amd.environments import (bandit, cartpole, catch, deep_sea, discounting_chain,
                                     memory_chain, mnist, mountain_car, umbrella_chain)
from bsuite_amd.utils import wrappers

CTORS = dict(
    deep_sea=deep_sea.DeepSea, catch=catch.Catch, bandit=bandit.SimpleBandit,
    memory_chain=memory_chain.MemoryChain, umbrella_chain=umbrella_chain.UmbrellaChain,
    discounting_chain=discounting_chain.DiscountingChain, cartpole=cartpole.Cartpole,
    cartpole_swingup=cartpole.CartpoleSwingup, mountain_car=mountain_car.MountainCar,
    mnist=mnist.MNISTBandit)


def make_env(family, kwargs, batch, lane_offset, seed, wrap=None, num_buffers=1, **engine_kwargs):
  import warnings
  kw = dict(kwargs)
  kw.pop('seed', None)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    env = CTORS[family](**kw, seed=seed, batch=batch, lane_offset=lane_offset,
                        num_buffers=num_buffers, **engine_kwargs)
  if wrap:
    env = apply_wrap(env, wrap, seed)
  return env


def apply_wrap(env, wrap, seed):
  """wrap: ('noise', sigma) | ('scale', s) | ('scale_noise', s, sigma) = RewardNoise(RewardScale(env)) |
  ('noise_scale', sigma, s) = RewardScale(RewardNoise(env)) — the reference composes them freely."""
  kind = wrap[0]
  if kind == 'noise':
    return wrappers.RewardNoise(env, noise_scale=wrap[1], seed=seed)
  if kind == 'scale':
    return wrappers.RewardScale(env, reward_scale=wrap[1], seed=seed)
  if kind == 'scale_noise':
    return wrappers.RewardNoise(wrappers.RewardScale(env, reward_scale=wrap[1], seed=seed), noise_scale=wrap[2], seed=seed)
  if kind == 'noise_scale':
    return wrappers.RewardScale(wrappers.RewardNoise(env, noise_scale=wrap[1], seed=seed), reward_scale=wrap[2], seed=seed)
  raise KeyError(kind)


def raw(env):
  return env.raw_env if hasattr(env, 'raw_env') else env


def to_np(ts):
  return (ts.step_type.cpu().numpy(), ts.reward.cpu().numpy(), ts.discount.cpu().numpy(),
          ts.observation.cpu().numpy())


def f32_bits(x):
  return np.ascontiguousarray(x, np.float32).view(np.uint32)


# ---------------------------------------------------------------------------------------------------
# Physics families (cartpole, swing-up, mountain_car): the contract is per step, teacher-forced from
# the reference state, |a - b| <= 1e-6 * max(1, |b|) (BASELINE north_star / SURVEY §8c).
PHYS_TOL = 1e-6


def within_tol(got, want):
  """Elementwise |a-b| <= 1e-6*max(1,|b|) and the error in units of that bound's scale."""
  got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
  scale = np.maximum(1.0, np.abs(want))
  err = np.abs(got - want) / scale
  return err <= PHYS_TOL, err


def assert_within_tol(got, want, err_msg=''):
  ok, err = within_tol(got, want)
  assert ok.all(), f'{err_msg}: max |a-b|/max(1,|b|) = {err.max():.3e} > {PHYS_TOL:g} at {np.argwhere(~ok)[:5].tolist()}'
  return float(err.max()) if err.size else 0.0


def physics_ties(family, cfg, x=None, theta=None, theta_dot=None, position=None):
  """Per lane: does a quantity that decides step_type / reward / sign flags sit within the tolerance
  of its threshold (reference f64 state AFTER the call)?  Only then may the f32 engine land on the
  other side: cos(theta) vs height_threshold, |x| vs x_threshold (cartpole.py:143-146), swing-up's
  |theta_dot| vs theta_dot_threshold and |x| vs x_reward_threshold (cartpole_swingup.py:104-123,
  147-149), mountain_car's position vs the 0.5 goal (mountain_car.py:88)."""
  def near(q, thr):
    q = np.asarray(q, np.float64)
    return np.abs(q - thr) <= PHYS_TOL * np.maximum(1.0, np.abs(q))
  if family == 'mountain_car':
    return near(position, 0.5)
  tie = near(np.cos(theta), cfg['height_threshold']) | near(np.abs(x), cfg['x_threshold'])
  if family == 'cartpole_swingup':
    tie |= near(np.abs(theta_dot), cfg['theta_dot_threshold']) | near(np.abs(x), cfg['x_reward_threshold'])
  return tie


def cartpole_cfg(family, kwargs):
  sw = family == 'cartpole_swingup'
  return dict(height_threshold=kwargs.get('height_threshold', 0.5 if sw else 0.8),
              x_threshold=kwargs.get('x_threshold', 3.), theta_dot_threshold=kwargs.get('theta_dot_threshold', 1.),
              x_reward_threshold=kwargs.get('x_reward_threshold', 1.))


class PhysicsChecker:
  """Compares engine TimeSteps with reference ones call by call.  The CONTINUOUS observation components
  (x, x_dot, sin, cos, theta_dot, time | position, velocity, time) must be within 1e-6*max(1,|b|) on EVERY
  lane and call — a threshold tie waives nothing there.  On a call where `physics_ties` holds for a lane —
  a deciding quantity within the tolerance of its threshold in the reference's f64 state — that lane may
  disagree on what the threshold decides, and only that: step_type (and the discount that follows from
  it), the reward, and swing-up's two sign-flag observation elements (obs[6], obs[7]:
  cartpole_swingup.py:147-149).  Lanes that ever tied are `tainted`: their bsuite_info accumulators
  legitimately differ from then on."""

  SIGN_FLAGS = {'cartpole_swingup': (6, 7)}

  def __init__(self, family, kwargs, batch):
    self.family, self.cfg = family, cartpole_cfg(family, kwargs)
    self.tainted = np.zeros(batch, bool)
    self.max_err = dict(observation=0.0, reward=0.0)
    self.calls = 0
    self.ties = 0

  def check(self, got, want, state, msg=''):
    """got / want: (step_type, reward, discount, obs) numpy; state: dict of the reference's f64
    post-call state (x, theta, theta_dot | position)."""
    gst, gr, gd, go = got
    st, r, d, o = want
    B = len(st)
    tie = physics_ties(self.family, self.cfg, **state)
    ok_obs, err_obs = within_tol(go, o)
    ok_obs, err_obs = ok_obs.reshape(B, -1), err_obs.reshape(B, -1)
    flag = np.zeros(ok_obs.shape[1], bool)
    flag[list(self.SIGN_FLAGS.get(self.family, ()))] = True
    # never waivable: the continuous components
    cont_ok = ok_obs[:, ~flag].all(axis=1)
    assert cont_ok.all(), (f'{msg}: lanes {np.flatnonzero(~cont_ok)[:5].tolist()}: continuous observation components beyond '
                           f'1e-6*max(1,|b|): max err {err_obs[:, ~flag].max():.3e} (a threshold tie does not waive these)')
    live = st != 0
    ok_r, err_r = within_tol(gr, r)
    ok_r |= ~live                              # FIRST: reward is None in the reference
    ok_d = (gd == np.where(live, d, 1.0).astype(np.float32)) | (gst != st)
    decided_ok = ok_obs[:, flag].all(axis=1) & ok_r & (gst == st) & ok_d
    bad = ~decided_ok & ~tie
    assert not bad.any(), (f'{msg}: lanes {np.flatnonzero(bad)[:5].tolist()} differ in step_type / reward / sign flag without a '
                           f'threshold tie; step_type {gst[bad][:5]} vs {st[bad][:5]}, reward {gr[bad][:5]} vs {r[bad][:5]}')
    self.ties += int((~decided_ok).sum())
    self.tainted |= ~decided_ok
    self.max_err['observation'] = max(self.max_err['observation'], float(err_obs[:, ~flag].max()))
    good = decided_ok & live
    if good.any():
      self.max_err['reward'] = max(self.max_err['reward'], float(err_r[good].max()))
    self.calls += 1

  def assert_few_ties(self, max_fraction=1e-5):
    """Measured tie rate: 0 in 6.5e6 lane-steps per family (profiles/r02/physics_error.json)."""
    n = self.calls * len(self.tainted)
    assert self.ties <= max(1, int(max_fraction * n)), f'{self.ties} threshold ties in {n} lane-steps'


def oracle_physics_state(orc, family):
  if family == 'mountain_car':
    return dict(position=orc.s['position'].copy())
  s = orc.s['state']
  return dict(x=s[:, 0].copy(), theta=s[:, 2].copy(), theta_dot=s[:, 3].copy())


def teacher_force(raw_env, orc, family, lanes=None):
  """device f32 state := f32(reference f64 state); step counter and reset flag := the reference's.  `lanes`: an index
  tensor — only those lanes of the environment (the oracle then holds exactly them, in that order)."""
  if family == 'mountain_car':
    st32 = np.stack([orc.s['position'], orc.s['velocity']]).astype(np.float32)
    k = orc.s['timestep'].astype(np.int32)
  else:
    st32 = orc.s['state'][:, :4].T.astype(np.float32)
    k = np.rint(orc.s['state'][:, 4] / orc.cfg.timescale).astype(np.int32)
  st_t = torch.from_numpy(np.ascontiguousarray(st32)).cuda()
  k_t = torch.from_numpy(k | (orc.reset_next.astype(np.int32) << 30)).cuda()
  if lanes is None:
    raw_env._state['state'].copy_(st_t)
    raw_env._state['steps'].copy_(k_t)
  else:
    raw_env._state['state'][:, lanes] = st_t
    raw_env._state['steps'][lanes] = k_t


def _force_physics_state(env, fam, phys_prev, idx):
  r = raw(env)
  if fam == 'mountain_car':
    st = np.stack([phys_prev[idx, 0], phys_prev[idx, 1]]).astype(np.float32)
  else:
    st = phys_prev[idx, :4].T.astype(np.float32)
  r._state['state'].copy_(torch.from_numpy(np.ascontiguousarray(st)).to(r.device))


def check_against_case(name, meta, g):
  """Steps the engine through one recorded case of the reference (a tests/golden fixture, or one recorded live by
  tests/test_gpu_vs_reference_live.py): `meta`, `g` as oracle/make_golden.run_case writes them.  Integer / grid
  families bit-exact; physics families teacher-forced per step at 1e-6*max(1,|b|)."""
  from tests import golden_util as gu
  fam = meta['family']
  phys = fam in gu.PHYSICS
  wrap = tuple(meta['wrap']) if meta['wrap'] else None
  T = g['actions'].shape[0]
  kwargs = dict(meta['kwargs'])
  if fam == 'mnist':
    kwargs['images'], kwargs['labels'] = gu.mnist_dataset()
  for (i0, lane0, n) in gu.contiguous_runs(g['lanes']):
    idx = slice(i0, i0 + n)
    if meta.get('bsuite_id'):     # recorded from the reference's own load_from_id: ours builds the engine side
      import bsuite_amd
      import warnings
      ekw = dict(images=kwargs['images'], labels=kwargs['labels']) if fam == 'mnist' else {}
      if meta.get('seed_is_ours', True):
        ekw['seed'] = meta['seed']
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        env = bsuite_amd.load_from_id(meta['bsuite_id'], batch=n, lane_offset=lane0, num_buffers=1, **ekw)
    else:
      env = make_env(fam, kwargs, batch=n, lane_offset=lane0, seed=meta['seed'], wrap=wrap)
    raw(env)._step_index = meta['step0']
    logged = None
    if meta.get('log'):
      from bsuite_amd.utils import wrappers
      env = logged = wrappers.Logging(env, None, log_by_step=meta['log'] == 'by_step',
                                      log_every=meta['log'] == 'every', max_rows=g['log_rows'].shape[1] + 3)
    for t in range(T):
      if phys and t > 0:
        _force_physics_state(env, fam, g['phys'][t - 1], idx)
      if t in meta['reset_at']:
        ts = env.reset()
      else:
        ts = env.step(torch.from_numpy(g['actions'][t, idx]).to('cuda'))
      st, r, d, o = to_np(ts)
      gst, gr, gd, go = g['step_type'][t, idx], g['reward'][t, idx], g['discount'][t, idx], g['obs'][t, idx]
      np.testing.assert_array_equal(st, gst, err_msg=f'{name} step_type t={t}')
      first = gst == 0
      assert (r[first] == 0).all() and (d[first] == 1).all()
      np.testing.assert_array_equal(d[~first], gd[~first].astype(np.float32))
      if phys:      # |a-b| <= 1e-6*max(1,|b|), the north_star bound (not rtol+atol = 2e-6)
        assert_within_tol(o, go, err_msg=f'{name} obs t={t}')
        assert_within_tol(r[~first], gr[~first], err_msg=f'{name} reward t={t}')
      else:
        np.testing.assert_array_equal(f32_bits(r[~first]), f32_bits(gr[~first].astype(np.float32)),
                                      err_msg=f'{name} reward t={t}')
        np.testing.assert_array_equal(f32_bits(o), f32_bits(go), err_msg=f'{name} obs t={t}')
      info = env.bsuite_info()
      for j, k in enumerate(meta['info_keys']):
        got = info[k].cpu().numpy()
        if phys:
          np.testing.assert_allclose(got, g['info'][t, idx, j], rtol=1e-9, atol=1e-9, err_msg=f'{k} t={t}')
        else:
          np.testing.assert_array_equal(got, g['info'][t, idx, j], err_msg=f'{name} {k} t={t}')
    if logged is not None:   # rows the unmodified reference Logging wrapper wrote, per lane
      assert list(raw(env).logging_columns()[:5]) == meta['log_columns'][:5]
      cols = [meta['log_columns'].index(c) for c in raw(env).logging_columns() if not c.startswith('_')]
      keep = [j for j, c in enumerate(raw(env).logging_columns()) if not c.startswith('_')]
      n_rows = logged.num_rows().cpu().numpy()
      np.testing.assert_array_equal(n_rows, g['log_n_rows'][idx], err_msg=f'{name} n_rows')
      rows = logged._lg['rows'].cpu().numpy()
      for l in range(n):
        want = g['log_rows'][i0 + l, :n_rows[l]][:, cols]
        got = rows[l, :n_rows[l]][:, keep]
        if phys:
          np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9, err_msg=f'{name} lane {l}')
        else:
          np.testing.assert_array_equal(got, want, err_msg=f'{name} lane {l}')
