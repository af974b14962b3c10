"""GPU: EVERY bsuite_id of sweep.SWEEP (468) loaded through `load_from_id` and stepped against the C
oracle configured independently from the reference's experiment files (bsuite/experiments/*/:
which constructor, which keyword, which wrapper).  Integer / grid families bit-exact, physics
families teacher-forced at 1e-6.  Pins the loader registry + sweep settings + kernels end to end."""
import numpy as np
import pytest
import torch

import bsuite_amd
from bsuite_amd import sweep
from oracle import coracle
from tests import engine_util as eu
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

PHYSICS = ('cartpole', 'cartpole_swingup', 'mountain_car')


def oracle_config(bsuite_id):
  """(family, kwargs, wrap, fixed_seed) as the reference's experiment loader would build it."""
  name = bsuite_id.split('/')[0]
  st = dict(sweep.SETTINGS[bsuite_id])
  wrap = None
  base = name
  if name.endswith('_noise'):
    base, wrap = name[:-6], ('noise', st.pop('noise_scale'))
  elif name.endswith('_scale'):
    base, wrap = name[:-6], ('scale', st.pop('reward_scale'))
  fixed_seed = st.pop('seed', None)                              # <exp>_noise/_scale fix the seed (e.g. catch_noise.py:23-30)
  if base == 'bandit':
    return 'bandit', dict(mapping_seed=st['mapping_seed']), wrap, fixed_seed   # experiments/bandit/sweep.py:20
  if base == 'catch':
    return 'catch', {}, wrap, fixed_seed
  if base == 'deep_sea':
    return 'deep_sea', dict(size=st['size'], mapping_seed=st['mapping_seed']), wrap, fixed_seed
  if base == 'deep_sea_stochastic':                              # deep_sea_stochastic.py:22-30
    return 'deep_sea', dict(size=st['size'], mapping_seed=st['mapping_seed'], deterministic=False), wrap, fixed_seed
  if base == 'discounting_chain':
    return 'discounting_chain', dict(mapping_seed=st['mapping_seed']), wrap, fixed_seed
  if base == 'memory_len':                                       # memory_len.py:31-39 (seed=0)
    return 'memory_chain', dict(memory_length=st['memory_length'], num_bits=1), wrap, 0
  if base == 'memory_size':                                      # memory_size.py:31-39 (seed=0)
    return 'memory_chain', dict(memory_length=2, num_bits=st['num_bits']), wrap, 0
  if base == 'umbrella_length':                                  # experiments/umbrella_length/sweep.py:26
    return 'umbrella_chain', dict(chain_length=st['chain_length'], n_distractor=st['n_distractor']), wrap, fixed_seed
  if base == 'umbrella_distract':                                # umbrella_distract.py:22-30 (seed=0)
    return 'umbrella_chain', dict(chain_length=20, n_distractor=st['n_distractor']), wrap, 0
  if base == 'cartpole':
    return 'cartpole', {}, wrap, fixed_seed
  if base == 'cartpole_swingup':                                 # experiments/cartpole_swingup/sweep.py:22-23
    return 'cartpole_swingup', dict(height_threshold=st['height_threshold'],
                                    x_reward_threshold=st['x_reward_threshold']), wrap, fixed_seed
  if base == 'mountain_car':
    return 'mountain_car', {}, wrap, fixed_seed
  if base == 'mnist':
    return 'mnist', {}, wrap, fixed_seed
  raise KeyError(bsuite_id)


@pytest.mark.parametrize('chunk', range(12))
def test_every_bsuite_id_matches_the_oracle(chunk):
  ids = [b for j, b in enumerate(sweep.SWEEP) if j % 12 == chunk]
  images, labels = gu.mnist_dataset()
  B, off = 37, 1000
  for bid in ids:
    fam, kw, wrap, fixed = oracle_config(bid)
    seed = 4321 if fixed is None else fixed
    ekw = dict(seed=seed) if fixed is None else {}            # unseeded settings: the seed is ours to choose
    okw = dict(kw)
    if fam == 'mnist':
      ekw.update(images=images, labels=labels)
      okw.update(images=images, labels=labels)
    env = bsuite_amd.load_from_id(bid, batch=B, lane_offset=off, num_buffers=1, **ekw)
    assert env.bsuite_num_episodes == sweep.EPISODES[bid]
    raw = eu.raw(env)
    orc = coracle.OracleEnv(fam, okw, np.arange(off, off + B, dtype=np.uint64), seed=seed, wrap=wrap)
    assert tuple(env.observation_spec().shape) == tuple(orc.obs_shape), bid
    assert env.action_spec().num_values == orc.num_actions, bid
    rng = np.random.default_rng(len(bid))
    if fam in PHYSICS:
      _physics_id(bid, fam, kw, env, raw, orc, rng, B)
      continue
    # at least one FULL episode plus the auto-reset after it
    horizon = dict(deep_sea=kw.get('size', 0) + 3, memory_chain=kw.get('memory_length', 0) + 4,
                   umbrella_chain=kw.get('chain_length', 0) + 3, discounting_chain=103).get(fam, 14)
    for t in range(max(14, horizon)):
      a = rng.integers(0, orc.num_actions, size=B).astype(np.int32)
      ts = env.step(torch.from_numpy(a).cuda())
      st, r, d, o = orc.call(a, t)
      gst, gr, gd, go = eu.to_np(ts)
      live = st != 0
      np.testing.assert_array_equal(gst, st, err_msg=f'{bid} t={t}')
      np.testing.assert_array_equal(eu.f32_bits(go), eu.f32_bits(o), err_msg=f'{bid} obs t={t}')
      np.testing.assert_array_equal(eu.f32_bits(gr[live]), eu.f32_bits(r[live].astype(np.float32)),
                                    err_msg=f'{bid} reward t={t}')
    info = raw.bsuite_info()
    for k_, v in orc.bsuite_info().items():
      np.testing.assert_array_equal(info[k_].cpu().numpy(), v, err_msg=f'{bid} {k_}')


def _physics_id(bid, fam, kw, env, raw, orc, rng, B):
  """Physics ids run, teacher-forced, until EVERY lane has finished an episode and been auto-reset
  (cartpole ~10^2 calls; swing-up / mountain_car 1002: their random-policy episodes time out): all
  observation components (swing-up's two sign flags included), rewards through the noise / scale
  wrappers, termination, restart, and raw_return / best_episode / total_upright at the end."""
  chk = eu.PhysicsChecker(fam, kw, B)
  finished = np.zeros(B, bool)
  restarted = np.zeros(B, bool)
  for t in range(1100):
    a = rng.integers(0, orc.num_actions, size=B).astype(np.int32)
    if t > 0:
      eu.teacher_force(raw, orc, fam)
    ts = env.step(torch.from_numpy(a).cuda())
    st, r, d, o = orc.call(a, t)
    chk.check(eu.to_np(ts), (st, r, d, o), eu.oracle_physics_state(orc, fam), msg=f'{bid} t={t}')
    restarted |= finished & (st == 0)
    finished |= st == 2
    if restarted.all():
      break
  assert restarted.all(), bid
  chk.assert_few_ties()
  info = env.bsuite_info()
  clean = ~chk.tainted
  for k_, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(info[k_].cpu().numpy()[clean], v[clean], err_msg=f'{bid} {k_}')


def test_oracle_config_covers_the_sweep():
  fams = {oracle_config(b)[0] for b in sweep.SWEEP}
  assert fams == {'bandit', 'catch', 'deep_sea', 'discounting_chain', 'memory_chain', 'umbrella_chain',
                  'cartpole', 'cartpole_swingup', 'mountain_car', 'mnist'}
  assert len(sweep.SWEEP) == 468
