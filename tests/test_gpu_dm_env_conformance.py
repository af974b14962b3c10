"""GPU: the scalar (batch=None) view against the dm_env interface contract.

Restates the checks of dm_env.test_utils.EnvironmentTestMixin that every reference
`bsuite/environments/*_test.py` / `bsuite/experiments/*/*_test.py` runs, on
the same constructor arguments and the same 100-action sequence recipe
(`np.random.RandomState(42).choice(valid_actions)`, e.g. environments/catch_test.py:25-35)."""
import warnings

import numpy as np
import pytest

import bsuite_amd
from bsuite_amd import dm_env_compat as dm_env
from bsuite_amd.environments import (bandit, cartpole, catch, deep_sea, discounting_chain,
                                     memory_chain, mountain_car, umbrella_chain)

pytestmark = pytest.mark.gpu


def _envs():
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    return {
        'deep_sea_10': lambda: deep_sea.DeepSea(10),                                  # deep_sea_test.py:27
        'deep_sea_5_stochastic': lambda: deep_sea.DeepSea(5, deterministic=False),    # deep_sea_test.py:40
        'catch': lambda: catch.Catch(rows=10, columns=5),                             # catch_test.py:28
        'cartpole': lambda: cartpole.Cartpole(seed=22),                               # cartpole_test.py:28
        'mountain_car': lambda: mountain_car.MountainCar(2),                          # mountain_car_test.py:27
        'memory_chain': lambda: memory_chain.MemoryChain(memory_length=10, num_bits=3),
        'umbrella_chain': lambda: umbrella_chain.UmbrellaChain(chain_length=20, n_distractor=22),
        'bandit': lambda: bandit.SimpleBandit(5),                                     # bandit_test.py:27
        'discounting_chain': lambda: discounting_chain.DiscountingChain(10),
        'bandit_noise': lambda: bsuite_amd.load('bandit_noise', dict(noise_scale=1., seed=42, mapping_seed=42)),
        'catch_scale': lambda: bsuite_amd.load('catch_scale', dict(reward_scale=10., seed=22)),
        'deep_sea_stochastic': lambda: bsuite_amd.load('deep_sea_stochastic', dict(size=22)),
        'cartpole_swingup': lambda: cartpole.CartpoleSwingup(seed=42),
        'catch/0': lambda: bsuite_amd.load_from_id('catch/0'),
    }


@pytest.mark.parametrize('name', sorted(_envs()))
def test_dm_env_interface(name):
  env = _envs()[name]()
  obs_spec, act_spec = env.observation_spec(), env.action_spec()
  reward_spec, discount_spec = env.reward_spec(), env.discount_spec()
  valid = list(range(act_spec.num_values))
  actions = np.random.RandomState(42).choice(valid, 100)

  def check(ts, must_be_first=False):
    assert isinstance(ts, dm_env.TimeStep)
    if must_be_first:
      assert ts.first()
    obs_spec.validate(ts.observation)
    assert isinstance(ts.observation, np.ndarray)
    if ts.first():
      assert ts.reward is None and ts.discount is None
    else:
      reward_spec.validate(np.asarray(ts.reward, dtype=float))
      discount_spec.validate(np.asarray(ts.discount, dtype=float))
      assert isinstance(ts.reward, float) and ts.discount in (0.0, 1.0)
      if ts.last():
        assert ts.discount == 0.0
    return ts

  ts = check(env.step(int(actions[0])), must_be_first=True)      # step on a fresh env == reset
  check(env.reset(), must_be_first=True)
  was_last, n_last = False, 0
  for a in actions:
    ts = check(env.step(int(a)), must_be_first=was_last)          # step after LAST restarts
    was_last = ts.last()
    n_last += was_last
  check(env.reset(), must_be_first=True)
  info = env.bsuite_info()
  assert isinstance(info, dict) and all(isinstance(v, (int, float)) for v in info.values())
  assert env.bsuite_num_episodes > 0
  if name in ('bandit', 'bandit_noise', 'deep_sea_5_stochastic', 'mountain_car', 'catch'):
    assert n_last > 0


def test_invalid_actions_raise_like_the_reference():
  with pytest.raises(IndexError):
    c = catch.Catch()
    c.reset()
    c.step(3)                              # catch.py:84 `_ACTIONS[action]`
  with pytest.raises(IndexError):
    b = bandit.SimpleBandit(0)
    b.reset()
    b.step(11)                             # bandit.py:61 `self._rewards[action]`
  d = deep_sea.DeepSea(6, mapping_seed=1)
  d.reset()
  assert d.step(7).mid()                   # deep_sea accepts any int: != mapping => left (:118)
  with pytest.raises(NotImplementedError):
    d._step(0)
  # a step that auto-resets never looks at its action (base.py:59-62): nothing is raised there
  b = bandit.SimpleBandit(0)
  assert b.step(99).first()                # fresh environment
  assert b.step(3).last()
  assert b.step(99).first()                # after LAST


def test_discounting_chain_indexes_like_the_reference():
  """discounting_chain.py:76-81: the episode's first action becomes the context and indexes two Python lists at every
  step of the episode — IndexError outside -5..4 on that first step, -5..-1 wrap (the observation keeps the negative
  context), later actions are never looked at."""
  for bad in (5, 100, -6):
    dc = discounting_chain.DiscountingChain(mapping_seed=0)
    dc.reset()
    with pytest.raises(IndexError):
      dc.step(bad)
  dc = discounting_chain.DiscountingChain(mapping_seed=0)
  dc.reset()
  assert dc.step(2).mid()
  assert dc.step(99).mid() and dc.step(-77).mid()          # only the first action of an episode matters
  pay, bonus_chain = [1, 3, 10, 30, 100], 4 % 5
  for first in (-1, -2, -5, 4, 0):
    dc = discounting_chain.DiscountingChain(mapping_seed=4)
    dc.reset()
    chain = first % 5
    for t in range(1, 101):
      ts = dc.step(first if t == 1 else 3)
      assert ts.observation[0, 0] == np.float32(first) and ts.observation[0, 1] == np.float32(t / 100)
      want = (1.1 if chain == bonus_chain else 1.0) if t == pay[chain] else 0.0
      assert ts.reward == want, (first, t, ts.reward)
      assert ts.last() == (t == 100)
    assert dc.step(5).first()                               # auto-reset: the action is not looked at
  many = discounting_chain.DiscountingChain(mapping_seed=4, batch=64)
  import torch
  many.step(torch.zeros(64, dtype=torch.int32, device='cuda'))
  many.step(torch.arange(-32, 32, dtype=torch.int32, device='cuda'))
  assert int(many.invalid_action_count().item()) == 64 - 10       # -5..4 are the reference's legal first actions


def test_scalar_view_is_lane_zero_of_the_batched_view():
  import torch
  seed, T = 77, 60
  one = catch.Catch(seed=seed)
  many = catch.Catch(seed=seed, batch=64)
  acts = np.random.RandomState(1).choice(3, T)
  for a in acts:
    ts1 = one.step(int(a))
    tsb = many.step(torch.full((64,), int(a), dtype=torch.int32, device='cuda'))
    assert int(ts1.step_type) == int(tsb.step_type[0].item())
    np.testing.assert_array_equal(ts1.observation, tsb.observation[0].cpu().numpy())
    if not ts1.first():
      assert np.float32(ts1.reward) == tsb.reward[0].item()
  assert one.bsuite_info()['total_regret'] == many.bsuite_info()['total_regret'][0].item()


def test_wrappers_keep_the_reference_surface():
  env = bsuite_amd.load_from_id('catch_noise/3', seed=5)
  assert env.raw_env is not env and type(env.raw_env).__name__ == 'Catch'
  assert env.bsuite_info() == env.raw_env.bsuite_info()
  assert env._rows == 10                                     # attribute delegation (wrappers.py:308-310)
  with pytest.raises(NotImplementedError):
    env._step(0)
  ts = env.reset()
  assert ts.first() and ts.reward is None
  rewards = []
  for _ in range(30):
    ts = env.step(1)
    if not ts.first():
      rewards.append(ts.reward)
  assert len(set(rewards)) > 5                               # noise_scale 0.1 perturbs every reward


def test_reward_wrappers_stack_like_the_reference():
  """utils/wrappers_test.py:123-131 (`test_unwrap`): RewardNoise(RewardScale(env)) under Logging unwraps to
  the raw environment; both stacking orders fold into the one fused epilogue; a second wrapper of the
  same kind is refused (the epilogue holds one of each)."""
  from bsuite_amd.utils import wrappers
  raw_env = catch.Catch(seed=3)
  scale_env = wrappers.RewardScale(raw_env, reward_scale=30.)
  noise_env = wrappers.RewardNoise(scale_env, noise_scale=1., seed=3)
  logging_env = wrappers.Logging(noise_env, logger=None)
  assert logging_env.raw_env is raw_env and noise_env.raw_env is raw_env
  plain = catch.Catch(seed=3)
  only_noise = wrappers.RewardNoise(catch.Catch(seed=3), noise_scale=1., seed=3)
  rs = np.random.RandomState(0)
  for _ in range(40):
    a = int(rs.randint(3))
    ts, tp, tn = logging_env.step(a), plain.step(a), only_noise.step(a)
    if not ts.first():
      z = tn.reward - tp.reward                      # the wrapper stream's draw of this call
      assert ts.reward == tp.reward * 30. + 1. * z   # r*s + sigma*z, in f64 like the reference
  other = wrappers.RewardScale(wrappers.RewardNoise(catch.Catch(seed=3), noise_scale=1., seed=3), reward_scale=30.)
  plain2, noise2 = catch.Catch(seed=3), wrappers.RewardNoise(catch.Catch(seed=3), noise_scale=1., seed=3)
  for _ in range(40):
    a = int(rs.randint(3))
    ts, tp, tn = other.step(a), plain2.step(a), noise2.step(a)
    if not ts.first():
      assert ts.reward == tn.reward * 30.            # (r + sigma*z) * s
  with pytest.raises(NotImplementedError):
    wrappers.RewardScale(scale_env, reward_scale=2.)
  with pytest.raises(NotImplementedError):
    wrappers.RewardNoise(noise_env, noise_scale=2.)
