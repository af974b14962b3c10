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
    kind, param = wrap
    if kind == 'noise':
      env = wrappers.RewardNoise(env, noise_scale=param, seed=seed)
    else:
      env = wrappers.RewardScale(env, reward_scale=param, seed=seed)
  return env


def raw(env):
  return env.raw_env if hasattr(env, 'raw_env') else env


def to_np(ts):
  return (ts.step_type.cpu().numpy(), ts.reward.cpu().numpy(), ts.discount.cpu().numpy(),
          ts.observation.cpu().numpy())


def f32_bits(x):
  return np.ascontiguousarray(x, np.float32).view(np.uint32)
