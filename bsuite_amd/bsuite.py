"""Loader surface: `load`, `load_from_id`, the experiment registry (counterpart of
bsuite/bsuite.py:57-108 and the `bsuite/experiments/<name>/<name>.py` load functions).

Every loader accepts the reference's keyword arguments (the sweep settings) plus the engine's
keyword-only extras: `batch` (None = scalar dm_env view, B = vectorised), `device`, `lane_offset`,
`num_buffers`, and — where the reference setting has `seed: None` — an optional `seed` override.

`load_and_record*` (bsuite.py:111-167) wrap the env in the batched `Logging` wrapper with a CSV or
terminal logger: the per-step bookkeeping runs inside the kernels, the CSV
files use bsuite's wire format.
"""
from typing import Any, Mapping, Tuple

from bsuite_amd import sweep
from bsuite_amd.environments import bandit as _bandit
from bsuite_amd.environments import base
from bsuite_amd.environments import cartpole as _cartpole
from bsuite_amd.environments import catch as _catch
from bsuite_amd.environments import deep_sea as _deep_sea
from bsuite_amd.environments import discounting_chain as _discounting_chain
from bsuite_amd.environments import memory_chain as _memory_chain
from bsuite_amd.environments import mnist as _mnist_env
from bsuite_amd.environments import mountain_car as _mountain_car
from bsuite_amd.environments import umbrella_chain as _umbrella_chain
from bsuite_amd.utils import wrappers


def _noisy(make_env):
  """<exp>_noise.load (e.g. experiments/catch_noise/catch_noise.py:23-30)."""
  def load(noise_scale, seed, **kw):
    wrap_seed = kw.pop('wrap_seed', seed)
    env = wrappers.RewardNoise(env=make_env(seed=seed, **kw), noise_scale=noise_scale,
                               seed=wrap_seed)
    return env
  return load


def _scaled(make_env):
  """<exp>_scale.load (e.g. experiments/catch_scale/catch_scale.py:23-30)."""
  def load(reward_scale, seed, **kw):
    return wrappers.RewardScale(env=make_env(seed=seed, **kw), reward_scale=reward_scale,
                                seed=seed)
  return load


def _bandit_env(seed=None, mapping_seed=None, num_actions=11, **kw):
  # experiments/bandit_noise/bandit_noise.py:27-34: the env takes mapping_seed, the wrapper `seed`.
  return _bandit.SimpleBandit(mapping_seed, num_actions=num_actions, seed=seed, **kw)


def _deep_sea_stochastic(size: int, mapping_seed=0, **kw):
  # experiments/deep_sea_stochastic/deep_sea_stochastic.py:22-30
  return _deep_sea.DeepSea(size=size, deterministic=False, mapping_seed=mapping_seed, **kw)


def _memory_len(memory_length: int, seed=0, **kw):
  # experiments/memory_len/memory_len.py:31-39
  return _memory_chain.MemoryChain(memory_length=memory_length, num_bits=1, seed=seed, **kw)


def _memory_size(num_bits: int, seed=0, **kw):
  # experiments/memory_size/memory_size.py:31-39
  return _memory_chain.MemoryChain(memory_length=2, num_bits=num_bits, seed=seed, **kw)


def _umbrella_distract(n_distractor: int, seed=0, **kw):
  # experiments/umbrella_distract/umbrella_distract.py:22-30
  return _umbrella_chain.UmbrellaChain(chain_length=20, n_distractor=n_distractor, seed=seed, **kw)


# Mapping from experiment name to environment constructor or load function (bsuite.py:57-81).
EXPERIMENT_NAME_TO_ENVIRONMENT = dict(
    bandit=_bandit.SimpleBandit,
    bandit_noise=_noisy(_bandit_env),
    bandit_scale=_scaled(_bandit_env),
    cartpole=_cartpole.Cartpole,
    cartpole_noise=_noisy(_cartpole.Cartpole),
    cartpole_scale=_scaled(_cartpole.Cartpole),
    cartpole_swingup=_cartpole.CartpoleSwingup,
    catch=_catch.Catch,
    catch_noise=_noisy(_catch.Catch),
    catch_scale=_scaled(_catch.Catch),
    deep_sea=_deep_sea.DeepSea,
    deep_sea_stochastic=_deep_sea_stochastic,
    discounting_chain=_discounting_chain.DiscountingChain,
    memory_len=_memory_len,
    memory_size=_memory_size,
    mnist=_mnist_env.MNISTBandit,
    mnist_noise=_noisy(_mnist_env.MNISTBandit),
    mnist_scale=_scaled(_mnist_env.MNISTBandit),
    mountain_car=_mountain_car.MountainCar,
    mountain_car_noise=_noisy(_mountain_car.MountainCar),
    mountain_car_scale=_scaled(_mountain_car.MountainCar),
    umbrella_distract=_umbrella_distract,
    umbrella_length=_umbrella_chain.UmbrellaChain,
)


def unpack_bsuite_id(bsuite_id: str) -> Tuple[str, int]:
  """Returns the experiment name and setting index given a bsuite_id (bsuite.py:84-90)."""
  parts = bsuite_id.split(sweep.SEPARATOR)
  assert len(parts) == 2
  return parts[0], int(parts[1])


def load(experiment_name: str, kwargs: Mapping[str, Any], **engine_kwargs) -> base.Environment:
  """Returns a bsuite environment given an experiment name and settings (bsuite.py:93-98)."""
  merged = dict(kwargs)
  merged.update(engine_kwargs)
  env = EXPERIMENT_NAME_TO_ENVIRONMENT[experiment_name](**merged)
  return env


def load_from_id(bsuite_id: str, **engine_kwargs) -> base.Environment:
  """Returns a bsuite environment given a bsuite_id (bsuite.py:101-108).

  engine_kwargs: batch=None|B, device, lane_offset, num_buffers, seed (overrides a `None` seed).
  """
  kwargs = dict(sweep.SETTINGS[bsuite_id])
  experiment_name, _ = unpack_bsuite_id(bsuite_id)
  if engine_kwargs.get('seed') is not None and kwargs.get('seed', None) is not None:
    raise ValueError(f'{bsuite_id} fixes seed={kwargs["seed"]} in its sweep settings')
  if 'seed' in engine_kwargs and engine_kwargs['seed'] is None:
    engine_kwargs.pop('seed')
  env = load(experiment_name, kwargs, **engine_kwargs)
  env.bsuite_num_episodes = sweep.EPISODES[bsuite_id]
  return env


def load_and_record(bsuite_id: str, save_path: str, logging_mode: str = 'csv',
                    overwrite: bool = False, **engine_kwargs):
  """Returns a bsuite environment wrapped with CSV or terminal logging (bsuite.py:111-122)."""
  if logging_mode == 'csv':
    return load_and_record_to_csv(bsuite_id, save_path, overwrite, **engine_kwargs)
  elif logging_mode == 'terminal':
    return load_and_record_to_terminal(bsuite_id, **engine_kwargs)
  else:
    raise ValueError((f'Unrecognised logging_mode "{logging_mode}". '
                      'Must be "csv" or "terminal".'))


def load_and_record_to_csv(bsuite_id: str, results_dir: str, overwrite: bool = False,
                           **engine_kwargs):
  """Returns a bsuite environment that saves results to CSV (bsuite.py:125-159); the files load
  with the reference's `bsuite.logging.csv_load.load_bsuite(results_dir)`."""
  from bsuite_amd.logging import csv_logging  # pylint: disable=import-outside-toplevel
  raw_env = load_from_id(bsuite_id, **engine_kwargs)
  return csv_logging.wrap_environment(env=raw_env, bsuite_id=bsuite_id, results_dir=results_dir,
                                      overwrite=overwrite)


def load_and_record_to_terminal(bsuite_id: str, **engine_kwargs):
  """Returns a bsuite environment that logs to terminal (bsuite.py:162-167)."""
  from bsuite_amd.logging import terminal_logging  # pylint: disable=import-outside-toplevel
  raw_env = load_from_id(bsuite_id, **engine_kwargs)
  return terminal_logging.wrap_environment(raw_env)
