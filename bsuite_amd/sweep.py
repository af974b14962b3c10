"""The bsuite sweep: every bsuite_id, its environment settings, tags and episode budget.

Value-for-value counterpart of bsuite/sweep.py:108-150 (SETTINGS, SWEEP, TAGS, TESTING, EPISODES and
the per-experiment tuples) and of the 23 `bsuite/experiments/<name>/sweep.py` files it is assembled
from (cited per row).  It is data, not code: 468 ids across 23 experiments; tests pin it against
tests/golden/sweep.json, which was dumped from the reference.
"""
import types
from typing import Any, Dict, Mapping, Tuple

# Common type aliases (sweep.py:65-67).
BSuiteId = str
Tag = str
EnvKWargs = Dict[str, Any]

SEPARATOR = '/'                              # sweep.py:70
IGNORE_FOR_TESTING = ('_noise', '_scale')    # sweep.py:72

_LOG_SPACED_100 = tuple(range(1, 11)) + (12, 14, 17, 20, 25) + tuple(range(30, 105, 10))
_LOG_SPACED_40 = tuple(range(1, 11)) + (12, 14, 17, 20, 25) + tuple(range(30, 50, 10))
_NOISE_SCALES = (0.1, 0.3, 1.0, 3., 10.)
_REWARD_SCALES = (0.001, 0.03, 1.0, 30., 1000.)


def _noise(extra=None, key='noise_scale', scales=_NOISE_SCALES):
  out = []
  for scale in scales:
    for n in range(4):
      setting = {key: scale, 'seed': None}
      if extra:
        setting[extra] = n
      out.append(setting)
  return tuple(out)


def _scale(extra=None):
  return _noise(extra, 'reward_scale', _REWARD_SCALES)


# (experiment name, NUM_EPISODES, SETTINGS, TAGS) in the order bsuite/sweep.py:108-131 parses them.
_EXPERIMENTS = (
    # experiments/bandit/sweep.py:19-22
    ('bandit', 10000, tuple({'mapping_seed': n} for n in range(20)), ('basic',)),
    # experiments/bandit_noise/sweep.py:20-29
    ('bandit_noise', 10000, _noise('mapping_seed'), ('noise',)),
    # experiments/bandit_scale/sweep.py:20-29
    ('bandit_scale', 10000, _scale('mapping_seed'), ('scale',)),
    # experiments/cartpole/sweep.py:19-22
    ('cartpole', 1000, tuple({'seed': None} for _ in range(20)),
     ('basic', 'credit_assignment', 'generalization')),
    ('cartpole_noise', 1000, _noise(), ('noise', 'generalization')),
    ('cartpole_scale', 1000, _scale(), ('scale', 'generalization')),
    # experiments/cartpole_swingup/sweep.py:20-25
    ('cartpole_swingup', 1000,
     tuple({'height_threshold': n / 20, 'x_reward_threshold': 1 - n / 20} for n in range(20)),
     ('exploration', 'generalization')),
    # experiments/catch/sweep.py:19-22
    ('catch', 10000, tuple({'seed': None} for _ in range(20)), ('basic', 'credit_assignment')),
    ('catch_noise', 10000, _noise(), ('noise', 'credit_assignment')),
    ('catch_scale', 10000, _scale(), ('scale', 'credit_assignment')),
    # experiments/deep_sea/sweep.py:19-22
    ('deep_sea', 10000, tuple({'size': n, 'mapping_seed': 42} for n in range(10, 51, 2)),
     ('exploration',)),
    ('deep_sea_stochastic', 10000,
     tuple({'size': n, 'mapping_seed': 42} for n in range(10, 51, 2)), ('exploration', 'noise')),
    # experiments/discounting_chain/sweep.py:19-22
    ('discounting_chain', 1000, tuple({'mapping_seed': n} for n in range(20)),
     ('credit_assignment',)),
    # experiments/memory_len/sweep.py:19-27
    ('memory_len', 10000, tuple({'memory_length': n} for n in _LOG_SPACED_100), ('memory',)),
    # experiments/memory_size/sweep.py:21-29
    ('memory_size', 10000, tuple({'num_bits': n} for n in _LOG_SPACED_40), ('memory',)),
    # experiments/mnist/sweep.py:19-22
    ('mnist', 10000, tuple({'seed': None} for _ in range(20)), ('basic', 'generalization')),
    ('mnist_noise', 10000, _noise(), ('noise', 'generalization')),
    ('mnist_scale', 10000, _scale(), ('scale', 'generalization')),
    # experiments/mountain_car/sweep.py:19-22
    ('mountain_car', 1000, tuple({'seed': None} for _ in range(20)), ('basic', 'generalization')),
    ('mountain_car_noise', 1000, _noise(), ('noise', 'generalization')),
    ('mountain_car_scale', 1000, _scale(), ('scale', 'generalization')),
    # experiments/umbrella_distract/sweep.py:21-29
    ('umbrella_distract', 10000, tuple({'n_distractor': n} for n in _LOG_SPACED_100),
     ('credit_assignment', 'noise')),
    # experiments/umbrella_length/sweep.py:19-27
    ('umbrella_length', 10000,
     tuple({'chain_length': n, 'n_distractor': 20} for n in _LOG_SPACED_100),
     ('credit_assignment', 'noise')),
)

_settings, _sweep, _tags, _testing, _episodes = {}, [], {}, [], {}
for _name, _num_episodes, _exp_settings, _exp_tags in _EXPERIMENTS:
  _ids = tuple(f'{_name}{SEPARATOR}{i}' for i in range(len(_exp_settings)))
  for _id, _setting in zip(_ids, _exp_settings):
    _settings[_id] = _setting
    _episodes[_id] = _num_episodes
  if not _name.endswith(IGNORE_FOR_TESTING):
    _testing.append(_ids[0])
  for _tag in _exp_tags:
    _tags.setdefault(_tag, []).extend(_ids)
  _sweep.extend(_ids)
  globals()[_name.upper()] = _ids          # BANDIT, BANDIT_NOISE, ... (sweep.py:108-131)

# Mapping from bsuite id to keyword arguments for the corresponding environment (read-only).
SETTINGS: Mapping[BSuiteId, EnvKWargs] = types.MappingProxyType(_settings)
# Tuple containing all bsuite_ids. Used for hyperparameter sweeps.
SWEEP: Tuple[BSuiteId, ...] = tuple(_sweep)
# Mapping from tag (e.g. 'memory') to experiment `bsuite_id`s with that tag.
TAGS: Mapping[Tag, Tuple[BSuiteId, ...]] = types.MappingProxyType(
    {k: tuple(v) for k, v in _tags.items()})
# Tuple containing a representative subset bsuite_ids used for agent tests.
TESTING: Tuple[BSuiteId, ...] = tuple(_testing)
# Mapping from bsuite_id to bsuite_num_episodes = how many episodes to run.
EPISODES: Mapping[BSuiteId, int] = types.MappingProxyType(_episodes)
