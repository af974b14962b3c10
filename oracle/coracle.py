"""ORACLE (test infrastructure): ctypes front-end to oracle/oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker.  It owns plain numpy state in the reference's own terms (row, column, ...),
deliberately unlike the packed device layout.  Host-side constants that the reference builds with
np.random.RandomState (deep_sea action mapping deep_sea.py:77-85, bandit rewards bandit.py:43-47)
are built here with numpy too — this file has no dependency on bsuite_amd.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle.so')
_SRC = os.path.join(_HERE, 'oracle.c')

FIRST, MID, LAST = 0, 1, 2
WRAP = {None: 0, 'scale': 1, 'noise': 2,
        'scale_noise': 3,     # RewardNoise(RewardScale(env)): wrap = ('scale_noise', scale, sigma)
        'noise_scale': 4}     # RewardScale(RewardNoise(env)): wrap = ('noise_scale', sigma, scale)


def build(force=False):
  if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
    subprocess.check_call(['gcc', '-O2', '-std=gnu99', '-ffp-contract=off', '-fno-fast-math', '-shared',
                           '-fPIC', '-o', _SO, _SRC, '-lm'])
  return _SO


class _Call(ctypes.Structure):
  _fields_ = [('n_lanes', ctypes.c_int64), ('lane_ids', ctypes.c_void_p), ('seed', ctypes.c_uint64),
              ('step', ctypes.c_uint64), ('force_reset', ctypes.c_int32),
              ('wrap_kind', ctypes.c_int32), ('wrap_param', ctypes.c_double),
              ('wrap_seed', ctypes.c_uint64), ('action', ctypes.c_void_p),
              ('step_type', ctypes.c_void_p), ('reward', ctypes.c_void_p),
              ('discount', ctypes.c_void_p), ('obs', ctypes.c_void_p), ('wrap_param2', ctypes.c_double)]


class _CartpoleCfg(ctypes.Structure):
  _fields_ = [('swingup', ctypes.c_int32)] + [(n, ctypes.c_double) for n in (
      'height_threshold', 'x_threshold', 'theta_dot_threshold', 'x_reward_threshold', 'move_cost',
      'timescale', 'max_time', 'init_range', 'mass_cart', 'mass_pole', 'length', 'force_mag',
      'gravity')]


_lib = None


def lib():
  global _lib  # pylint: disable=global-statement
  if _lib is None:
    _lib = ctypes.CDLL(build())
  return _lib


def _p(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def stream_words(seed, lane, step, stream_id, n):
  out = np.zeros(n, np.uint32)
  lib().orc_stream_words(ctypes.c_uint64(seed), ctypes.c_uint64(lane), ctypes.c_uint64(step),
                         ctypes.c_uint32(stream_id), ctypes.c_int(n), _p(out))
  return out


def normals(k53):
  k53 = np.ascontiguousarray(k53, np.uint64)
  out = np.zeros(k53.shape[0], np.float64)
  lib().orc_normals(_p(k53), ctypes.c_int64(k53.shape[0]), _p(out))
  return out


def deep_sea_mapping(size, mapping_seed, randomize_actions=True):
  """deep_sea.py:76-85 with numpy on the host (values as the reference holds them)."""
  if randomize_actions:
    return np.random.RandomState(mapping_seed).binomial(1, 0.5, [size, size]).astype(np.float64)
  return np.ones([size, size], np.float64)


def bandit_rewards(mapping_seed, num_actions=11):
  """bandit.py:43-47."""
  rng = np.random.RandomState(mapping_seed)
  mask = rng.choice(range(num_actions), size=num_actions, replace=False)
  return np.linspace(0, 1, num_actions)[mask].astype(np.float64)


class OracleEnv:
  """A batch of independent reference environments restated in C.

  family/kwargs use the reference constructor's own argument names.
  """

  def __init__(self, family, kwargs, lane_ids, seed=42, wrap=None, wrap_seed=None):
    self.family, self.kw = family, dict(kwargs)
    self.lane_ids = np.ascontiguousarray(lane_ids, np.uint64)
    self.B = B = int(self.lane_ids.shape[0])
    self.seed = int(seed)
    self.wrap = wrap
    self.wrap_seed = self.seed if wrap_seed is None else int(wrap_seed)
    i32 = lambda fill=0: np.full(B, fill, np.int32)  # noqa: E731
    f64 = lambda: np.zeros(B, np.float64)  # noqa: E731
    self.reset_next = i32(1)                   # base.py:52 `_reset_next_step = True`
    kw = self.kw
    L = lib()
    if family == 'deep_sea':
      N = kw['size']
      self.obs_shape = (N, N)
      self.num_actions = 2
      self.mapping = deep_sea_mapping(N, kw.get('mapping_seed'), kw.get('randomize_actions', True))
      self.s = dict(row=i32(), col=i32(), bad=i32())
      self.info = dict(total_bad_episodes=f64(), denoised_return=f64())
      self._fn = lambda c: L.orc_deep_sea(
          c, ctypes.c_int(N), ctypes.c_int(int(kw.get('deterministic', True))),
          ctypes.c_double(kw.get('unscaled_move_cost', 0.01)), _p(self.mapping), _p(self.s['row']),
          _p(self.s['col']), _p(self.s['bad']), _p(self.reset_next),
          _p(self.info['total_bad_episodes']), _p(self.info['denoised_return']))
    elif family == 'catch':
      rows, cols = kw.get('rows', 10), kw.get('columns', 5)
      self.obs_shape = (rows, cols)
      self.num_actions = 3
      self.s = dict(ball_x=i32(), ball_y=i32(), paddle_x=i32())
      self.info = dict(total_regret=f64())
      self._fn = lambda c: L.orc_catch(
          c, ctypes.c_int(rows), ctypes.c_int(cols), _p(self.s['ball_x']), _p(self.s['ball_y']),
          _p(self.s['paddle_x']), _p(self.reset_next), _p(self.info['total_regret']))
    elif family == 'bandit':
      na = kw.get('num_actions', 11)
      self.obs_shape = (1, 1)
      self.num_actions = na
      self.rewards = bandit_rewards(kw.get('mapping_seed'), na)
      self.s = {}
      self.info = dict(total_regret=f64())
      self._fn = lambda c: L.orc_bandit(c, ctypes.c_int(na), _p(self.rewards), _p(self.reset_next),
                                        _p(self.info['total_regret']))
    elif family == 'memory_chain':
      ml, nb = kw['memory_length'], kw.get('num_bits', 1)
      self.obs_shape = (1, nb + 2)
      self.num_actions = 2
      self.s = dict(timestep=i32(), query=i32(), context=np.zeros((B, nb), np.int32))
      self.info = dict(total_perfect=f64(), total_regret=f64())
      self._fn = lambda c: L.orc_memory_chain(
          c, ctypes.c_int(ml), ctypes.c_int(nb), _p(self.s['timestep']), _p(self.s['query']),
          _p(self.s['context']), _p(self.reset_next), _p(self.info['total_perfect']),
          _p(self.info['total_regret']))
    elif family == 'umbrella_chain':
      cl, nd = kw['chain_length'], kw.get('n_distractor', 0)
      self.obs_shape = (1, 3 + nd)
      self.num_actions = 2
      self.s = dict(timestep=i32(), need=i32(), has=i32())
      self.info = dict(total_regret=f64())
      self._fn = lambda c: L.orc_umbrella_chain(
          c, ctypes.c_int(cl), ctypes.c_int(nd), _p(self.s['timestep']), _p(self.s['need']),
          _p(self.s['has']), _p(self.reset_next), _p(self.info['total_regret']))
    elif family == 'discounting_chain':
      ms = kw.get('mapping_seed')
      assert ms is not None, 'mapping_seed=None draws from the global numpy RNG in the reference'
      self.obs_shape = (1, 2)
      self.num_actions = 5
      self.s = dict(timestep=i32(), context=i32(-1))
      self.info = {}
      self._fn = lambda c: L.orc_discounting_chain(
          c, ctypes.c_int(ms), _p(self.s['timestep']), _p(self.s['context']), _p(self.reset_next))
    elif family in ('cartpole', 'cartpole_swingup'):
      sw = family == 'cartpole_swingup'
      self.cfg = _CartpoleCfg(
          swingup=int(sw), height_threshold=kw.get('height_threshold', 0.5 if sw else 0.8),
          x_threshold=kw.get('x_threshold', 3.), theta_dot_threshold=kw.get('theta_dot_threshold', 1.),
          x_reward_threshold=kw.get('x_reward_threshold', 1.), move_cost=kw.get('move_cost', 0.1),
          timescale=kw.get('timescale', 0.01), max_time=kw.get('max_time', 10.),
          init_range=kw.get('init_range', 0.05), mass_cart=1., mass_pole=0.1, length=0.5,
          force_mag=10., gravity=9.8)
      self.obs_shape = (1, 8 if sw else 6)
      self.num_actions = 3
      self.s = dict(state=np.zeros((B, 5), np.float64))
      self.info = dict(raw_return=f64(), best_episode=f64())
      self._episode_return = f64()
      self._upright = f64()
      if sw:
        self.info['total_upright'] = self._upright
      self._fn = lambda c: L.orc_cartpole(
          c, ctypes.byref(self.cfg), _p(self.s['state']), _p(self.reset_next),
          _p(self.info['raw_return']), _p(self.info['best_episode']), _p(self._episode_return),
          _p(self._upright))
    elif family == 'mountain_car':
      ms = kw.get('max_steps', 1000)
      self.obs_shape = (1, 3)
      self.num_actions = 3
      self.s = dict(position=f64(), velocity=f64(), timestep=i32())
      self.info = dict(raw_return=f64())
      self._fn = lambda c: L.orc_mountain_car(
          c, ctypes.c_int(ms), _p(self.s['position']), _p(self.s['velocity']),
          _p(self.s['timestep']), _p(self.reset_next), _p(self.info['raw_return']))
    elif family == 'mnist':
      self.images = np.ascontiguousarray(kw['images'], np.int8)
      self.labels = np.ascontiguousarray(kw['labels'], np.uint8)
      nd = int(kw.get('fraction', 1.) * len(self.labels))
      self.obs_shape = tuple(self.images.shape[1:])
      npix = int(np.prod(self.obs_shape))
      self.num_actions = 10
      self.s = dict(correct_label=i32())
      self.info = dict(total_regret=f64())
      self._fn = lambda c: L.orc_mnist(
          c, ctypes.c_int(nd), ctypes.c_int(npix), _p(self.images), _p(self.labels),
          _p(self.s['correct_label']), _p(self.reset_next), _p(self.info['total_regret']))
    else:
      raise KeyError(family)
    self.obs_numel = int(np.prod(self.obs_shape))
    self.step_type = np.zeros(B, np.int8)
    self.reward = np.zeros(B, np.float64)
    self.discount = np.zeros(B, np.float64)
    self.obs = np.zeros((B,) + self.obs_shape, np.float32)

  def call(self, actions, step, force_reset=False):
    """One reset()/step() call on every lane.  Returns views of the oracle's output buffers."""
    actions = np.ascontiguousarray(actions, np.int32)
    assert actions.shape == (self.B,)
    kind = WRAP[self.wrap[0]] if self.wrap else 0
    param = float(self.wrap[1]) if self.wrap else 0.0
    param2 = float(self.wrap[2]) if self.wrap and len(self.wrap) > 2 else 0.0
    c = _Call(self.B, _p(self.lane_ids), self.seed, int(step), int(bool(force_reset)), kind, param,
              self.wrap_seed, _p(actions), _p(self.step_type), _p(self.reward), _p(self.discount),
              _p(self.obs), param2)
    self._fn(ctypes.byref(c))
    return self.step_type, self.reward, self.discount, self.obs

  def bsuite_info(self):
    return self.info
