"""GPU: the fused Logging bookkeeping at batch scale against the oracle restatement, and the scalar
`load_and_record_to_csv` drop-in writing bsuite's CSV wire format."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

import bsuite_amd
from bsuite_amd.utils import wrappers
from oracle import coracle
from oracle import logging_oracle
from tests import engine_util as eu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('family,kwargs,wrap,by_step', [
    ('bandit', dict(mapping_seed=1), None, False),
    ('catch', dict(), ('noise', 0.3), False),
    ('deep_sea', dict(size=5, deterministic=False, mapping_seed=1), None, True),
    ('memory_chain', dict(memory_length=2, num_bits=3), ('scale', 0.03), False),
    ('discounting_chain', dict(mapping_seed=1), None, True)])
def test_batched_logging_matches_oracle(family, kwargs, wrap, by_step):
  B, T, seed = 2051, 150, 9
  env = eu.make_env(family, kwargs, batch=B, lane_offset=3, seed=seed, wrap=wrap)
  log = wrappers.Logging(env, None, log_by_step=by_step)
  orc = coracle.OracleEnv(family, kwargs, np.arange(3, 3 + B, dtype=np.uint64), seed=seed, wrap=wrap)
  trk = logging_oracle.TrackOracle(B, list(orc.bsuite_info()), log_by_step=by_step)
  rng = np.random.default_rng(0)
  for t in range(T):
    a = rng.integers(0, orc.num_actions, size=B).astype(np.int32)
    log.step(torch.from_numpy(a).cuda())
    st, r, _, _ = orc.call(a, t)
    trk.track(st, r, orc.bsuite_info())
  c = log.counters()
  np.testing.assert_array_equal(c['steps'].cpu().numpy(), trk.steps)
  np.testing.assert_array_equal(c['episode'].cpu().numpy(), trk.episode)
  np.testing.assert_array_equal(c['total_return'].cpu().numpy(), trk.total_return)
  np.testing.assert_array_equal(c['episode_return'].cpu().numpy(), trk.episode_return)
  np.testing.assert_array_equal(c['episode_len'].cpu().numpy(), trk.episode_len)
  n_rows = log.num_rows().cpu().numpy()
  np.testing.assert_array_equal(n_rows, [len(r) for r in trk.rows])
  rows = log._lg['rows'].cpu().numpy()
  cols = list(eu.raw(env).logging_columns())
  order = [0, 1, 2, 3, 4] + [5 + cols[5:].index(k) for k in trk.info_keys]
  for l in (0, 1, 77, B - 1):
    np.testing.assert_array_equal(rows[l, :n_rows[l]][:, order], np.array(trk.rows[l]).reshape(n_rows[l], -1))


def test_scalar_load_and_record_to_csv(tmp_path):
  env = bsuite_amd.load_and_record_to_csv('catch/0', str(tmp_path), seed=3)
  assert env.bsuite_num_episodes == 10000
  ts = env.reset()
  n_ep = 0
  while n_ep < 27:
    ts = env.step(1)
    n_ep += ts.last()
  path = os.path.join(str(tmp_path), 'bsuite_id_-_catch-0.csv')
  df = pd.read_csv(path)
  assert list(df.columns) == ['steps', 'episode', 'total_return', 'episode_len', 'episode_return', 'total_regret']
  assert df['episode'].tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 17, 20, 25]
  assert (df['episode_len'] == 9).all() and (df['steps'] == 9 * df['episode']).all()
  assert (df['total_regret'] == (df['episode'] - df['total_return'])).all()   # regret 2 per miss, return +-1
  with pytest.raises(ValueError):
    bsuite_amd.load_and_record('catch/0', str(tmp_path), logging_mode='sqlite')
  with pytest.raises(ValueError):
    bsuite_amd.load_and_record_to_csv('catch/0', str(tmp_path))              # file exists, overwrite=False


class _Collect:
  def __init__(self):
    self.rows = []

  def write(self, data):
    self.rows.append(dict(data))


def test_scalar_log_every_never_fills_the_row_buffer():
  """with log_every=True the reference logs every episode; the scalar view hands each row to
  the logger right after the step and rewinds its (8-row) buffer, so a run far longer than any buffer
  keeps logging (it used to stop after 4096 rows)."""
  from bsuite_amd.environments import bandit
  sink = _Collect()
  env = wrappers.Logging(bandit.SimpleBandit(3, seed=1), sink, log_every=True)
  n = 5000
  for _ in range(n):
    env.reset()
    env.step(2)
  assert len(sink.rows) == n and [r['episode'] for r in sink.rows[:3]] == [1, 2, 3] and sink.rows[-1]['episode'] == n
  assert sink.rows[-1]['steps'] == n


def test_by_step_logging_rows_fit_and_overflow_is_loud():
  """log_by_step logs a log-point LAST and the FIRST after it (same step count, wrappers.py:96-102): the
  default buffer holds both; an undersized buffer raises instead of silently dropping rows."""
  B, T = 64, 400
  env = eu.make_env('bandit', dict(mapping_seed=1), batch=B, lane_offset=0, seed=2)
  log = wrappers.Logging(env, None, log_by_step=True)
  orc = coracle.OracleEnv('bandit', dict(mapping_seed=1), np.arange(B, dtype=np.uint64), seed=2)
  trk = logging_oracle.TrackOracle(B, list(orc.bsuite_info()), log_by_step=True)
  a = np.zeros(B, np.int32)
  for t in range(T):
    log.step(torch.from_numpy(a).cuda())
    st, r, _, _ = orc.call(a, t)
    trk.track(st, r, orc.bsuite_info())
  assert not bool(log.overflowed().any())
  rows = log.all_rows()
  assert [len(r) for r in rows] == [len(r) for r in trk.rows] and len(rows[0]) > 40     # LAST + FIRST at each point
  small = wrappers.Logging(eu.make_env('bandit', dict(mapping_seed=1), batch=4, lane_offset=0, seed=2), None,
                           log_every=True, max_rows=5)
  for _ in range(40):
    small.step(torch.zeros(4, dtype=torch.int32, device='cuda'))
  assert bool(small.overflowed().all())
  with pytest.raises(RuntimeError):
    small.rows(0)
  with pytest.raises(RuntimeError):
    small.all_rows()
  assert set(small.counters()) >= {'steps', 'episode'}          # live tensors: no host read, no raise
  with pytest.raises(RuntimeError):
    small.counters(check=True)
  with pytest.raises(RuntimeError):
    small.flush()


def test_state_dict_carries_logging_and_wrapper_state():
  """Resuming under Logging + RewardNoise: counters, rows and the fused wrapper configuration travel."""
  B = 300
  mk = lambda: wrappers.Logging(eu.make_env('catch', {}, batch=B, lane_offset=0, seed=4, wrap=('noise', 0.5)), None)
  a, b = mk(), mk()
  g = torch.Generator(device='cuda'); g.manual_seed(1)
  acts = torch.randint(3, (90, B), generator=g, device='cuda', dtype=torch.int32)
  for t in range(40):
    a.step(acts[t])
  b.raw_env.load_state_dict(a.raw_env.state_dict())
  for t in range(40, 90):
    ta, tb = a.step(acts[t]), b.step(acts[t])
    assert torch.equal(ta.reward, tb.reward) and torch.equal(ta.step_type, tb.step_type)
  for k, v in a.counters().items():
    assert torch.equal(v, b.counters()[k]), k
  assert torch.equal(a.num_rows(), b.num_rows()) and a.all_rows() == b.all_rows()
  plain = eu.make_env('catch', {}, batch=B, lane_offset=0, seed=4)
  with pytest.raises(ValueError):
    plain.load_state_dict(a.raw_env.state_dict())            # taken with Logging, loaded without


def test_reward_wrapper_around_logging_is_refused():
  """the fused Logging bookkeeping sees the reward the kernel's epilogue returns — it is always the
  OUTERMOST wrapper.  RewardNoise(Logging(env)) would log raw rewards in the reference (utils/wrappers.py:74-77) and
  perturbed ones here: refused instead of silently different."""
  env = eu.make_env('catch', {}, batch=64, lane_offset=0, seed=1)
  logged = wrappers.Logging(env, None)
  with pytest.raises(NotImplementedError):
    wrappers.RewardNoise(logged, noise_scale=0.5, seed=1)
  with pytest.raises(NotImplementedError):
    wrappers.RewardScale(logged, reward_scale=2.0, seed=1)
  wrappers.Logging(wrappers.RewardNoise(eu.make_env('catch', {}, batch=64, lane_offset=0, seed=1), noise_scale=0.5, seed=1), None)


@pytest.mark.parametrize('family,kwargs,na', [('catch', {}, 3), ('mountain_car', dict(max_steps=15), 3)])
def test_logging_wrapped_mid_run_takes_over_the_folded_info_columns(family, kwargs, na):
  """Families that keep part of a bsuite_info column outside it while no Logging wrapper runs — catch counts its
  misses in spare bits of the packed state and folds them once per 127, mountain_car / cartpole fold at episode ends —
  hand the column over exactly when the wrapper arrives mid-run: bsuite_info() is the oracle's before, at, and after
  the hand-over, and the rows logged afterwards carry the reference's values."""
  B, seed = 700, 12
  env = eu.make_env(family, kwargs, batch=B, lane_offset=3, seed=seed)
  orc = coracle.OracleEnv(family, kwargs, np.arange(3, 3 + B, dtype=np.uint64), seed=seed)
  rng = np.random.default_rng(4)
  phys = family == 'mountain_car'

  def run(e, t0, n):
    for t in range(t0, t0 + n):
      a = rng.integers(0, na, size=B).astype(np.int32)
      if phys and t > 0:
        eu.teacher_force(eu.raw(env), orc, family)
      e.step(torch.from_numpy(a).cuda())
      orc.call(a, t)
      if t % 37 == 0 or t == t0 + n - 1:
        for k, v in orc.bsuite_info().items():
          np.testing.assert_array_equal(eu.raw(env).bsuite_info()[k].cpu().numpy(), v, err_msg=f'{family} {k} t={t}')

  run(env, 0, 420)                       # > 127 misses for many catch lanes? no: ~38 episodes; the fold path runs in the fuzz
  if family == 'catch':
    assert int(((eu.raw(env)._state['state'] >> 25) & 0x7F).max()) > 0       # something is pending outside the column
  logged = wrappers.Logging(env, None)
  if family == 'catch':
    assert int(((eu.raw(env)._state['state'] >> 25) & 0x7F).max()) == 0       # ... and has been handed over
  for k, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(eu.raw(env).bsuite_info()[k].cpu().numpy(), v, err_msg=f'{family} {k} at the hand-over')
  run(logged, 420, 200)
  rows = logged.all_rows()               # snapshot rows have been written since, on every lane
  assert len(rows) == B and all(len(r) > 0 for r in rows)


def test_catch_folds_total_regret_once_per_127_misses():
  """The fold itself: always-left on a 2-row board (an episode every 2 calls, the ball lands anywhere in 5 columns) runs
  several hundred misses per lane through the 7-bit counter; total_regret equals the oracle's at every check."""
  B, seed = 300, 5
  kwargs = dict(rows=2, columns=5)
  env = eu.make_env('catch', kwargs, batch=B, lane_offset=0, seed=seed)
  orc = coracle.OracleEnv('catch', kwargs, np.arange(B, dtype=np.uint64), seed=seed)
  a = np.zeros(B, np.int32)
  folded = False
  for t in range(900):
    env.step(torch.from_numpy(a).cuda())
    orc.call(a, t)
    if t % 50 == 49:
      np.testing.assert_array_equal(env.bsuite_info()['total_regret'].cpu().numpy(), orc.bsuite_info()['total_regret'])
      folded |= bool((eu.raw(env)._info[0] >= 254.0).any())
  assert folded                           # the column itself has received at least one batch of 127 misses
