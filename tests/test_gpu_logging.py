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
