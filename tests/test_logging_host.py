"""CPU: host side of the batched Logging row: the log-point table is exactly the
reference predicate, the oracle restatement of `_track` reproduces the rows the unmodified reference
wrapper wrote (fixtures), and the CSV logger uses bsuite's wire format."""
import os

import numpy as np
import pandas as pd
import pytest

from bsuite_amd.logging import csv_logging
from bsuite_amd.utils import wrappers
from oracle import coracle
from oracle import logging_oracle
from tests import golden_util as gu


def test_log_points_are_the_reference_predicate():
  pts = wrappers.logarithmic_logging_points(10 ** 5)
  brute = [n for n in range(0, 10 ** 5 + 1) if wrappers._logarithmic_logging(n)]
  assert pts == brute
  assert pts[:16] == [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 17, 20, 25, 30]
  big = wrappers.logarithmic_logging_points(10 ** 9)
  assert big[-1] == 10 ** 9 and all(logging_oracle.logarithmic_logging(np.array(big)))


@pytest.mark.parametrize('name', [n for n in gu.case_names() if n.startswith('logging_')])
def test_track_oracle_matches_reference_rows(name):
  meta, g = gu.load_case(name)
  fam = meta['family']
  env = coracle.OracleEnv(fam, meta['kwargs'], g['lanes'], seed=meta['seed'],
                          wrap=tuple(meta['wrap']) if meta['wrap'] else None)
  trk = logging_oracle.TrackOracle(len(g['lanes']), meta['info_keys'], meta['log'] == 'by_step',
                                   meta['log'] == 'every')
  for t in range(g['actions'].shape[0]):
    st, r, _, _ = env.call(g['actions'][t], meta['step0'] + t, force_reset=t in meta['reset_at'])
    trk.track(st, r, env.bsuite_info())
  for l in range(len(g['lanes'])):
    n = int(g['log_n_rows'][l])
    assert len(trk.rows[l]) == n
    got, want = np.array(trk.rows[l]).reshape(n, -1), g['log_rows'][l, :n]
    if fam in gu.PHYSICS:
      np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
    else:
      np.testing.assert_array_equal(got, want)


def test_csv_logger_wire_format(tmp_path):
  lg = csv_logging.Logger('deep_sea/3', str(tmp_path))
  lg.write(dict(steps=10, episode=1, total_return=0.5, episode_len=10, episode_return=0.5, total_bad_episodes=0))
  lg.write(dict(steps=20, episode=2, total_return=1.5, episode_len=10, episode_return=1.0, total_bad_episodes=1))
  path = os.path.join(str(tmp_path), 'bsuite_id_-_deep_sea-3.csv')        # csv_logging.py:29-31,72-75
  assert os.path.exists(path)
  df = pd.read_csv(path)
  assert list(df.columns) == ['steps', 'episode', 'total_return', 'episode_len', 'episode_return', 'total_bad_episodes']
  assert df['episode'].tolist() == [1, 2]
  with pytest.raises(ValueError):
    csv_logging.Logger('deep_sea/3', str(tmp_path))                       # refuses to overwrite (:77-80)
  csv_logging.Logger('deep_sea/3', str(tmp_path), overwrite=True)


def test_reference_csv_loader_reads_our_files(tmp_path):
  """Where the reference tree exists (build container): the UNMODIFIED bsuite.logging.csv_load reads
  files written by our csv_logging.Logger and joins the sweep metadata onto them."""
  from oracle import replay
  if not replay.reference_available():
    pytest.skip('reference tree not on this box')
  replay.import_reference()
  from bsuite.logging import csv_load
  for bid, key in (('catch/0', 'total_regret'), ('deep_sea/10', 'total_bad_episodes')):
    lg = csv_logging.Logger(bid, str(tmp_path))
    for ep in (1, 2, 3):
      lg.write({'steps': 9 * ep, 'episode': ep, 'total_return': float(ep), 'episode_len': 9,
                'episode_return': 1.0, key: 0.0})
  df, sweep_vars = csv_load.load_bsuite(str(tmp_path))
  assert set(df.bsuite_id) == {'catch/0', 'deep_sea/10'}
  assert set(df.bsuite_env) == {'catch', 'deep_sea'}
  assert df[df.bsuite_id == 'deep_sea/10']['size'].iloc[0] == 30
