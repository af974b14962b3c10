"""GPU: end-to-end results parity at the logged-CSV level.

tests/golden/csv/*.csv were written by the reference's OWN stack run untouched in the build
container (oracle/make_golden.py: environment on its np.random.RandomState(seed), `Logging` wrapper,
CSV logger, `baselines.random` agent, `baselines.experiment.run` loop).  Here the same agent draws
drive `bsuite_amd.load_and_record_to_csv(..., rng='mt19937')` through the same loop, and the CSV
the engine's fused bookkeeping produces must contain the same rows, value for value."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import bsuite_amd
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
CSV_DIR = os.path.join(gu.GOLDEN_DIR, 'csv')
RUNS = json.load(open(os.path.join(CSV_DIR, 'runs.json')))


class RandomAgent:
  """bsuite/baselines/random/agent.py:26-37 restated: uniform actions from RandomState(seed)."""

  def __init__(self, num_actions, seed):
    self._num_actions, self._rng = num_actions, np.random.RandomState(seed)

  def select_action(self, timestep):
    del timestep
    return self._rng.randint(self._num_actions)


def run(agent, environment, num_episodes):
  """bsuite/baselines/experiment.py:43-57 restated."""
  for _ in range(num_episodes):
    timestep = environment.reset()
    while not timestep.last():
      action = agent.select_action(timestep)
      timestep = environment.step(action)


@pytest.mark.parametrize('cfg', RUNS, ids=[r['bsuite_id'] for r in RUNS])
def test_logged_csv_equals_the_reference_csv(cfg, tmp_path):
  bid = cfg['bsuite_id']
  kw = dict(rng='mt19937')
  if cfg['env_seed'] is not None:
    kw['seed'] = cfg['env_seed']
  env = bsuite_amd.load_and_record_to_csv(bid, str(tmp_path), **kw)
  agent = RandomAgent(env.action_spec().num_values, cfg['agent_seed'])
  run(agent, env, cfg['episodes'])
  fname = 'bsuite_id_-_' + bid.replace('/', '-') + '.csv'
  got = pd.read_csv(os.path.join(str(tmp_path), fname))
  want = pd.read_csv(os.path.join(CSV_DIR, fname))
  assert list(got.columns) == list(want.columns)
  assert len(got) == len(want) > 5
  pd.testing.assert_frame_equal(got, want, check_dtype=False, check_exact=True)
