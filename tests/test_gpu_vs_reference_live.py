"""GPU: the HIP engine against the UNMODIFIED reference, LIVE on the GPU box.

The reference package is imported from the byte-code tree `__graft_entry__.build()` stages under
oracle/_ref (oracle/stage_reference.py; compiled from /root/reference, git-ignored, never imported by the
product) — or from /root/reference itself where that exists.  For random families / constructor
arguments / wrappers (RewardNoise, RewardScale, both stacking orders: utils/wrappers.py:275-283,338-346) /
explicit resets / scripted and random action streams, the reference environments
(bsuite/environments/*.py, experiments/cartpole_swingup/cartpole_swingup.py) run lane by lane with their
RandomState swapped for a replay of the engine's draw stream (oracle/replay.py), and the engine must
reproduce every TimeStep and every bsuite_info() call by call: bit-exact for the integer / grid
families, teacher-forced 1e-6*max(1,|b|) for the physics families.  A second test goes through
`load_from_id` on both sides (registry + sweep settings + experiment loaders of the reference itself).
Unlike tests/golden/*.npz (recorded in the build container) nothing here is a committed fixture: the
reference computes its side in this very process."""
import os

import numpy as np
import pytest

from oracle import replay
from tests import engine_util as eu

pytestmark = pytest.mark.gpu


# BSX_LIVE_CASES=<n per chunk> scales the run (tools/profiles.sh logs a 2000-case run under profiles/rNN/)
N_CHUNKS, CASES_PER_CHUNK = 8, int(os.environ.get('BSX_LIVE_CASES', '30'))


@pytest.fixture(scope='module')
def ref():
  if not replay.reference_available():
    pytest.skip('no reference on this box: neither /root/reference nor the staged oracle/_ref '
                '(python -m oracle.stage_reference in the build container)')
  bs = replay.import_reference()
  from oracle import make_golden as mg
  from bsuite_amd.utils import datasets as _ds          # only the idx *writer* (wire format)
  imgs, labs = mg.synthetic_mnist()                      # == tests/golden/mnist_synthetic_dataset.npz
  _ds.write_idx_files(mg.MNIST_DIR, imgs, labs)          # the reference's hard-wired /tmp/mnist (utils/datasets.py:42)
  return bs, mg


def _random_case(rng):
  fam = str(rng.choice(['deep_sea', 'catch', 'bandit', 'memory_chain', 'umbrella_chain', 'discounting_chain',
                        'cartpole', 'cartpole_swingup', 'mountain_car', 'mnist']))
  pol = ['random']
  if fam == 'deep_sea':
    kw = dict(size=int(rng.integers(1, 34)), deterministic=bool(rng.integers(2)), mapping_seed=int(rng.integers(100)),
              unscaled_move_cost=float(rng.choice([0.01, 0.05, 0.0])), randomize_actions=bool(rng.integers(4) > 0))
    pol = ['optimal', 'anti', 'random']
  elif fam == 'catch':
    kw = dict(rows=int(rng.integers(2, 16)), columns=int(rng.integers(1, 10)))
    pol = ['optimal', 'left', 'random']
  elif fam == 'bandit':
    kw = dict(mapping_seed=int(rng.integers(50)), num_actions=int(rng.integers(1, 20)))
  elif fam == 'memory_chain':
    kw = dict(memory_length=int(rng.integers(1, 12)), num_bits=int(rng.integers(1, 45)))
    pol = ['optimal', 'random']
  elif fam == 'umbrella_chain':
    kw = dict(chain_length=int(rng.integers(1, 10)), n_distractor=int(rng.integers(0, 110)))
    pol = ['optimal', 'random']
  elif fam == 'discounting_chain':
    kw = dict(mapping_seed=int(rng.integers(20)))
  elif fam == 'cartpole':
    kw = dict(max_time=float(rng.choice([10., 0.2])))
    pol = ['optimal', 'random']
  elif fam == 'cartpole_swingup':
    kw = dict(height_threshold=float(rng.random()), x_reward_threshold=float(rng.random()),
              init_range=float(rng.choice([0.05, 3.0])))
    pol = ['optimal', 'random']
  elif fam == 'mountain_car':
    kw = dict(max_steps=int(rng.integers(2, 60)))
    pol = ['optimal', 'random']
  else:
    kw = dict(fraction=float(rng.choice([1.0, 0.25])))
    pol = ['optimal', 'random']
  r = rng.random()
  wrap = None
  if r < 0.2:
    wrap = ('noise', float(rng.choice([0.1, 1.0, 10.0])))
  elif r < 0.4:
    wrap = ('scale', float(rng.choice([0.001, 30.0])))
  elif r < 0.5:
    wrap = ('scale_noise', float(rng.choice([0.03, 30.0])), float(rng.choice([0.3, 3.0])))
  elif r < 0.6:
    wrap = ('noise_scale', float(rng.choice([0.3, 3.0])), float(rng.choice([0.03, 30.0])))
  L = int(rng.integers(3, 10))
  lane0 = int(rng.choice([0, 5, (1 << 32) - 4, (1 << 40) + 3]))
  T = int(rng.integers(20, 70)) if fam not in ('discounting_chain',) else 110
  reset_at = tuple(int(x) for x in rng.choice(np.arange(2, T), size=int(rng.integers(0, 3)), replace=False))
  policies = [str(rng.choice(pol)) for _ in range(L)]
  return dict(family=fam, kwargs=kw, wrap=wrap, lanes=list(range(lane0, lane0 + L)), T=T, seed=int(rng.integers(1 << 40)),
              step0=int(rng.choice([0, (1 << 32) - 3, mg_big_step()])), reset_at=reset_at, policies=policies)


def mg_big_step():
  return (1 << 34) + 77


@pytest.mark.parametrize('chunk', range(N_CHUNKS))
def test_engine_matches_the_live_reference(ref, chunk):
  bs, mg = ref
  rng = np.random.default_rng(9000 + chunk)
  for j in range(CASES_PER_CHUNK):
    c = _random_case(rng)
    name = f"live{chunk}_{j}_{c['family']}"
    meta, g = mg.run_case(bs, name, c['family'], c['kwargs'], c['lanes'], c['T'], seed=c['seed'], step0=c['step0'],
                          wrap=c['wrap'], policies=c['policies'], reset_at=c['reset_at'], case_seed=chunk * 100 + j,
                          write=False)
    eu.check_against_case(f'{name} {c}', meta, g)


@pytest.mark.parametrize('chunk', range(4))
def test_load_from_id_both_sides(ref, chunk):
  """Random bsuite_ids: reference `bsuite.load_from_id(id)` (bsuite.py:101-108 → experiment loader → environment
  [+ wrapper]) vs `bsuite_amd.load_from_id(id, batch=L)`."""
  bs, mg = ref
  from bsuite_amd import sweep
  from tests.test_gpu_all_ids import oracle_config
  rng = np.random.default_rng(7000 + chunk)
  ids = [str(b) for b in rng.choice(np.array(sweep.SWEEP), size=12, replace=False)]
  for j, bid in enumerate(ids):
    fam, kw, _, fixed = oracle_config(bid)
    seed = int(rng.integers(1 << 31)) if fixed is None else fixed
    T = {'cartpole_swingup': 40, 'mountain_car': 40, 'discounting_chain': 105}.get(fam, min(70, 12 + 2 * kw.get('size', 10)))
    lanes = list(range(17, 17 + 4))
    meta, g = mg.run_case(bs, bid, fam, kw, lanes, T, seed=seed, case_seed=chunk * 50 + j, bsuite_id=bid, write=False)
    meta['seed_is_ours'] = fixed is None
    eu.check_against_case(bid, meta, g)


def test_the_reference_here_is_the_unmodified_one():
  """The staged tree's manifest lists the sha256 of every source it was compiled from; where the sources
  exist (build container) they must still hash to it."""
  from oracle import stage_reference as sr
  import json
  if not sr.staged():
    pytest.skip('running from /root/reference directly')
  with open(sr.MANIFEST) as f:
    m = json.load(f)
  assert 'environments/deep_sea.py' in m['sources'] and 'utils/wrappers.py' in m['sources']
  for s in m['sources']:
    assert os.path.exists(os.path.join(sr.STAGE_DIR, sr.PACKAGE, s[:-3] + '.pyc')), s
  if sr.reference_present():
    top = os.path.join(sr.REFERENCE_ROOT, sr.PACKAGE)
    assert {s: sr._sha(os.path.join(top, s)) for s in m['sources']} == m['sources']
