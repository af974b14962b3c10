"""Helpers shared by the parity tests: load tests/golden/*.npz (reference outputs)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
PHYSICS = ('cartpole', 'cartpole_swingup', 'mountain_car')


def case_names():
  names = [os.path.basename(p)[:-4] for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))]
  return [n for n in names if n not in ('host_constants', 'mnist_synthetic_dataset', 'image_adapter', 'gym_adapter')]


def mnist_dataset():
  """The synthetic dataset the mnist fixtures were generated on (int8 view, as the reference parses)."""
  d = np.load(os.path.join(GOLDEN_DIR, 'mnist_synthetic_dataset.npz'))
  return d['images_u8'].view(np.int8), d['labels']


def replay_case_names():
  """Fixtures made with the engine's draw stream replayed into the reference (not the mt_* ones)."""
  return [n for n in case_names() if not n.startswith('mt_')]


def mt_case_names():
  """Fixtures made from the unmodified reference running on its own np.random.RandomState(seed)."""
  return [n for n in case_names() if n.startswith('mt_')]


def load_case(name):
  g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
  meta = json.loads(str(g['meta']))
  return meta, {k: g[k] for k in g.files if k != 'meta'}


def contiguous_runs(lanes):
  """Split a list of global lane ids into (start_index, lane_offset, count) contiguous runs."""
  runs, i = [], 0
  lanes = [int(x) for x in lanes]
  while i < len(lanes):
    j = i
    while j + 1 < len(lanes) and lanes[j + 1] == lanes[j] + 1:
      j += 1
    runs.append((i, lanes[i], j - i + 1))
    i = j + 1
  return runs


def image_adapter_cases():
  """(name, shape, obs [n,*obs_shape], image [n,*shape]) written by the reference's own to_image."""
  with open(os.path.join(GOLDEN_DIR, 'adapters.json')) as f:
    meta = json.load(f)
  g = np.load(os.path.join(GOLDEN_DIR, 'image_adapter.npz'))
  return [(k, tuple(m['shape']), g[k + '__obs'], g[k + '__image'])
          for k, m in sorted(meta.items()) if k + '__obs' in g.files]
