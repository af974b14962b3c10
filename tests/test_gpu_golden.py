"""GPU: the HIP kernels, through the C ABI, against outputs of the unmodified reference
(tests/golden/*.npz).  Integer / grid families: bit-exact (f32 reward == np.float32(reference f64
reward), observation bitwise).  Physics families: per-step teacher-forced, |a-b| <= 1e-6*max(1,|b|)
(BASELINE.json north_star tolerance)."""
import numpy as np
import pytest
import torch

from tests import engine_util as eu
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _force_physics_state(env, fam, phys_prev, idx):
  r = eu.raw(env)
  if fam == 'mountain_car':
    st = np.stack([phys_prev[idx, 0], phys_prev[idx, 1]]).astype(np.float32)
  else:
    st = phys_prev[idx, :4].T.astype(np.float32)
  r._state['state'].copy_(torch.from_numpy(np.ascontiguousarray(st)).to(r.device))


@pytest.mark.parametrize('name', gu.replay_case_names())
def test_engine_matches_reference(name):
  meta, g = gu.load_case(name)
  fam = meta['family']
  phys = fam in gu.PHYSICS
  wrap = tuple(meta['wrap']) if meta['wrap'] else None
  T = g['actions'].shape[0]
  kwargs = dict(meta['kwargs'])
  if fam == 'mnist':
    kwargs['images'], kwargs['labels'] = gu.mnist_dataset()
  for (i0, lane0, n) in gu.contiguous_runs(g['lanes']):
    idx = slice(i0, i0 + n)
    env = eu.make_env(fam, kwargs, batch=n, lane_offset=lane0, seed=meta['seed'], wrap=wrap)
    eu.raw(env)._step_index = meta['step0']
    logged = None
    if meta.get('log'):
      from bsuite_amd.utils import wrappers
      env = logged = wrappers.Logging(env, None, log_by_step=meta['log'] == 'by_step',
                                      log_every=meta['log'] == 'every', max_rows=g['log_rows'].shape[1] + 3)
    for t in range(T):
      if phys and t > 0:
        _force_physics_state(env, fam, g['phys'][t - 1], idx)
      if t in meta['reset_at']:
        ts = env.reset()
      else:
        ts = env.step(torch.from_numpy(g['actions'][t, idx]).to('cuda'))
      st, r, d, o = eu.to_np(ts)
      gst, gr, gd, go = g['step_type'][t, idx], g['reward'][t, idx], g['discount'][t, idx], g['obs'][t, idx]
      np.testing.assert_array_equal(st, gst, err_msg=f'{name} step_type t={t}')
      first = gst == 0
      assert (r[first] == 0).all() and (d[first] == 1).all()
      np.testing.assert_array_equal(d[~first], gd[~first].astype(np.float32))
      if phys:      # |a-b| <= 1e-6*max(1,|b|), the north_star bound (not rtol+atol = 2e-6)
        eu.assert_within_tol(o, go, err_msg=f'{name} obs t={t}')
        eu.assert_within_tol(r[~first], gr[~first], err_msg=f'{name} reward t={t}')
      else:
        np.testing.assert_array_equal(eu.f32_bits(r[~first]), eu.f32_bits(gr[~first].astype(np.float32)),
                                      err_msg=f'{name} reward t={t}')
        np.testing.assert_array_equal(eu.f32_bits(o), eu.f32_bits(go), err_msg=f'{name} obs t={t}')
      info = env.bsuite_info()
      for j, k in enumerate(meta['info_keys']):
        got = info[k].cpu().numpy()
        if phys:
          np.testing.assert_allclose(got, g['info'][t, idx, j], rtol=1e-9, atol=1e-9, err_msg=f'{k} t={t}')
        else:
          np.testing.assert_array_equal(got, g['info'][t, idx, j], err_msg=f'{name} {k} t={t}')
    if logged is not None:   # rows the unmodified reference Logging wrapper wrote, per lane
      assert list(eu.raw(env).logging_columns()[:5]) == meta['log_columns'][:5]
      cols = [meta['log_columns'].index(c) for c in eu.raw(env).logging_columns() if not c.startswith('_')]
      keep = [j for j, c in enumerate(eu.raw(env).logging_columns()) if not c.startswith('_')]
      n_rows = logged.num_rows().cpu().numpy()
      np.testing.assert_array_equal(n_rows, g['log_n_rows'][idx], err_msg=f'{name} n_rows')
      rows = logged._lg['rows'].cpu().numpy()
      for l in range(n):
        want = g['log_rows'][i0 + l, :n_rows[l]][:, cols]
        got = rows[l, :n_rows[l]][:, keep]
        if phys:
          np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9, err_msg=f'{name} lane {l}')
        else:
          np.testing.assert_array_equal(got, want, err_msg=f'{name} lane {l}')
