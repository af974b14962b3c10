"""GPU: the single-launch catch step (catch_step1_kernel, csrc/catch.hip; ABI v12 BSX_CALL_STATE_TAGGED for catch) against
the C oracle — the path a lean Catch with columns <= 8 takes beyond the fused-tile range (more than 128 MiB of boards per
step: 2^20 lanes of the default board, tests/test_gpu_full_size.py::test_catch_full_batch): the threads of the
observation store stream recompute the transition of the lanes whose rows they write, bit 7 of the packed state word
(the parity of the next call's index) tells an already-advanced word from one that is not, and the ball column of a
lane's next episode is drawn ahead of time — on the calls whose index is a multiple of `rows` — and parked in spare bits
of the word (bits 6, 14, 15, 22).

The size gate makes the kernel unreachable at test-sized batches in the product library, so the tests below run in a
subprocess against the tuning build with BSX_CATCH_STEP1_MIN_MIB=0 (test_single_launch_catch_at_small_shapes starts it,
together with every other catch parity test of the suite); run directly they skip.  Covered: boards whose 16-byte chunks
do and do not straddle lanes, the widest eligible board, batches from 4 lanes up with a 64-bit lane offset,
explicit reset() calls in mid-episode (those take the two-launch step and un-park), a state_dict() moved between call
parities and lanes assembled at mixed episode phases (no parked draw: the threads draw for themselves until the next
parking call), the device-resident call counter under a HIP graph, an action ring, and the neighbours that must NOT take
the single launch (9 columns, an odd float count, a reward wrapper) — all bit-exact, bsuite_info() and the episode
counters included."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import coracle
from tests import engine_util as eu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INNER = os.environ.get('BSX_TEST_CATCH_STEP1') == '1'
inner = pytest.mark.skipif(not INNER, reason='runs in the subprocess of test_single_launch_catch_at_small_shapes (tuning build)')
PARK_VALID, TAG = 1 << 6, 1 << 7


@pytest.mark.timeout(1200)
@pytest.mark.skipif(INNER, reason='the outer test')
def test_single_launch_catch_at_small_shapes():
  from bsuite_amd import build as _build
  env = dict(os.environ, BSX_NATIVE_LIB=_build.build(tuning=True), BSX_CATCH_STEP1_MIN_MIB='0', BSX_FUSED_TILE_MAX_CELLS='0',
             BSX_TEST_CATCH_STEP1='1', PYTHONPATH=ROOT)
  p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                      'tests/test_gpu_catch_single_launch.py', 'tests/test_gpu_golden.py', 'tests/test_gpu_oracle_batch.py',
                      'tests/test_gpu_engine_features.py', 'tests/test_gpu_dm_env_conformance.py', 'tests/test_gpu_logging.py',
                      '-k', 'catch or engine or single_launch'],
                     cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1100)
  tail = p.stdout[-3000:]
  assert p.returncode == 0, tail
  assert ' passed' in tail and 'failed' not in tail, tail


def _check(ts, want, msg):
  st, r, d, o = want
  gst, gr, gd, go = eu.to_np(ts)
  np.testing.assert_array_equal(gst, st, err_msg=msg)
  live = st != 0
  np.testing.assert_array_equal(eu.f32_bits(gr[live]), eu.f32_bits(r[live].astype(np.float32)), err_msg='reward ' + msg)
  np.testing.assert_array_equal(gd[live], d[live].astype(np.float32), err_msg='discount ' + msg)
  assert (gr[~live] == 0).all() and (gd[~live] == 1).all(), msg
  np.testing.assert_array_equal(eu.f32_bits(go), eu.f32_bits(o), err_msg='observation ' + msg)


@inner
@pytest.mark.parametrize('rows,cols', [(10, 5), (7, 3), (5, 4), (9, 8), (64, 8), (2, 2), (3, 7)])
@pytest.mark.parametrize('batch,lane_offset', [(4, 0), (336, 5), (4100, (1 << 32) - 18)])   # multiples of 4: every board's float count is
def test_single_launch_step_bit_exact(rows, cols, batch, lane_offset):
  kw = dict(rows=rows, columns=cols)
  seed = 77
  env = eu.make_env('catch', kw, batch=batch, lane_offset=lane_offset, seed=seed)
  orc = coracle.OracleEnv('catch', kw, np.arange(lane_offset, lane_offset + batch, dtype=np.uint64), seed=seed)
  rng = np.random.default_rng(rows * 100 + cols + batch)
  T = 3 * rows + 7
  n_last = n_first = 0
  seen_parked = False
  for t in range(T):
    a = rng.integers(0, 3, size=batch).astype(np.int32)
    if t % 5 == 2:
      a[:] = t % 3
    force = t in (rows // 2 + 1, rows + 3)
    ts = env.reset() if force else env.step(torch.from_numpy(a).cuda())
    want = orc.call(a, t, force_reset=force)
    _check(ts, want, f'{rows}x{cols} B={batch} t={t}')
    n_last += int((want[0] == 2).sum()); n_first += int((want[0] == 0).sum())
    st = eu.raw(env)._state['state']
    if force:
      assert int((st & PARK_VALID).sum()) == 0                     # an explicit reset() un-parks (the two-launch step)
    elif t % rows == 0:
      assert bool(((st & PARK_VALID) != 0).all())                  # a parking call: every writer parked its lane's next column
      seen_parked = True
    assert bool((((st & TAG) != 0) == bool((t + 1) & 1)).all())    # every advance of every path writes the next parity
  assert seen_parked and eu.raw(env)._call_desc.flags == 1
  info = env.bsuite_info()
  for k, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(info[k].cpu().numpy(), v, err_msg=k)
  c = eu.raw(env).episode_counters().cpu().numpy()
  assert (int(c[0]), int(c[1])) == (n_last, n_first)


@inner
def test_state_dict_moves_between_call_parities_and_phases():
  kw = dict()
  B, seed = 2000, 3
  a_env = eu.make_env('catch', kw, batch=B, lane_offset=0, seed=seed)
  rng = np.random.default_rng(1)
  acts = [torch.from_numpy(rng.integers(0, 3, size=B).astype(np.int32)).cuda() for _ in range(80)]
  for t in range(7):                                               # 7 calls: the next index is odd
    a_env.step(acts[t])
  sd = a_env.state_dict()
  assert int((sd['state'] & ((3 << 6) | (3 << 14) | (3 << 22))).sum()) == 0          # a dict carries neither tag nor parked draw
  b_env = eu.make_env('catch', kw, batch=B, lane_offset=0, seed=seed)
  for t in range(4):
    b_env.step(acts[40 + t])
  b_env.load_state_dict(sd)
  for t in range(7, 45):
    x, y = a_env.step(acts[t]), b_env.step(acts[t])
    for u, v in zip(eu.to_np(x), eu.to_np(y)):
      np.testing.assert_array_equal(u, v, err_msg=f't={t}')
  # mixed phases: lane i keeps the state it had after (i % 7) + 1 calls — what bench.stagger_phases does; the lanes then
  # reset on different calls, most of them before the next parking call: their threads draw for themselves
  c_env = eu.make_env('catch', kw, batch=B, lane_offset=0, seed=seed)
  final = None
  phase = torch.arange(B, device='cuda') % 7
  for k in range(7):
    c_env.step(acts[k])
    sdk = c_env.state_dict()
    if final is None:
      final = sdk
    else:
      for key, val in sdk.items():
        if torch.is_tensor(val) and val.dim() >= 1 and val.shape[-1] == B and key != '__counters':
          final[key] = torch.where(phase == k, val, final[key])
        else:
          final[key] = val
  c_env.load_state_dict(final)
  groups = {}
  for k in range(7):
    o2 = coracle.OracleEnv('catch', kw, np.arange(B, dtype=np.uint64), seed=seed)
    for t in range(k + 1):
      o2.call(acts[t].cpu().numpy(), t)
    groups[k] = o2
  for t in range(7, 50):
    got = eu.to_np(c_env.step(acts[t]))
    for k in range(7):
      want = groups[k].call(acts[t].cpu().numpy(), t)
      sel = (np.arange(B) % 7) == k
      np.testing.assert_array_equal(got[0][sel], want[0][sel], err_msg=f'step_type group {k} t={t}')
      np.testing.assert_array_equal(eu.f32_bits(got[3][sel]), eu.f32_bits(want[3][sel]), err_msg=f'obs group {k} t={t}')
      live = want[0][sel] != 0
      np.testing.assert_array_equal(got[1][sel][live], want[1][sel][live].astype(np.float32), err_msg=f'reward group {k} t={t}')


@inner
def test_single_launch_under_a_device_step_counter_hip_graph_and_action_ring():
  import bsuite_amd
  B = 3000
  env = bsuite_amd.load_from_id('catch/0', batch=B, seed=5, device_step_counter=True, num_buffers=2)
  ref = bsuite_amd.load_from_id('catch/0', batch=B, seed=5, num_buffers=2)
  g = torch.Generator(device='cuda'); g.manual_seed(2)
  acts = torch.randint(3, (8, B), generator=g, device='cuda', dtype=torch.int32)
  env.step(acts[0]); ref.step(acts[0])
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
      with env.step_counter_deferred():
        for t in range(1, 8):                                      # SEVEN steps per replay: parity and parking calls shift every replay
          out = env.step(acts[t])
  torch.cuda.current_stream().wait_stream(side)
  for rep in range(9):
    graph.replay()
    for t in range(1, 8):
      want = ref.step(acts[t])
    torch.cuda.synchronize()
    for u, v in zip(eu.to_np(out), eu.to_np(want)):
      np.testing.assert_array_equal(u, v, err_msg=f'replay {rep}')
  # an action ring walked on the device: row (call index mod R) of [R, B]
  R = 4
  env2 = eu.make_env('catch', dict(), batch=B, lane_offset=0, seed=9)
  orc = coracle.OracleEnv('catch', dict(), np.arange(B, dtype=np.uint64), seed=9)
  ring = torch.randint(3, (R, B), generator=g, device='cuda', dtype=torch.int32)
  ring_np = ring.cpu().numpy()
  raw = eu.raw(env2)
  raw._ensure_allocated()
  raw._call_desc.action_ring = R
  try:
    for t in range(35):
      b = raw._call(ring.data_ptr(), False)
      _check(raw._wrap_output(b), orc.call(ring_np[t % R], t), f'ring t={t}')
  finally:
    raw._call_desc.action_ring = 0
  assert bool(((raw._state['state'] & PARK_VALID) != 0).any())


@inner
@pytest.mark.parametrize('kw,wrap,batch', [(dict(rows=6, columns=9), None, 1500),              # 9 columns: the parked draw has 3 bits
                                           (dict(), None, 1501),                                # odd lanes x 50 floats: a ragged tail
                                           (dict(), ('noise', 0.5), 1500)])                     # not lean
def test_neighbours_of_the_single_launch_path(kw, wrap, batch):
  seed = 11
  env = eu.make_env('catch', kw, batch=batch, lane_offset=3, seed=seed, wrap=wrap)
  orc = coracle.OracleEnv('catch', kw, np.arange(3, 3 + batch, dtype=np.uint64), seed=seed, wrap=wrap)
  rng = np.random.default_rng(0)
  for t in range(45):
    a = rng.integers(0, 3, size=batch).astype(np.int32)
    _check(env.step(torch.from_numpy(a).cuda()), orc.call(a, t), f'{kw} t={t}')
    assert int((eu.raw(env)._state['state'] & PARK_VALID).sum()) == 0       # the two-launch step: nothing is ever parked
  info = env.bsuite_info()
  for k, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(info[k].cpu().numpy(), v, err_msg=k)
