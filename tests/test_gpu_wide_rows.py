"""GPU: the wide observation rows of memory_chain / umbrella_chain on the ROW path (ABI v12, bsx_call_t.row_scratch;
csrc/row_stream.h, csrc/bsx_rows.h): the lane's thread leaves its row packed in a scratch column and a barrier-free
store stream decodes it (round 5, second form: every wave of the advance builds its 64 lanes' FLAT bit planes in
wave-private LDS, csrc/bsx_rows.h) — against the C oracle, bit-exact, at every row shape class (one / two / several bit words,
rows that are and are not multiples of 16 bytes, the widest rows the ABI takes), ragged / one-lane / several-thousand-lane
batches with a 64-bit lane offset, explicit reset() calls, both reward wrappers, the Logging wrapper, the MT19937-exact
mode and a captured HIP graph; equal, call for call, to the one-launch LDS path the same environment takes without a
scratch; and inside the whole-sweep group (both schedules), where the group's store stream writes the rows.
bsuite/environments/memory_chain.py:60-97, umbrella_chain.py:60-92."""
import ctypes

import numpy as np
import pytest
import torch

import bsuite_amd
from bsuite_amd import _native
from bsuite_amd import sweep_batch as sb
from bsuite_amd.environments import base
from bsuite_amd.utils import wrappers
from oracle import coracle, logging_oracle
from tests import engine_util as eu

pytestmark = pytest.mark.gpu

CASES = [
    ('umbrella_chain', dict(chain_length=4, n_distractor=20), None),          # umbrella_length's row: 23 floats
    ('umbrella_chain', dict(chain_length=3, n_distractor=6), None),           # 9 floats: the shortest row on the path
    ('umbrella_chain', dict(chain_length=3, n_distractor=29), None),          # 32 floats: 16-byte multiples, no straddle
    ('umbrella_chain', dict(chain_length=3, n_distractor=32), None),          # exactly one bit word
    ('umbrella_chain', dict(chain_length=3, n_distractor=33), None),          # two
    ('umbrella_chain', dict(chain_length=20, n_distractor=100), None),        # umbrella_distract/22
    ('umbrella_chain', dict(chain_length=2, n_distractor=253), None),         # the widest row: 256 floats, 8 bit words
    ('umbrella_chain', dict(chain_length=6, n_distractor=64), ('noise', 0.1)),
    ('umbrella_chain', dict(chain_length=5, n_distractor=40), ('scale', 30.0)),
    ('memory_chain', dict(memory_length=3, num_bits=7), None),                # 9 floats
    ('memory_chain', dict(memory_length=2, num_bits=30), None),               # 32 floats
    ('memory_chain', dict(memory_length=2, num_bits=32), None),
    ('memory_chain', dict(memory_length=2, num_bits=33), None),
    ('memory_chain', dict(memory_length=2, num_bits=40), None),               # memory_size/16
    ('memory_chain', dict(memory_length=1, num_bits=62), None),               # the widest context
    ('memory_chain', dict(memory_length=9, num_bits=17), ('noise', 0.5)),
    ('memory_chain', dict(memory_length=4, num_bits=11), ('noise_scale', 0.5, 3.0)),
]


@pytest.fixture
def row_path(monkeypatch):
  """Every batched chain environment with a wide row brings a row scratch, whatever its size."""
  monkeypatch.setattr(base.Environment, 'row_path_min_bytes', 0)


def _on_row_path(env):
  return bool(eu.raw(env)._call_desc.row_scratch)


def _check(ts, want, msg):
  st, r, d, o = want
  gst, gr, gd, go = eu.to_np(ts)
  np.testing.assert_array_equal(gst, st, err_msg=msg)
  live = st != 0
  np.testing.assert_array_equal(eu.f32_bits(gr[live]), eu.f32_bits(r[live].astype(np.float32)), err_msg='reward ' + msg)
  np.testing.assert_array_equal(gd[live], d[live].astype(np.float32), err_msg='discount ' + msg)
  assert (gr[~live] == 0).all() and (gd[~live] == 1).all(), msg
  np.testing.assert_array_equal(eu.f32_bits(go), eu.f32_bits(o), err_msg='observation ' + msg)


@pytest.mark.parametrize('family,kwargs,wrap', CASES)
@pytest.mark.parametrize('batch,lane_offset', [(1, 0), (1000, 0), (4099, (1 << 32) - 17)])
def test_row_path_bit_exact(row_path, family, kwargs, wrap, batch, lane_offset):
  seed = 4321
  env = eu.make_env(family, kwargs, batch=batch, lane_offset=lane_offset, seed=seed, wrap=wrap)
  lds = eu.make_env(family, kwargs, batch=batch, lane_offset=lane_offset, seed=seed, wrap=wrap)
  eu.raw(lds)._ensure_allocated()
  eu.raw(lds)._call_desc.row_scratch = None                       # the same environment on the one-launch LDS path
  orc = coracle.OracleEnv(family, kwargs, np.arange(lane_offset, lane_offset + batch, dtype=np.uint64), seed=seed, wrap=wrap)
  rng = np.random.default_rng(batch + len(str(kwargs)))
  for t in range(45):
    a = rng.integers(0, orc.num_actions, size=batch).astype(np.int32)
    force = t in (11, 12)
    at = torch.from_numpy(a).cuda()
    ts, tl = (env.reset(), lds.reset()) if force else (env.step(at), lds.step(at))
    _check(ts, orc.call(a, t, force_reset=force), f'{family} {kwargs} t={t}')
    for u, v in zip(eu.to_np(ts), eu.to_np(tl)):
      np.testing.assert_array_equal(u, v, err_msg=f'row path vs LDS path t={t}')
  assert _on_row_path(env) and not _on_row_path(lds)
  info = env.bsuite_info()
  for k, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(info[k].cpu().numpy(), v, err_msg=k)
  torch.testing.assert_close(eu.raw(env).episode_counters(), eu.raw(lds).episode_counters(), rtol=0, atol=0)


def test_default_threshold_and_short_rows(row_path):
  """Rows of at most 8 floats never take the path (their thread stores them itself); the scalar view never does; the
  default threshold is by bytes of observations per step."""
  assert not _on_row_path(_stepped(eu.make_env('umbrella_chain', dict(chain_length=3, n_distractor=5), batch=64, lane_offset=0, seed=1)))
  assert not _on_row_path(_stepped(eu.make_env('memory_chain', dict(memory_length=3, num_bits=1), batch=64, lane_offset=0, seed=1)))
  assert not _on_row_path(_stepped(eu.make_env('catch', dict(), batch=64, lane_offset=0, seed=1)))
  scalar = bsuite_amd.load_from_id('umbrella_length/3')
  scalar.reset()
  assert not _on_row_path(scalar)


def _stepped(env):
  env.step(torch.zeros(eu.raw(env).batch_size, dtype=torch.int32, device='cuda'))
  return env


def test_threshold_by_bytes(monkeypatch):
  assert base.Environment.row_path_min_bytes is None                       # the product default: never (measured slower)
  assert not _on_row_path(_stepped(bsuite_amd.load_from_id('umbrella_length/10', batch=1 << 17, seed=1)))
  monkeypatch.setattr(base.Environment, 'row_path_min_bytes', 8 << 20)
  small = _stepped(bsuite_amd.load_from_id('umbrella_length/10', batch=4096, seed=1))                # 92 B x 4096 < 8 MiB
  assert not _on_row_path(small)
  big = _stepped(bsuite_amd.load_from_id('umbrella_length/10', batch=1 << 17, seed=1))              # 12 MB
  assert _on_row_path(big)


@pytest.mark.parametrize('family,kwargs,wrap,by_step', [
    ('umbrella_chain', dict(chain_length=4, n_distractor=20), None, False),
    ('memory_chain', dict(memory_length=2, num_bits=40), ('noise', 0.3), True)])
def test_row_path_under_the_logging_wrapper(row_path, family, kwargs, wrap, by_step):
  B, T, seed = 777, 60, 6
  env = wrappers.Logging(eu.make_env(family, kwargs, batch=B, lane_offset=9, seed=seed, wrap=wrap), None, log_by_step=by_step)
  orc = coracle.OracleEnv(family, kwargs, np.arange(9, 9 + B, dtype=np.uint64), seed=seed, wrap=wrap)
  trk = logging_oracle.TrackOracle(B, list(orc.bsuite_info()), log_by_step=by_step)
  rng = np.random.default_rng(3)
  for t in range(T):
    a = rng.integers(0, 2, size=B).astype(np.int32)
    ts = env.step(torch.from_numpy(a).cuda())
    want = orc.call(a, t)
    _check(ts, want, f'logging {family} t={t}')
    trk.track(want[0], want[1], orc.bsuite_info())
  assert _on_row_path(env)
  assert [len(r) for r in env.all_rows()] == [len(r) for r in trk.rows]
  np.testing.assert_array_equal(env.counters()['total_return'].cpu().numpy(), trk.total_return)


def test_row_path_in_mt19937_exact_mode(row_path):
  """rng='mt19937': the lanes' own RandomState generators feed the same sink — equal to the LDS path draw for draw."""
  kw = dict(chain_length=5, n_distractor=45)
  B = 96
  a_env = eu.make_env('umbrella_chain', kw, batch=B, lane_offset=0, seed=11, rng='mt19937')
  b_env = eu.make_env('umbrella_chain', kw, batch=B, lane_offset=0, seed=11, rng='mt19937')
  eu.raw(b_env)._ensure_allocated()
  eu.raw(b_env)._call_desc.row_scratch = None
  m_env = eu.make_env('memory_chain', dict(memory_length=3, num_bits=37), batch=B, lane_offset=0, seed=5, rng='mt19937')
  n_env = eu.make_env('memory_chain', dict(memory_length=3, num_bits=37), batch=B, lane_offset=0, seed=5, rng='mt19937')
  eu.raw(n_env)._ensure_allocated()
  eu.raw(n_env)._call_desc.row_scratch = None
  rng = np.random.default_rng(0)
  for t in range(40):
    a = torch.from_numpy(rng.integers(0, 2, size=B).astype(np.int32)).cuda()
    for x, y in ((a_env.step(a), b_env.step(a)), (m_env.step(a), n_env.step(a))):
      for u, v in zip(eu.to_np(x), eu.to_np(y)):
        np.testing.assert_array_equal(u, v, err_msg=f't={t}')
  assert _on_row_path(a_env) and _on_row_path(m_env)


def test_row_path_rollout_and_graph(row_path):
  """rollout(T) keeps the fused one-launch kernel (== T step() calls on the row path); a captured HIP graph of row-path
  steps replays."""
  kw = dict(chain_length=7, n_distractor=20)
  B, T = 3000, 12
  env = eu.make_env('umbrella_chain', kw, batch=B, lane_offset=0, seed=2, device_step_counter=True, num_buffers=1)
  ref = eu.make_env('umbrella_chain', kw, batch=B, lane_offset=0, seed=2)
  g = torch.Generator(device='cuda'); g.manual_seed(4)
  acts = torch.randint(2, (T, B), generator=g, device='cuda', dtype=torch.int32)
  roll = ref.rollout(acts)
  stepped = [tuple(x.copy() for x in eu.to_np(env.step(acts[t]))) for t in range(T)]
  for t in range(T):
    for u, v in zip(stepped[t], (roll.step_type[t], roll.reward[t], roll.discount[t], roll.observation[t])):
      np.testing.assert_array_equal(u, v.cpu().numpy(), err_msg=f'rollout slice {t}')
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  graph = torch.cuda.CUDAGraph()
  static = acts[0].clone()
  with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
      out = env.step(static)
  torch.cuda.current_stream().wait_stream(side)
  ref2 = eu.make_env('umbrella_chain', kw, batch=B, lane_offset=0, seed=2)
  for t in range(T):                                                # bring the eager twin to the same call index
    ref2.step(acts[t])
  for rep in range(9):
    static.copy_(acts[rep])
    graph.replay()
    want = ref2.step(acts[rep])
    torch.cuda.synchronize()
    for u, v in zip(eu.to_np(out), eu.to_np(want)):
      np.testing.assert_array_equal(u, v, err_msg=f'replay {rep}')


def test_c_abi_checks_the_scratch():
  env = eu.make_env('umbrella_chain', dict(chain_length=4, n_distractor=20), batch=512, lane_offset=0, seed=1)
  raw = eu.raw(env)
  raw._ensure_allocated()
  scratch = raw._row_scratch()
  assert scratch.numel() * 4 == _native.lib.bsx_row_scratch_bytes(_native.FAMILY_IDS['umbrella_chain'], 23, 512) == 4 * (8 * 46 + 512)
  raw._call_desc.row_scratch = scratch.data_ptr() + 4             # not 16-byte aligned
  with pytest.raises(RuntimeError, match='aligned'):
    env.step(torch.zeros(512, dtype=torch.int32, device='cuda'))
  raw._call_desc.row_scratch = scratch.data_ptr()
  env.step(torch.zeros(512, dtype=torch.int32, device='cuda'))


IDS = ['umbrella_length/3', 'umbrella_length/22', 'umbrella_distract/3', 'umbrella_distract/12', 'umbrella_distract/22',
       'memory_size/2', 'memory_size/9', 'memory_size/12', 'memory_size/16', 'memory_len/6', 'bandit/3', 'catch/0',
       'deep_sea/2', 'cartpole/0', 'discounting_chain/4']


@pytest.mark.parametrize('pipelined', [False, True])
@pytest.mark.parametrize('lanes_per_id', [257, 2240])
def test_sweep_rows_in_the_store_stream_equal_rows_built_in_phase0(pipelined, lanes_per_id):
  """The whole-sweep group with the chains' wide rows written by the phase-1 store stream (the default) == the same
  group with phase 0 building them as LDS bit planes (rows_in_stream=False), TimeStep for TimeStep, both schedules."""
  total, seed, reps = len(IDS) * lanes_per_id + 5, 31, 27
  outs = {}
  for rows_in_stream in (True, False):
    batch = sb.SweepBatch(IDS, total, seed=seed)
    acts = batch.random_actions(seed=3, ring=4)
    o = batch.prepare_groups(acts, pipelined=pipelined, rows_in_stream=rows_in_stream, split=False)
    n_stream = sum(1 for v in batch._row_scratch.values() if v is not None)
    wide = sum(1 for e in batch.envs if eu.raw(e)._abi_name in ('memory_chain', 'umbrella_chain') and np.prod(e.observation_spec().shape) > 8)
    assert wide == 7                                                                  # (memory_len/6, memory_size/2, umbrella_distract/3: short rows)
    assert n_stream == ((wide * (2 if pipelined else 1)) if rows_in_stream else 0)
    for _ in range(reps):
      last = batch.step_grouped()
    batch.sync()
    o = last if pipelined else o
    outs[rows_in_stream] = [tuple(x.copy() for x in eu.to_np(ts)) for ts in o]
    outs[(rows_in_stream, 'info')] = [{k: v.cpu().numpy().copy() for k, v in e.bsuite_info().items()} for e in batch.envs]
    batch.release_groups()
    del batch
  for bid, x, y in zip(IDS, outs[True], outs[False]):
    for u, v in zip(x, y):
      np.testing.assert_array_equal(u, v, err_msg=bid)
  for bid, x, y in zip(IDS, outs[(True, 'info')], outs[(False, 'info')]):
    for k in x:
      np.testing.assert_array_equal(x[k], y[k], err_msg=f'{bid} {k}')


def test_pipelined_pair_must_not_share_a_row_scratch():
  """bsx_group_step_pipelined: BSX_EMODE when both groups hand a chain segment the SAME scratch (the stream of step s
  would decode rows the advance of step s+1 is writing); a segment set again with its own scratch is accepted."""
  env = bsuite_amd.load_from_id('umbrella_length/3', batch=3000, device_step_counter=True)
  acts = torch.zeros(3000, dtype=torch.int32, device='cuda')
  scratch = eu.raw(env)._row_scratch()
  other = eu.raw(env)._row_scratch(fresh=True)
  stream = torch.cuda.current_stream().cuda_stream

  def group(rs):
    h = ctypes.c_void_p()
    _native.check(_native.lib.bsx_group_create(_native.FAMILY_IDS['sweep_mixed'], 1, ctypes.byref(h)), 'bsx_group_create')
    _native.check(eu.raw(env)._group_set(h, 0, acts, row_scratch=rs), 'bsx_group_set_umbrella_chain')
    _native.check(_native.lib.bsx_group_commit(h), 'bsx_group_commit')
    return h

  a, b, c = group(scratch), group(scratch), group(other)
  assert _native.lib.bsx_group_step_pipelined(a, b, stream) == _native.BSX_EMODE
  _native.check(_native.lib.bsx_group_step_phase(a, 0, stream), 'phase 0')
  _native.check(_native.lib.bsx_group_step_pipelined(a, c, stream), 'pipelined with separate scratches')
  torch.cuda.synchronize()
  for h in (a, b, c):
    _native.lib.bsx_group_destroy(h)
