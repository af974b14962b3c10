"""GPU: the heterogeneous sweep (BASELINE config 5) — segments of different families advanced by one
captured HIP graph reproduce, lane for lane, standalone environments stepped eagerly; plus the
host-side segment table / bin packing."""
import numpy as np
import pytest
import torch

import bsuite_amd
from bsuite_amd import sweep
from bsuite_amd import sweep_batch as sb
from tests import engine_util as eu
from tests import golden_util as gu

IDS = ['bandit/3', 'bandit_noise/7', 'catch/0', 'catch_scale/5', 'deep_sea/2', 'deep_sea_stochastic/1',
       'discounting_chain/4', 'memory_len/6', 'memory_size/9', 'umbrella_length/3', 'umbrella_distract/12',
       'cartpole/0', 'cartpole_noise/4', 'cartpole_swingup/7', 'mountain_car/0', 'mountain_car_scale/9',
       'mnist/0', 'mnist_noise/2']


def test_segment_table_and_bin_packing():
  table = sb.segment_table(sweep.SWEEP, 1 << 20)
  assert len(table) == 468 and table[0] == ('bandit/0', 0, 2240)
  assert table[-1][2] == 2240 + (1 << 20) - 2240 * 468 and table[-1][1] + table[-1][2] == 1 << 20
  for (_, b0, n0), (_, b1, _) in zip(table, table[1:]):
    assert b0 + n0 == b1
  costs = [5.0, 1.0, 1.0, 1.0, 4.0, 3.0, 2.0, 2.0]
  ranks = sb.assign_segments(costs, 3)
  loads = [sum(c for c, r in zip(costs, ranks) if r == k) for k in range(3)]
  assert max(loads) - min(loads) <= 2.0 and sorted(set(ranks)) == [0, 1, 2]


@pytest.mark.gpu
def test_graph_replayed_sweep_equals_standalone_envs(tmp_path):
  from bsuite_amd.utils import datasets
  imgs, labels = gu.mnist_dataset()
  datasets.write_idx_files(str(tmp_path), imgs.view(np.uint8), labels)
  mn = dict(data_dir=str(tmp_path))
  kw = dict(mnist=mn, mnist_noise=mn, mnist_scale=mn)
  total, seed, reps = 18 * 300 + 7, 123, 25
  batch = sb.SweepBatch(IDS, total, seed=seed, env_kwargs=kw, num_streams=4)
  assert batch.lanes() == total and len(batch.envs) == len(IDS)
  acts = batch.random_actions(seed=1)
  batch.capture(acts)                                    # call 0 eager + capture
  for _ in range(reps):
    outs = batch.replay()
  torch.cuda.synchronize()
  for (bid, begin, lanes), a, out in zip(batch.segments, acts, outs):
    name = bid.split('/')[0]
    ekw = dict(kw.get(name, {}))
    if sweep.SETTINGS[bid].get('seed', 0) is None or 'seed' not in sweep.SETTINGS[bid]:
      ekw['seed'] = seed
    ref = bsuite_amd.load_from_id(bid, batch=lanes, lane_offset=begin, num_buffers=1, **ekw)
    for _ in range(reps + 1):
      ts = ref.step(a)
    physics = name.startswith(('cartpole', 'mountain_car'))
    for x, y in zip(eu.to_np(out), eu.to_np(ts)):
      np.testing.assert_array_equal(x, y, err_msg=bid)   # same kernels, same draws: identical even for f32 physics
    for k, v in ref.bsuite_info().items():
      torch.testing.assert_close(batch.envs[batch.segments.index((bid, begin, lanes))].bsuite_info()[k], v,
                                 rtol=0, atol=0)
    del physics
  s = batch.summary()
  assert set(s) == set(IDS) and s['bandit/3']['episodes_started'] >= 300


@pytest.mark.gpu
def test_sweep_rank_assignment_is_a_partition():
  a = sb.SweepBatch(IDS[:8], 8 * 64, rank=0, world_size=2, seed=1)
  b = sb.SweepBatch(IDS[:8], 8 * 64, rank=1, world_size=2, seed=1)
  ids_a = [s[0] for s in a.segments]
  ids_b = [s[0] for s in b.segments]
  assert sorted(ids_a + ids_b) == sorted(IDS[:8]) and not set(ids_a) & set(ids_b)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['whole_sweep', 'whole_sweep_split', 'whole_sweep_split_rows', 'whole_sweep_pipelined', 'whole_sweep_graph', 'eager', 'streams', 'graph', 'graph_phased', 'per_family',
                                  'pairs_per_family'])
def test_grouped_sweep_equals_standalone_envs(tmp_path, mode):
  """One grouped launch per family advances every segment exactly like its standalone environment
  (eager on one stream, or as concurrent branches of one captured HIP graph)."""
  from bsuite_amd.utils import datasets
  imgs, labels = gu.mnist_dataset()
  datasets.write_idx_files(str(tmp_path), imgs.view(np.uint8), labels)
  mn = dict(data_dir=str(tmp_path))
  kw = dict(mnist=mn, mnist_noise=mn, mnist_scale=mn)
  ids = IDS + ['deep_sea/7', 'catch_noise/11', 'bandit_scale/3', 'umbrella_distract/3', 'memory_size/2',
               'cartpole_swingup/1', 'discounting_chain/9', 'mountain_car_noise/5', 'umbrella_distract/22',
               'memory_size/16']
  total, seed, reps = len(ids) * 257 + 3, 77, 23
  batch = sb.SweepBatch(ids, total, seed=seed, env_kwargs=kw)
  acts = batch.random_actions(seed=2)
  outs = batch.prepare_groups(acts, mix_small=(mode != 'per_family'), mix_pairs=(mode not in ('per_family', 'pairs_per_family')),
                              mix_all=mode.startswith('whole_sweep'), pipelined=(mode == 'whole_sweep_pipelined'),
                              split=mode.startswith('whole_sweep_split'), rows_in_stream=(mode == 'whole_sweep_split_rows'))
  assert batch._split == mode.startswith('whole_sweep_split')
  assert any(v is not None for v in batch._row_scratch.values()) == (mode == 'whole_sweep_split_rows')
  ahead = 0
  if mode == 'whole_sweep_pipelined':
    # two groups (state columns swapped, own TimeStep buffers) alternate: ONE launch per sweep step carries
    # the store stream of step s beside the lane advance of step s+1, so the lanes run one advance ahead
    assert len(batch._groups) == 2
    ahead = 1
  elif mode.startswith('whole_sweep'):
    assert len(batch._groups) == 1               # ONE group: two launches per sweep step, nothing else
  elif mode == 'per_family':
    assert len(batch._groups) == 9               # families, not segments
  elif mode == 'pairs_per_family':
    assert len(batch._groups) == 4               # deep_sea, catch, mnist + one mixed small-observation group
  else:
    assert len(batch._groups) == 2               # one mixed two-kernel group + one mixed small-observation group
  if mode in ('graph', 'graph_phased', 'pairs_per_family', 'whole_sweep_graph'):
    # runs sweep step 0 eagerly, captures one step; phased: advance kernels + small groups on one
    # branch, every observation stream kernel on another as soon as its advance kernel is done
    assert batch.capture_grouped(num_streams=4 if mode == 'graph' else 2, phased=(mode != 'graph')) is outs
    for _ in range(reps - 1):
      batch.replay_grouped()
  elif mode == 'streams':                        # eager on the batch's two HIP streams
    for _ in range(reps):
      batch.step_grouped_streams()
    batch.join_streams()
  else:
    for _ in range(reps):
      last = batch.step_grouped()
    if ahead:
      outs = last                                # the TimeSteps of the step whose stream ran last
  batch.sync()
  assert all(eu.raw(e).step_index == reps + ahead for e in batch.envs)
  for (bid, begin, lanes), a, out, env in zip(batch.segments, acts, outs, batch.envs):
    name = bid.split('/')[0]
    ekw = dict(kw.get(name, {}))
    if sweep.SETTINGS[bid].get('seed', 0) is None or 'seed' not in sweep.SETTINGS[bid]:
      ekw['seed'] = seed
    ref = bsuite_amd.load_from_id(bid, batch=lanes, lane_offset=begin, num_buffers=1, **ekw)
    for _ in range(reps):
      ts = ref.step(a)
    for x, y in zip(eu.to_np(out), eu.to_np(ts)):
      np.testing.assert_array_equal(x, y, err_msg=bid)
    for _ in range(ahead):
      ref.step(a)
    for k, v in ref.bsuite_info().items():
      torch.testing.assert_close(env.bsuite_info()[k], v, rtol=0, atol=0)
    torch.testing.assert_close(eu.raw(env).episode_counters(), eu.raw(ref).episode_counters(), rtol=0, atol=0)
  batch.release_groups()
  if ahead:                                      # the lanes come back in the environments' own state columns
    for (bid, begin, lanes), a, env in zip(batch.segments, acts, batch.envs):
      if bid.split('/')[0] not in ('deep_sea', 'catch', 'catch_noise', 'mnist'):
        continue
      name = bid.split('/')[0]
      ekw = dict(kw.get(name, {}))
      if sweep.SETTINGS[bid].get('seed', 0) is None or 'seed' not in sweep.SETTINGS[bid]:
        ekw['seed'] = seed
      ref = bsuite_amd.load_from_id(bid, batch=lanes, lane_offset=begin, num_buffers=1, **ekw)
      for _ in range(reps + 1):
        ref.step(a)
      np.testing.assert_array_equal(eu.raw(env).state_dict()['state'].cpu().numpy(),
                                    eu.raw(ref).state_dict()['state'].cpu().numpy(), err_msg=bid)


@pytest.mark.gpu
def test_group_set_rejects_actions_it_would_misread():
  """grouped launches read the action tensor in place every step — an int64 / strided /
  wrong-shape tensor must be refused, not reinterpreted."""
  batch = sb.SweepBatch(['catch/0', 'bandit/0'], 600, seed=1)
  acts = batch.random_actions(seed=0)
  with pytest.raises(ValueError):
    batch.prepare_groups([acts[0].to(torch.int64), acts[1]])
  with pytest.raises(ValueError):
    batch.prepare_groups([acts[0][:-1], acts[1]])
  with pytest.raises(ValueError):
    batch.prepare_groups([torch.zeros((300, 2), dtype=torch.int32, device='cuda')[:, 0], acts[1]])
  batch.prepare_groups(acts)                                   # and the right ones are accepted
  batch.step_grouped()
  batch.sync()


@pytest.mark.gpu
def test_pipelined_groups_refuse_the_other_schedules():
  batch = sb.SweepBatch(['catch/0', 'bandit/0', 'deep_sea/0'], 900, seed=1)
  acts = batch.random_actions(seed=0)
  with pytest.raises(ValueError):
    batch.prepare_groups(acts, mix_all=False, pipelined=True)      # needs the whole-sweep group
  outs = batch.prepare_groups(acts, pipelined=True)
  assert len(outs) == 2 and len(batch._groups) == 2
  with pytest.raises(RuntimeError):
    batch.capture_grouped()
  with pytest.raises(RuntimeError):
    batch.step_grouped_streams()
  first = batch.step_grouped()
  assert first is outs[0] and batch.step_grouped() is outs[1] and batch.step_grouped() is outs[0]
  batch.sync()
  batch.release_groups()


@pytest.mark.gpu
def test_pipelined_catch_segments_with_odd_lane_counts(tmp_path):
  """Python and C must agree on WHICH catch segments phase 0 writes itself (fused tiles, one state column)
  and which stay two-kernel segments of the store stream (second column needed): both now use
  BSX_FUSED_CATCH_MAX_CELLS and nothing else, so an odd lane count — lanes x 50 floats not a multiple of 4 — is fused
  like any other.  The segments are big enough (hundreds of workgroups) that a stream reading a column the next
  step's advance is already rewriting would show: every step of every lane is compared with a stand-alone env."""
  ids = ['catch/0', 'catch_noise/3', 'catch_scale/7', 'deep_sea/2', 'cartpole/0', 'bandit/1']
  total, seed, reps = len(ids) * 60001, 5, 14
  batch = sb.SweepBatch(ids, total, seed=seed)
  assert all(lanes % 2 == 1 and (lanes * 50) % 4 != 0 for _, _, lanes in batch.segments[:3])
  acts = batch.random_actions(seed=3, ring=4)
  refs = []
  for (bid, begin, lanes) in batch.segments:
    ekw = {'seed': seed} if (sweep.SETTINGS[bid].get('seed', 0) is None or 'seed' not in sweep.SETTINGS[bid]) else {}
    refs.append(bsuite_amd.load_from_id(bid, batch=lanes, lane_offset=begin, num_buffers=1, **ekw))
  batch.prepare_groups(acts, pipelined=True)
  for s in range(reps):
    outs = batch.step_grouped()
    batch.sync()
    for (bid, begin, lanes), a, out, ref in zip(batch.segments, acts, outs, refs):
      ts = ref.step(a[s % 4])
      for x, y in zip(eu.to_np(out), eu.to_np(ts)):
        np.testing.assert_array_equal(x, y, err_msg=f'{bid} step {s}')
  batch.release_groups()


@pytest.mark.gpu
def test_step_pipelined_refuses_a_stream_segment_without_a_second_state_column():
  """bsx_group_step_pipelined returns BSX_EMODE for a two-kernel segment that keeps workgroups in the store stream
  but was set without state_alt (the stream of step s would read the column the advance of step s+1 writes)."""
  import ctypes
  from bsuite_amd import _native
  env = bsuite_amd.load_from_id('deep_sea/2', batch=5000, device_step_counter=True)
  acts = torch.zeros(5000, dtype=torch.int32, device='cuda')
  groups = []
  for _ in range(2):
    h = ctypes.c_void_p()
    _native.check(_native.lib.bsx_group_create(_native.FAMILY_IDS['sweep_mixed'], 1, ctypes.byref(h)), 'bsx_group_create')
    _native.check(env._group_set(h, 0, acts), 'bsx_group_set_deep_sea')      # no state_alt
    _native.check(_native.lib.bsx_group_commit(h), 'bsx_group_commit')
    groups.append(h)
  stream = torch.cuda.current_stream().cuda_stream
  assert _native.lib.bsx_group_step_pipelined(groups[0], groups[1], stream) == _native.BSX_EMODE
  _native.check(_native.lib.bsx_group_step(groups[0], stream), 'bsx_group_step')   # in order it is a valid group
  torch.cuda.synchronize()
  for h in groups:
    _native.lib.bsx_group_destroy(h)


@pytest.mark.gpu
def test_enable_logging_refused_while_groups_are_prepared():
  from bsuite_amd.utils import wrappers
  batch = sb.SweepBatch(['catch/0', 'bandit/0'], 600, seed=1)
  acts = batch.random_actions(seed=0)
  batch.prepare_groups(acts, pipelined=True)
  with pytest.raises(RuntimeError, match='release_groups'):
    wrappers.Logging(batch.envs[0], None)
  batch.release_groups()
  wrappers.Logging(batch.envs[0], None)                      # fine afterwards


@pytest.mark.gpu
@pytest.mark.parametrize('pipelined', [False, True])
def test_action_ring_feeds_fresh_actions_every_group_step(tmp_path, pipelined):
  """bsx_call_t.action_ring: every segment reads row (sweep step mod R) of a pre-generated [R, lanes] ring on the
  device — 11 steps over a ring of 4 wrap it nearly three times — and reproduces stand-alone environments fed the
  same rows by the host.  Rings that are not a power of two are refused."""
  from bsuite_amd.utils import datasets
  imgs, labels = gu.mnist_dataset()
  datasets.write_idx_files(str(tmp_path), imgs.view(np.uint8), labels)
  mn = dict(data_dir=str(tmp_path))
  kw = dict(mnist=mn, mnist_noise=mn, mnist_scale=mn)
  seed, reps, ring = 31, 11, 4
  batch = sb.SweepBatch(IDS, len(IDS) * 300 + 1, seed=seed, env_kwargs=kw)
  with pytest.raises(ValueError):
    batch.random_actions(seed=1, ring=3)
  bad = [torch.zeros((3, l), dtype=torch.int32, device='cuda') for _, _, l in batch.segments]
  with pytest.raises(ValueError):
    batch.prepare_groups(bad)
  acts = batch.random_actions(seed=5, ring=ring)
  assert all(a.shape == (ring, l) for a, (_, _, l) in zip(acts, batch.segments))
  assert any(not torch.equal(a[0], a[1]) for a in acts)
  batch.prepare_groups(acts, pipelined=pipelined)
  for _ in range(reps):
    outs = batch.step_grouped()
  batch.sync()
  for (bid, begin, lanes), a, out, env in zip(batch.segments, acts, outs, batch.envs):
    name = bid.split('/')[0]
    ekw = dict(kw.get(name, {}), seed=seed)
    ref = bsuite_amd.load_from_id(bid, batch=lanes, lane_offset=begin, num_buffers=1, **ekw)
    for s in range(reps):
      ts = ref.step(a[s % ring])
    for x, y in zip(eu.to_np(out), eu.to_np(ts)):
      np.testing.assert_array_equal(x, y, err_msg=bid)
  batch.release_groups()


@pytest.mark.gpu
def test_eager_and_grouped_steps_alternate_on_one_counter():
  """a whole-sweep group bumps the shared call counter itself with plain device accesses; eager step()
  calls of the same segments bump it with bsx_counter_add.  Stream-ordered (one HIP stream) the two may alternate
  freely and every segment sees consecutive call indices."""
  ids = ['catch/0', 'bandit/3', 'deep_sea/2', 'cartpole/1', 'umbrella_distract/12', 'memory_size/9']
  seed = 9
  batch = sb.SweepBatch(ids, len(ids) * 500, seed=seed)
  acts = batch.random_actions(seed=2)
  batch.prepare_groups(acts)
  pattern = 'gegggeegge'
  for c in pattern:
    outs = batch.step_grouped() if c == 'g' else batch.step(acts)
  batch.sync()
  assert int(batch._step_counter.item()) == len(pattern)
  for (bid, begin, lanes), a, out in zip(batch.segments, acts, outs):
    ref = bsuite_amd.load_from_id(bid, batch=lanes, lane_offset=begin, num_buffers=1, seed=seed)
    for _ in pattern:
      ts = ref.step(a)
    for x, y in zip(eu.to_np(out), eu.to_np(ts)):
      np.testing.assert_array_equal(x, y, err_msg=bid)
  batch.release_groups()


@pytest.mark.gpu
@pytest.mark.parametrize('mix_all', [True, False])
def test_grouped_mnist_on_a_dataset_beyond_l2_equals_standalone_envs(tmp_path, mix_all):
  """An image table larger than the chip's 32 MiB of L2 (the real MNIST's class) switches the mnist observation stream to
  non-temporal stores (csrc/mnist.hip: `mnist_observe_args.nt`, per segment) — stand-alone, inside the whole-sweep group
  (6 KiB-runs per wave) and inside a per-family pair group: same TimeSteps and bsuite_info as the stand-alone environments."""
  from bsuite_amd.utils import datasets
  rng = np.random.default_rng(4)
  imgs = rng.integers(0, 256, size=(43000, 28, 28), dtype=np.uint8)
  labels = rng.integers(0, 10, size=43000).astype(np.uint8)
  datasets.write_idx_files(str(tmp_path), imgs, labels, imgs[:16], labels[:16])
  mn = dict(data_dir=str(tmp_path))
  kw = dict(mnist=mn, mnist_noise=mn, mnist_scale=mn)
  ids = ['mnist/0', 'deep_sea/3', 'mnist_noise/7', 'bandit/2', 'mnist_scale/12', 'catch/1']
  total, seed, reps = len(ids) * 700 + 5, 31, 6
  batch = sb.SweepBatch(ids, total, seed=seed, env_kwargs=kw)
  acts = batch.random_actions(seed=9)
  outs = batch.prepare_groups(acts, mix_all=mix_all)
  for _ in range(reps):
    batch.step_grouped()
  batch.sync()
  for (bid, begin, lanes), a, out, env in zip(batch.segments, acts, outs, batch.envs):
    name = bid.split('/')[0]
    ekw = dict(kw.get(name, {}))
    if sweep.SETTINGS[bid].get('seed', 0) is None or 'seed' not in sweep.SETTINGS[bid]:
      ekw['seed'] = seed
    ref = bsuite_amd.load_from_id(bid, batch=lanes, lane_offset=begin, num_buffers=1, **ekw)
    for _ in range(reps):
      ts = ref.step(a)
    for x, y in zip(eu.to_np(out), eu.to_np(ts)):
      np.testing.assert_array_equal(x, y, err_msg=bid)
    for k, v in ref.bsuite_info().items():
      torch.testing.assert_close(env.bsuite_info()[k], v, rtol=0, atol=0)
  batch.release_groups()
