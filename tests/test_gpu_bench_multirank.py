"""The world>1 path of bench.py and of the summary reduction, executed for real on ONE GPU: all ranks
on cuda:0 (BSX_BENCH_SINGLE_DEVICE=1) with gloo as the collective backend (BSX_BENCH_BACKEND=gloo).
On a multi-GPU node the same code runs one rank per device over RCCL.

`python bench.py --gpus 2` must spawn its own ranks (the reference's parallel entry is self-launching:
bsuite/baselines/utils/pool.py:28-54), and because draws, synthetic actions and episode phases are keyed
by GLOBAL lane id / segment index, a run sharded over 2 ranks must reproduce the 1-rank run's episode
counts and bsuite_info sums exactly."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*argv, ranks_on_one_gpu=False):
  env = dict(os.environ)
  env.pop('WORLD_SIZE', None)
  env.pop('RANK', None)
  if ranks_on_one_gpu:
    env.update(BSX_BENCH_BACKEND='gloo', BSX_BENCH_SINGLE_DEVICE='1')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline'] + list(argv),
                     env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert p.returncode == 0, p.stderr[-3000:]
  lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, p.stdout            # exactly ONE JSON line, from rank 0
  return json.loads(lines[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize('workload', ['catch', 'cartpole'])
def test_two_self_launched_ranks_reproduce_one_rank(workload):
  common = ['--workload', workload, '--steps', '32', '--warmup', '8']
  one = _bench('--gpus', '1', '--lanes', '8192', *common)
  two = _bench('--gpus', '2', '--lanes', '4096', '--weak', *common, ranks_on_one_gpu=True)     # weak: 4096 per rank
  assert (one['n_gpus'], two['n_gpus']) == (1, 2)
  assert one['config']['global_lanes'] == two['config']['global_lanes'] == 8192
  assert two['scaling'] == 'weak'
  assert one['episodes_finished'] == two['episodes_finished'] > 0
  assert one['bsuite_info_sums'] == two['bsuite_info_sums']
  assert one['timed_mix'] == two['timed_mix']
  # the default for N > 1: BASELINE's batch is GLOBAL, split over the ranks (strong scaling)
  strong = _bench('--gpus', '2', '--lanes', '8192', *common, ranks_on_one_gpu=True)
  assert strong['scaling'] == 'strong' and strong['config']['lanes_per_gpu'] == 4096
  assert strong['config']['global_lanes'] == 8192
  assert strong['episodes_finished'] == one['episodes_finished']
  assert strong['bsuite_info_sums'] == one['bsuite_info_sums']


@pytest.mark.timeout(900)
def test_default_line_two_ranks_is_strong_with_weak_and_sweep_records():
  two = _bench('--gpus', '2', '--lanes', '8192', '--steps', '8', '--warmup', '4', ranks_on_one_gpu=True)
  assert two['n_gpus'] == 2 and two['config']['workload'].startswith('deep_sea/10')
  assert two['scaling'] == 'strong' and two['config']['lanes_per_gpu'] == 4096 and two['config']['global_lanes'] == 8192
  assert two['roofline']['alg_bytes'] == 3621 * 4096
  also = two['also']
  assert list(also)[0] == 'catch/0' and 'error' not in also['catch/0'] and two['also_common']['lanes_per_gpu'] == 4096
  assert also['weak']['scaling'] == 'weak' and also['weak']['lanes_per_gpu'] == 8192 and also['weak']['global_lanes'] == 16384
  sweep = also['sweep']
  assert sum(sweep['segments_per_rank']) == 468 and min(sweep['segments_per_rank']) > 0
  assert sweep['global_lanes'] == 8192 and sweep['action_ring'] == 16 and sweep['pipelined']['open_loop'] is True
  one = _bench('--gpus', '1', '--workload', 'sweep', '--lanes', '8192', '--steps', '100', '--warmup', '20')   # the sub-record's K / W
  assert one['episodes_finished'] == sweep['episodes_finished'] > 0


@pytest.mark.timeout(900)
def test_default_line_fits_the_drivers_tail():
  """The driver keeps the last 8000 characters of the line: the whole default line — headline, every BASELINE config
  with its roofline, the live cpu_baseline — must fit, catch/0 first among the sub-records."""
  env = dict(os.environ)
  env.pop('WORLD_SIZE', None)
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--lanes', '65536', '--steps', '8', '--warmup', '4'],
                     env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
  assert p.returncode == 0, p.stderr[-3000:]
  text = [l for l in p.stdout.splitlines() if l.startswith('{')][0]
  assert len(text) < 7800, len(text)
  line = json.loads(text)
  assert list(line['also'])[0] == 'catch/0'
  # every BASELINE config and every family north_star names, each with its roofline
  for k in ('catch/0', 'catch/0 r32', 'deep_sea/10 r16', 'cartpole/0', 'cartpole/0 r16', 'mountain_car/0', 'mountain_car/0 r16',
            'bandit/0', 'bandit/0 r16', 'discounting_chain/0', 'discounting_chain/0 r16', 'memory_len/10', 'memory_len/10 r16',
            'umbrella_length/10', 'umbrella_length/10 r16', 'mnist/0', 'sweep'):
    assert 'frac' in line['also'][k]['roofline'], (k, line['also'][k])
  assert line['also_common']['lanes_per_gpu'] == 65536 and line['also_common']['roofline']['peak'] == 8000.0
  assert line['also']['cartpole/0']['roofline']['frac_of_box_copy'] > 0 and line['host']['step_us'] > 0
  assert line['cpu_baseline']['kind'] in ('reference', 'port') and line['roofline']['frac'] > 0


@pytest.mark.timeout(1500)
def test_eight_self_launched_ranks_rehearsal():
  """The run the driver makes on an 8-GPU node (`bench.py --gpus 8`), rehearsed with all eight ranks on the one GPU
  there is here (gloo, BSX_BENCH_SINGLE_DEVICE): the default line parses, fits the driver's 8000-character tail, carries
  the strong record, the weak record and the sweep sharded over eight ranks (every bsuite_id on exactly one rank), and —
  draws, actions and episode phases being keyed by GLOBAL lane id / segment — reproduces the one-rank run's episode
  counts and bsuite_info sums exactly.  Reference fan-out: bsuite/baselines/utils/pool.py:28-54."""
  env = dict(os.environ)
  env.pop('WORLD_SIZE', None)
  env.pop('RANK', None)
  env.update(BSX_BENCH_BACKEND='gloo', BSX_BENCH_SINGLE_DEVICE='1')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--gpus', '8', '--lanes', '65536',
                      '--steps', '8', '--warmup', '4'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1400)
  assert p.returncode == 0, p.stderr[-3000:]
  lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1 and len(lines[0]) < 7800, (len(lines), [len(l) for l in lines])
  eight = json.loads(lines[0])
  assert eight['n_gpus'] == 8 and eight['scaling'] == 'strong'
  assert eight['config']['lanes_per_gpu'] == 8192 and eight['config']['global_lanes'] == 65536
  assert eight['roofline']['alg_bytes'] == 3621 * 8192
  also = eight['also']
  assert also['weak']['scaling'] == 'weak' and also['weak']['lanes_per_gpu'] == 65536 and also['weak']['global_lanes'] == 8 * 65536
  sweep = also['sweep']
  assert len(sweep['segments_per_rank']) == 8 and sum(sweep['segments_per_rank']) == 468 and min(sweep['segments_per_rank']) > 20
  assert sweep['global_lanes'] == 65536
  one = _bench('--gpus', '1', '--lanes', '65536', '--steps', '8', '--warmup', '4', '--no-also')
  assert one['episodes_finished'] == eight['episodes_finished'] > 0
  assert one['bsuite_info_sums'] == eight['bsuite_info_sums'] and one['timed_mix'] == eight['timed_mix']
  one_sweep = _bench('--gpus', '1', '--workload', 'sweep', '--lanes', '65536', '--steps', '100', '--warmup', '20')
  assert one_sweep['episodes_finished'] == sweep['episodes_finished'] > 0


# ---------------------------------------------------------------------------------------------------
def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


IDS = ['catch/3', 'bandit/1', 'deep_sea/2', 'memory_len/4', 'umbrella_distract/5', 'discounting_chain/2',
       'cartpole/1', 'mountain_car/2', 'cartpole_swingup/7', 'catch_noise/6', 'bandit_scale/9', 'deep_sea_stochastic/3']


def _summaries(rank, world, steps):
  import bsuite_amd
  from bsuite_amd import distributed as bdist
  from bsuite_amd import sweep_batch as sb
  # (1) one environment family sharded by lanes: local_summary(env) -> all_gather -> reduce
  off, n = bdist.shard_lanes(3000, rank, world)
  env = bsuite_amd.load_from_id('catch/0', batch=n, lane_offset=off, seed=7, device='cuda:0')
  lane = torch.arange(off, off + n, device='cuda:0')
  for t in range(steps):
    env.step(((lane * 7 + t * 3) % 3).to(torch.int32))
  vec, names = bdist.local_summary(env)
  red = bdist.reduce_summary(bdist.all_gather_summary(vec), names)
  # (2) a heterogeneous sweep sharded by whole segments: SweepBatch(rank, world).summary()
  batch = sb.SweepBatch(IDS, 1200, device='cuda:0', seed=5, rank=rank, world_size=world)
  acts = batch.random_actions(seed=3)
  for _ in range(steps):
    batch.step(acts)
  local = batch.summary()
  keys = sorted(local)
  blob = [None] * world
  if world > 1:
    dist.all_gather_object(blob, local)
  else:
    blob = [local]
  merged = {}
  for d in blob:
    assert not (set(d) & set(merged))              # every segment lives on exactly one rank
    merged.update(d)
  return red, merged, keys


def _worker(rank, world, port, steps, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  red, merged, keys = _summaries(rank, world, steps)
  q.put((rank, red, merged, keys))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_real_summaries_equal_one_rank():
  steps = 40
  ref_red, ref_merged, _ = _summaries(0, 1, steps)
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, steps, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted((q.get(timeout=500) for _ in range(2)), key=lambda x: x[0])
  for p in procs:
    p.join(60)
    assert p.exitcode == 0
  (_, red0, merged0, keys0), (_, red1, merged1, keys1) = res
  assert red0 == red1 == ref_red and ref_red['lanes'] == 3000 and ref_red['episodes_finished'] > 0
  assert merged0 == merged1 == ref_merged
  assert keys0 and keys1 and sorted(keys0 + keys1) == sorted(IDS)


@pytest.mark.timeout(600)
def test_rccl_code_path_on_one_rank():
  """backend nccl (= RCCL) with a one-rank process group: init, barrier, all-reduce of the timings and the
  all-gather of the summaries run on device tensors exactly as they do with N ranks."""
  env = dict(os.environ, BSX_BENCH_FORCE_PG='1')
  env.pop('WORLD_SIZE', None)
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-also', '--workload', 'catch',
                      '--lanes', '8192', '--steps', '32', '--warmup', '8'],
                     env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=500)
  assert p.returncode == 0, p.stderr[-3000:]
  line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
  plain = _bench('--no-also', '--workload', 'catch', '--lanes', '8192', '--steps', '32', '--warmup', '8')
  assert line['episodes_finished'] == plain['episodes_finished'] > 0
  assert line['bsuite_info_sums'] == plain['bsuite_info_sums']
