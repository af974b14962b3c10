"""bench.py — env-steps/sec of the batched bsuite step() path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload deep_sea|catch|...] [--lanes B]

One "step" = one env.step(actions) call on every lane of the batch (auto-reset calls included,
they are real API calls: bsuite/environments/base.py:61-62).  Default workload is BASELINE.json
configs[1]: deep_sea size=30 (bsuite_id deep_sea/10), 2^20 lanes per GPU, uniform random actions
pre-generated on the device (the batched analogue of bsuite/baselines/random/agent.py:35-37).
Every TimeStep field is materialised in HBM on every step (dense contract).  For N>1 the driver
launches one process per GPU (torch.distributed, backend nccl = RCCL); lanes shard with no
data-path collective; the only collective is the end-of-rollout all-gather of per-rank summaries.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# workload -> (bsuite_id, oracle family, oracle kwargs, obs_numel, state bytes in+out per lane)
WORKLOADS = {
    'deep_sea': ('deep_sea/10', 'deep_sea', dict(size=30, mapping_seed=42), 900, 8),
    'catch': ('catch/0', 'catch', dict(), 50, 8),
    'cartpole': ('cartpole/0', 'cartpole', dict(), 6, 48),
    'mountain_car': ('mountain_car/0', 'mountain_car', dict(), 3, 24),
    'bandit': ('bandit/0', 'bandit', dict(mapping_seed=0), 1, 8),
    'memory_len': ('memory_len/10', 'memory_chain', dict(memory_length=12, num_bits=1), 3, 24),
    'umbrella_length': ('umbrella_length/10', 'umbrella_chain', dict(chain_length=12, n_distractor=20), 23, 8),
    'discounting_chain': ('discounting_chain/0', 'discounting_chain', dict(mapping_seed=0), 2, 8),
    # the MNIST files cannot be fetched here: tests/golden/mnist_synthetic_dataset.npz (same idx wire format)
    'mnist': ('mnist/0', 'mnist', dict(), 784, 8),
}


def _synthetic_mnist():
  import numpy as np
  d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
  return d['images_u8'].view(np.int8), d['labels']


def algorithmic_bytes_per_step(obs_numel, state_bytes):
  # action i32 + reward f32 + discount f32 + step_type i8 + obs f32 + state in/out
  return 4 + 4 + 4 + 1 + 4 * obs_numel + state_bytes


def pmc_traffic(workload, lanes):
  """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN/<w>_pmc_traffic.json:
  WRITE_SIZE and FETCH_SIZE collected in separate runs, gfx950 corrections applied there)."""
  import glob
  hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', f'{workload}_pmc_traffic.json')))
  if not hits or lanes != (1 << 20):
    return None, None
  with open(hits[-1]) as f:
    d = json.load(f)
  return d['per_launch']['hbm_bytes'], os.path.relpath(hits[-1], ROOT)


def _oracle_loop(family, kwargs, num_actions, lanes, lane0, budget_s):
  """Steps one OracleEnv (its own lanes) for budget_s seconds; returns (env-steps, seconds)."""
  import numpy as np
  from oracle import coracle
  if family == 'mnist' and 'images' not in kwargs:
    kwargs = dict(kwargs)
    kwargs['images'], kwargs['labels'] = _synthetic_mnist()
  env = coracle.OracleEnv(family, kwargs, np.arange(lane0, lane0 + lanes, dtype=np.uint64), seed=42)
  rng = np.random.default_rng(lane0)
  acts = rng.integers(0, num_actions, size=(64, lanes)).astype(np.int32)
  for t in range(8):
    env.call(acts[t % 64], t)
  t0 = time.perf_counter()
  n = 0
  while time.perf_counter() - t0 < budget_s:
    for _ in range(16):
      env.call(acts[n % 64], 8 + n)
      n += 1
  return lanes * n, time.perf_counter() - t0


def cpu_baseline(family, kwargs, num_actions, budget_s=12.0, all_cores_budget_s=4.0):
  """The oracle (C restatement of the reference's numpy step) on ONE host core, bounded sample; plus
  the same loop on every host core at once (one process per core, each with its own OracleEnv) —
  the analogue of the reference's one-process-per-bsuite_id pool (bsuite/baselines/utils/pool.py:48)."""
  lanes = 4096
  steps, dt = _oracle_loop(family, kwargs, num_actions, lanes, 0, budget_s)
  out = dict(value=steps / dt, unit='env-steps/s', cores=1, kind='port',
             sample=f'{lanes} lanes x {steps // lanes} step() calls of {family} {kwargs} through '
                    f'oracle/oracle.c (gcc -O2, single thread, {dt:.1f} s)')
  cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  if cores > 1 and all_cores_budget_s > 0:
    # one PROCESS per core (threads would serialise on the interpreter lock between the short C calls)
    import subprocess
    code = ('import sys, json; sys.path.insert(0, %r); import bench; '
            'f, kw, na, lanes, lane0, b = sys.argv[1], json.loads(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), '
            'int(sys.argv[5]), float(sys.argv[6]); print(*bench._oracle_loop(f, kw, na, lanes, lane0, b))' % ROOT)
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, '-c', code, family, json.dumps(kwargs), str(num_actions), str(lanes),
                               str((j + 1) * lanes), str(all_cores_budget_s)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for j in range(cores)]
    res = []
    for pr in procs:
      o, _ = pr.communicate()
      try:
        a, b = o.split()
        res.append((float(a), float(b)))
      except ValueError:
        pass
    wall = time.perf_counter() - t0
    if res:
      out['all_cores'] = dict(value=sum(r[0] for r in res) / max(r[1] for r in res), cores=len(res),
                              sample=f'{len(res)} processes x {lanes} lanes x {all_cores_budget_s:.0f} s each, '
                                     f'{wall:.1f} s wall incl. start-up')
  return out


def bench_sweep(args, torch, dist, dev, rank, world):
  """BASELINE config 5: all 468 bsuite_ids as lane segments (2^20 lanes in total, split evenly per id,
  whole segments bin-packed over the ranks), one captured HIP graph per sweep step."""
  import tempfile
  import numpy as np
  from bsuite_amd import sweep_batch as sb
  from bsuite_amd.utils import datasets
  d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
  tmp = tempfile.mkdtemp(prefix='bsx_mnist_')
  datasets.write_idx_files(tmp, d['images_u8'], d['labels'])      # synthetic stand-in (no network)
  mn = dict(data_dir=tmp)
  batch = sb.SweepBatch(None, args.lanes, device=dev, seed=42, rank=rank, world_size=world,
                        num_streams=int(os.environ.get('BSX_SWEEP_STREAMS', '32')),
                        env_kwargs=dict(mnist=mn, mnist_noise=mn, mnist_scale=mn))
  acts = batch.random_actions(seed=1 + rank)
  mode = os.environ.get('BSX_SWEEP_MODE', 'grouped_graph')
  grouped = mode in ('grouped', 'grouped_graph')
  if grouped:
    batch.prepare_groups(acts, mix_small=os.environ.get('BSX_SWEEP_MIX_SMALL', '1') != '0')
    if mode == 'grouped_graph':              # group launches as concurrent branches of one HIP graph
      batch.capture_grouped(int(os.environ.get('BSX_SWEEP_STREAMS', '2')))
      batch.replay = batch.replay_grouped
    else:
      batch.replay = batch.step_grouped
  else:
    batch.capture(acts)

  def sync_all():
    torch.cuda.synchronize(dev)
    if world > 1:
      dist.barrier()
      torch.cuda.synchronize(dev)

  for _ in range(args.warmup):
    batch.replay()
  sync_all()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()
  for _ in range(args.steps):
    batch.replay()
  ev1.record()
  torch.cuda.synchronize(dev)
  wall = time.perf_counter() - t0
  sync_all()
  step_ms = ev0.elapsed_time(ev1) / args.steps
  local_bytes = sum(l * sb.bytes_per_step(int(np.prod(e.observation_spec().shape)))
                    for e, (_, _, l) in zip(batch.envs, batch.segments))
  if world > 1:
    t = torch.tensor([wall, step_ms, float(local_bytes)], dtype=torch.float64,
                     device=dev if dist.get_backend() == 'nccl' else 'cpu')
    mx = t.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    wall, step_ms, total_bytes = float(mx[0]), float(mx[1]), float(t[2])
  else:
    total_bytes = float(local_bytes)
  if rank == 0:
    achieved = total_bytes / world / (step_ms * 1e-3) / 1e9
    print(json.dumps({
        'metric': 'env-steps/sec', 'value': args.lanes * args.steps / wall, 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int32+f32',
        'data': 'synthetic (MNIST ids on a synthetic stand-in dataset)',
        'config': {'workload': 'sweep.SWEEP: 468 bsuite_ids as lane segments, random-action rollout, dense TimeStep',
                   'global_lanes': args.lanes, 'segments_on_rank0': len(batch.envs),
                   'sharding': f'whole segments bin-packed over {world} rank(s)'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBPS, 'traffic': None, 'kernel_ms': step_ms,
                     'algorithmic_bytes_per_launch': total_bytes / world},
        'launch': (f'{len(batch._groups)} grouped launches per sweep step ({len(batch.envs)} segments)'
                   + (' as concurrent branches of one HIP graph' if mode == 'grouped_graph' else '') if grouped else
                   f'one hipGraph per sweep step ({len(batch.envs)} segments over {batch.num_streams} streams)')}),
          flush=True)
  if world > 1:
    dist.destroy_process_group()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--workload', default='deep_sea', choices=sorted(WORKLOADS) + ['sweep'])
  ap.add_argument('--lanes', type=int, default=1 << 20, help='lanes per GPU')
  ap.add_argument('--strong', action='store_true',
                  help='strong scaling (SURVEY §8d): --lanes is the GLOBAL lane count, split evenly over the ranks '
                       '(default: weak scaling, --lanes per GPU)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--logging', action='store_true',
                  help='wrap the environment in the batched Logging wrapper (bookkeeping fused into the kernels)')
  ap.add_argument('--observation-mode', default='dense', choices=['dense', 'delta'],
                  help="'delta' (deep_sea, catch): persistent observation buffers patched in place; a "
                       'separate mode with its own byte accounting, NOT the dense contract of the headline')
  ap.add_argument('--rollout', type=int, default=0,
                  help='advance this many steps per entry-point call with env.rollout(actions[T,B]) '
                       '(fused T-step kernel for the small-observation families)')
  ap.add_argument('--graph', type=int, default=0,
                  help='capture this many consecutive step() launches into one HIP graph and replay it '
                       '(for the tiny families whose per-step kernel is shorter than a host launch)')
  args = ap.parse_args()

  import torch
  import torch.distributed as dist
  import bsuite_amd

  world = int(os.environ.get('WORLD_SIZE', '1'))
  if args.strong and args.workload != 'sweep':
    if args.lanes % world:
      raise SystemExit('--strong needs --lanes divisible by the number of ranks')
    args.lanes //= world
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  # Test hooks (single-GPU boxes): BSX_BENCH_BACKEND=gloo + BSX_BENCH_SINGLE_DEVICE=1 run all ranks
  # on cuda:0 so the multi-rank control flow can be exercised without a multi-GPU node.
  backend = os.environ.get('BSX_BENCH_BACKEND', 'nccl')
  if os.environ.get('BSX_BENCH_SINGLE_DEVICE'):
    local_rank = 0
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend)
  assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)

  if args.workload == 'sweep':
    return bench_sweep(args, torch, dist, dev, rank, world)

  def sync_all():
    torch.cuda.synchronize(dev)
    if world > 1:
      dist.barrier()
      torch.cuda.synchronize(dev)

  def measure(workload):
    """Times exactly args.steps step() calls of `workload` after args.warmup untimed ones."""
    bsuite_id, family, okw, obs_numel, state_bytes = WORKLOADS[workload]
    B = args.lanes
    delta = args.observation_mode == 'delta'
    if delta and family not in ('deep_sea', 'catch'):
      raise SystemExit('--observation-mode delta exists for deep_sea and catch only')
    extra = {}
    if family == 'mnist':
      extra['images'], extra['labels'] = _synthetic_mnist()
    env = bsuite_amd.load_from_id(bsuite_id, batch=B, device=dev, seed=42, lane_offset=rank * B,
                                  num_buffers=2, device_step_counter=bool(args.graph),
                                  observation_mode=args.observation_mode, **extra)
    if args.logging:
      # the Logging wrapper's per-lane bookkeeping + log-spaced snapshot rows, fused
      # into the same kernels (no logger object: rows stay in the device buffer)
      from bsuite_amd.utils import wrappers as _wrappers
      env = _wrappers.Logging(env, None)
    num_actions = env.action_spec().num_values
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    n_act = max(32, args.graph, args.rollout)
    actions = torch.randint(num_actions, (n_act, B), generator=gen, device=dev, dtype=torch.int32)

    if args.graph:
      assert args.steps % args.graph == 0 and args.warmup % args.graph == 0, '--steps/--warmup must be multiples of --graph'
      env.step(actions[0])                       # allocate outside capture
      side = torch.cuda.Stream(device=dev)
      side.wait_stream(torch.cuda.current_stream(dev))
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
          for t in range(args.graph):
            env.step(actions[t])
      torch.cuda.current_stream(dev).wait_stream(side)

      def run(n_steps):
        for _ in range(n_steps // args.graph):
          graph.replay()
    elif args.rollout:
      assert args.steps % args.rollout == 0 and args.warmup % args.rollout == 0, '--steps/--warmup must be multiples of --rollout'

      def run(n_steps):
        for _ in range(n_steps // args.rollout):
          env.rollout(actions[:args.rollout])
    else:
      def run(n_steps):
        for t in range(n_steps):
          env.step(actions[t % n_act])

    run(args.warmup)
    sync_all()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    sync_all()
    kernel_ms = ev0.elapsed_time(ev1) / args.steps   # HIP events on the launch stream

    # end-of-rollout summary: the only collective on the path (RCCL all-gather over xGMI)
    from bsuite_amd import distributed as bdist
    vec, names = bdist.local_summary(env)
    summary = bdist.reduce_summary(bdist.all_gather_summary(vec), names)
    if world > 1:
      t_wall = torch.tensor([wall, kernel_ms], dtype=torch.float64,
                            device=dev if dist.get_backend() == 'nccl' else 'cpu')
      dist.all_reduce(t_wall, op=dist.ReduceOp.MAX)
      wall, kernel_ms = float(t_wall[0].item()), float(t_wall[1].item())
    # state in/out counts once per T fused steps; only the small-observation families
    # fuse a rollout into one launch (their state then stays in L2 between the T steps)
    fused_T = args.rollout if (args.rollout and family not in ('deep_sea', 'catch', 'mnist')) else 1
    bytes_per_step = algorithmic_bytes_per_step(obs_numel, state_bytes / fused_T)
    if delta:
      # ACTUAL bytes of the delta mode (reported separately, never against the dense
      # contract): scalars + state in/out + paint column in/out + the 4-byte cell stores
      # (deep_sea: clear 1 + set 1; catch: up to 2 + 2).
      bytes_per_step = 13 + state_bytes + 8 + 4 * (2 if family == 'deep_sea' else 4)
    achieved = bytes_per_step * B / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(workload + ('_delta' if delta else ''), B)
    del env, actions
    torch.cuda.empty_cache()
    return dict(bsuite_id=bsuite_id, family=family, okw=okw, num_actions=num_actions, wall=wall,
                kernel_ms=kernel_ms, value=B * world * args.steps / wall, bytes_per_step=bytes_per_step,
                achieved=achieved, traffic=traffic, traffic_src=traffic_src,
                episodes_finished=summary['episodes_finished'])

  m = measure(args.workload)
  also = None
  if args.workload == 'deep_sea' and not (args.graph or args.rollout) and args.observation_mode == 'dense':
    also = measure('catch')        # the other half of BASELINE.json's metric, same K/W, same box

  # Pure-store ceiling of THIS box (one 16-B store per thread over 2 GiB, no other work): context for
  # the roofline fraction of the store-bound families.  Not part of the timed region.
  from bsuite_amd import _native
  scratch = torch.empty(1 << 31, dtype=torch.uint8, device=dev)   # 2 GiB: far beyond L2 + Infinity Cache
  nbytes = scratch.numel()
  stream_h = torch.cuda.current_stream(dev).cuda_stream
  for _ in range(3):
    _native.lib.bsx_calib_fill(scratch.data_ptr(), nbytes, 0, stream_h)
  c0 = torch.cuda.Event(enable_timing=True)
  c1 = torch.cuda.Event(enable_timing=True)
  c0.record()
  for _ in range(10):
    _native.lib.bsx_calib_fill(scratch.data_ptr(), nbytes, 0, stream_h)
  c1.record()
  torch.cuda.synchronize(dev)
  store_ceiling_gbps = nbytes * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9

  def roofline(r):
    return {'bound': 'hbm', 'achieved': r['achieved'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
            'frac': r['achieved'] / HBM_PEAK_GBPS, 'traffic': r['traffic'],
            'traffic_source': r['traffic_src'],
            'algorithmic_bytes_per_launch': r['bytes_per_step'] * args.lanes,
            'kernel_ms': r['kernel_ms'], 'box_store_ceiling_GBps': store_ceiling_gbps,
            'frac_of_box_store_ceiling': r['achieved'] / store_ceiling_gbps}

  if rank == 0:
    B = args.lanes
    line = {
        'metric': 'env-steps/sec', 'value': m['value'], 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': m['wall'] / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak',
        'vs_baseline': None, 'dtype': 'f32' if m['family'] in ('cartpole', 'mountain_car') else 'int32',
        'data': 'synthetic',
        'config': {'workload': f"{m['bsuite_id']} ({m['family']} {m['okw']}) random-action rollout, "
                               + ('dense TimeStep' if args.observation_mode == 'dense' else
                                  'DELTA observation mode (persistent buffers patched in place; not the dense contract)'),
                   'observation_mode': args.observation_mode, 'logging_wrapper': bool(args.logging),
                   'lanes_per_gpu': B, 'global_lanes': B * world, 'sharding': f'lanes x{world}',
                   'bytes_per_env_step': m['bytes_per_step']},
        'roofline': roofline(m),
        'launch': (f'hipGraph x{args.graph}' if args.graph else
                   f'rollout x{args.rollout} per call' if args.rollout else 'eager'),
        'episodes_finished': m['episodes_finished'],
    }
    if also is not None:
      line['also'] = {also['bsuite_id']: {
          'value': also['value'], 'unit': 'env-steps/s', 'ms_per_step': also['wall'] / args.steps * 1e3,
          'workload': f"{also['bsuite_id']} ({also['family']} 10x5) random-action rollout, dense TimeStep, "
                      f'{B} lanes per GPU', 'bytes_per_env_step': also['bytes_per_step'],
          'roofline': roofline(also)}}
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(m['family'], m['okw'], m['num_actions'])
    print(json.dumps(line), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
