"""bench.py — env-steps/sec of the batched bsuite step() path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload deep_sea|catch|...|sweep] [--lanes B] [--weak]

One "step" = one env.step(actions) call on every lane of the batch (auto-reset calls included,
they are real API calls: bsuite/environments/base.py:61-62).  Default workload is BASELINE.json
configs[1]: deep_sea size=30 (bsuite_id deep_sea/10), 2^20 lanes, uniform random actions
pre-generated on the device (the batched analogue of bsuite/baselines/random/agent.py:35-37).
Every TimeStep field is materialised in HBM on every step (dense contract).  Before the warm-up the
lanes are put at staggered episode phases, so that every timed call carries the steady-state mix of
FIRST / MID / LAST lanes whatever K is.

The default line also carries the other BASELINE configs as sub-records under "also" — catch/0
(configs[2]) first, then cartpole/0 and mountain_car/0 (configs[3]; eager step(), a HIP graph of 16
step() launches, the fused rollout(T=16)) and the 468-id sweep (configs[4]) — in a compact form: 5
significant digits, short codes instead of prose (legend: DESIGN.md §6), so that the whole line stays
well under 8000 characters.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); lanes shard with no
data-path collective; the only collective is the end-of-rollout all-gather of per-rank summaries.
Either launch the ranks yourself (`python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N`), or run `python bench.py --gpus N` and this script spawns them itself (the reference's
own parallel entry is self-launching too: bsuite/baselines/utils/pool.py:28-54).  `--lanes` is the
GLOBAL batch (BASELINE.json: "batch=2^20 ... at 1, 2, 4 and 8 GPUs"): with N > 1 the main
record is STRONG scaling — 2^20 / N lanes per GPU — and the weak-scaling figure (2^20 lanes per GPU) is
information under "also" (`--weak` makes it the main record).
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

# workload -> (bsuite_id, oracle family, oracle kwargs, obs_numel, state bytes in+out per lane,
#              episode period in calls used to stagger the lanes' phases)
WORKLOADS = {
    'deep_sea': ('deep_sea/10', 'deep_sea', dict(size=30, mapping_seed=42), 900, 8, 31),
    # deep_sea/9: rows of 784 floats — the geometry control of the mnist bandit's store stream (VERDICT r05 next #1a)
    'deep_sea_28': ('deep_sea/9', 'deep_sea', dict(size=28, mapping_seed=42), 784, 8, 29),
    'catch': ('catch/0', 'catch', dict(), 50, 8, 10),
    'catch_noise': ('catch_noise/0', 'catch', dict(), 50, 8, 10),     # RewardNoise(0.1): the non-lean kernel instantiations
    'cartpole': ('cartpole/0', 'cartpole', dict(), 6, 48, 128),
    'mountain_car': ('mountain_car/0', 'mountain_car', dict(), 3, 24, 1001),
    'cartpole_noise': ('cartpole_noise/0', 'cartpole', dict(), 6, 48, 128),   # RewardNoise(0.1): the wrapped rollout kernels
    'bandit': ('bandit/0', 'bandit', dict(mapping_seed=0), 1, 8, 2),
    'memory_len': ('memory_len/10', 'memory_chain', dict(memory_length=12, num_bits=1), 3, 24, 14),
    'umbrella_length': ('umbrella_length/10', 'umbrella_chain', dict(chain_length=12, n_distractor=20), 23, 8, 13),
    'umbrella_distract': ('umbrella_distract/22', 'umbrella_chain', dict(chain_length=20, n_distractor=100), 103, 8, 21),
    'memory_size': ('memory_size/16', 'memory_chain', dict(memory_length=2, num_bits=40), 42, 24, 4),
    'discounting_chain': ('discounting_chain/0', 'discounting_chain', dict(mapping_seed=0), 2, 8, 101),
    # the MNIST files cannot be fetched here: tests/golden/mnist_synthetic_dataset.npz (same idx wire format)
    'mnist': ('mnist/0', 'mnist', dict(), 784, 8, 2),
}


def _synthetic_mnist():
  import numpy as np
  if os.environ.get('BSX_BENCH_MNIST_DIR'):     # (information only: idx files of another size, e.g. the real 47 MB table)
    from bsuite_amd.utils import datasets
    (x, y), _ = datasets.load_mnist(os.environ['BSX_BENCH_MNIST_DIR'])
    return x, y
  d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
  return d['images_u8'].view(np.int8), d['labels']


def algorithmic_bytes_per_step(obs_numel, state_bytes):
  # action i32 + reward f32 + discount f32 + step_type i8 + obs f32 + state in/out
  return 4 + 4 + 4 + 1 + 4 * obs_numel + state_bytes


def _latest_profile(name):
  import glob
  hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', name)))
  if not hits:
    return None, None
  with open(hits[-1]) as f:
    return json.load(f), os.path.relpath(hits[-1], ROOT)


def pmc_traffic(workload, lanes, mode='eager', chunk=0, logging=False):
  """HBM bytes per env step of the whole batch from the committed rocprofv3 PMC passes (profiles/rNN/<w>[_logging][_<mode>]
  _pmc_traffic.json: WRITE_SIZE and FETCH_SIZE collected in separate runs over the timed launches, gfx950 corrections
  applied there; a fused rollout launch's bytes / T, a multi-launch rollout's summed over its launches: tools/pmc.py).
  A HIP-graph replay issues the eager launches: it cites their pass."""
  key = workload + ('_logging' if logging else '') + ('' if mode in ('eager', 'graph') else f'_{mode}{chunk}')
  d, src = _latest_profile(f'{key}_pmc_traffic.json')
  if d is None or lanes != d.get('lanes', 1 << 20) or not d['per_launch'].get('fetch_bytes_x2'):
    return None, None                      # (a pass that recorded no FETCH bytes is not evidence)
  return d['per_launch']['hbm_bytes'], src


# VALU issue peak of the chip: 256 CUs x 4 SIMDs, one wave64 instruction per 2 cycles at 2.4 GHz
# (/opt/skills/guides/MI355X_MICROARCH.md: "issues each VALU instruction over 2 cycles"), in 10^9 wave-instr/s
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2


def pmc_valu(workload, mode, lanes):
  """VALU wave-instructions per launch (and the measured VALU-busy share of the SIMD cycles) of a fused rollout
  kernel, from the committed SQ counter pass (profiles/rNN/<w>_<mode>_pmc_sq.json, tools/pmc.py)."""
  d, src = _latest_profile(f'{workload}_{mode}_pmc_sq.json')
  if d is None or lanes != d.get('lanes', 1 << 20):
    return None
  return dict(insts=d['per_launch']['SQ_INSTS_VALU'], busy=d['per_launch'].get('valu_busy_frac'), src=src)


def sig(x, n=5):
  """The JSON line in n significant digits (the driver keeps an 8000-character tail of it)."""
  if isinstance(x, float):
    return float(f'{x:.{n}g}') if x == x and abs(x) != float('inf') else None
  if isinstance(x, dict):
    return {k: sig(v, n) for k, v in x.items()}
  if isinstance(x, (list, tuple)):
    return [sig(v, n) for v in x]
  return x


# ------------------------------------------------------------------------------------ CPU baselines
def _reference_loop(bsuite_id, seconds, seed=0):
  """The UNMODIFIED reference stepped on this box: `bsuite.load_from_id(id)` driven by the loop of
  bsuite/baselines/experiment.py:43-57 with the random agent of bsuite/baselines/random/agent.py:35-37 inlined.
  env-steps count every environment call, reset() included (base.py:54-65).  The package is imported from
  /root/reference where that exists, else from the byte-code `__graft_entry__.build()` staged under oracle/_ref
  (oracle/stage_reference.py) — it is the checker's copy, never the product's."""
  import numpy as np
  from oracle import replay
  bs = replay.import_reference()
  env = bs.load_from_id(bsuite_id)
  num_actions = env.action_spec().num_values
  rng = np.random.RandomState(seed)
  calls = 0
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < seconds:
    timestep = env.reset()
    calls += 1
    while not timestep.last():
      timestep = env.step(rng.randint(num_actions))
      calls += 1
  return calls, time.perf_counter() - t0


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


def _host_cores():
  return len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)


def _reference_pool(bsuite_id, seconds, n_procs):
  """Runs in a fresh helper interpreter (no torch, no HIP): imports the reference once, then forks one worker
  PROCESS per core — the shape of the reference's own fan-out (bsuite/baselines/utils/pool.py:35,48; threads would
  serialise on the interpreter lock) — and prints one "<env-steps> <seconds>" pair per worker."""
  import multiprocessing
  from oracle import replay
  replay.import_reference()                       # the workers inherit the imported package: no 256 cold imports
  with multiprocessing.get_context('fork').Pool(n_procs) as pool:
    res = pool.starmap(_reference_loop, [(bsuite_id, seconds, 1 + j) for j in range(n_procs)])
  for calls, dt in res:
    print(calls, dt)


def cpu_port_baseline(family, kwargs, num_actions, budget_s=2.5):
  """The oracle (C restatement of the reference's numpy step) on ONE host core of THIS box, a bounded sample:
  context for the reference number below, neither the target nor the reference."""
  lanes = 4096
  steps, dt = _oracle_loop(family, kwargs, num_actions, lanes, 0, budget_s)
  return dict(value=steps / dt, cores=1, kind='port', sample=f'oracle.c, {lanes} lanes x {steps // lanes} calls, {dt:.1f} s')


def cpu_baseline(bsuite_id, family, kwargs, num_actions, single_s=8.0, all_s=5.0):
  """`cpu_baseline` of the JSON line: the reference's own numpy path timed LIVE on this box's host cores in this
  very run (`kind: "reference"`; BASELINE.json north_star) — one core, then one process per core — with the C
  port beside it.  Falls back to the port alone (and says so) where no reference can be imported."""
  from oracle import replay
  port = cpu_port_baseline(family, kwargs, num_actions)
  origin = replay.reference_origin()
  if origin is None:
    port.update(unit='env-steps/s', note='no reference on this box (oracle/_ref not staged): C port only')
    return port
  calls, dt = _reference_loop(bsuite_id, single_s)
  cores = _host_cores()
  out = dict(value=calls / dt, unit='env-steps/s', cores=1, kind='reference',
             sample=f"unmodified bsuite.load_from_id('{bsuite_id}') + random agent, {calls} reset()/step() calls in "
                    f"{dt:.1f} s on one core of this box ({origin}: {'/root/reference' if origin == 'source' else 'oracle/_ref byte-code'})")
  if cores > 1 and all_s > 0:
    import subprocess
    code = ('import sys; sys.path.insert(0, %r); import bench; '
            'bench._reference_pool(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]))' % ROOT)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, '-c', code, bsuite_id, str(all_s), str(cores)], stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL, text=True, check=False)
    wall = time.perf_counter() - t0
    res = [tuple(float(x) for x in l.split()) for l in p.stdout.splitlines() if len(l.split()) == 2]
    if res:
      out['all_cores'] = dict(value=sum(r[0] for r in res) / max(r[1] for r in res), cores=len(res),
                              sample=f'{len(res)} processes x {all_s:.0f} s (pool.py:35,48 shape), {wall:.1f} s wall incl. start-up')
  out['port'] = port
  return out


# ------------------------------------------------------------------------------------ helpers
def _raw(env):
  return env.raw_env if hasattr(env, 'raw_env') else env


def stagger_phases(env, actions, period):
  """Puts lane i at episode phase i % period: steps the batch `period` times and keeps, for each
  lane, the state it had after (i % period) + 1 calls (per-lane columns of state_dict(); the draw
  stream is keyed by (lane, call index), so any such mix is a valid engine state).  Afterwards every
  call sees the steady-state mix of FIRST / MID / LAST lanes instead of all lanes terminating on
  the same call."""
  import torch
  raw = _raw(env)
  B = raw.batch_size
  phase = (torch.arange(B, device=raw.device) + raw.lane_offset) % period     # keyed by GLOBAL lane id
  final = None
  for k in range(period):
    env.step(actions[k % actions.shape[0]])
    sd = raw.state_dict()
    if final is None:
      final = sd
      continue
    take = phase == k
    for key, val in sd.items():
      if torch.is_tensor(val) and val.dim() >= 1 and val.shape[-1] == B and key != '__counters':
        final[key] = torch.where(take, val, final[key])
      else:
        final[key] = val
  raw.load_state_dict(final)


def synthetic_actions(torch, num_actions, n_steps, lane0, lanes, dev, salt=0):
  """int32 [n_steps, lanes] uniform actions as a pure function of (global lane id, t): an integer hash
  evaluated on the device, so any sharding of the lanes feeds every lane the same action sequence
  (the batched, shardable analogue of bsuite/baselines/random/agent.py:35-37)."""
  lane = torch.arange(lane0, lane0 + lanes, dtype=torch.int64, device=dev)[None, :]
  t = torch.arange(n_steps, dtype=torch.int64, device=dev)[:, None]
  x = (lane * 2654435761 + t * 40503 + (salt * 7919 + 12345)) & 0xFFFFFFFF
  x = ((x ^ (x >> 15)) * 2246822519) & 0xFFFFFFFF
  x = ((x ^ (x >> 13)) * 3266489917) & 0xFFFFFFFF
  x = x ^ (x >> 16)
  return ((x * num_actions) >> 32).to(torch.int32)


def _free_port():
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def self_launch(args):
  """`python bench.py --gpus N` with no WORLD_SIZE in the environment: spawn the N ranks ourselves
  (one per device, torch.distributed.run on 127.0.0.1) and pass their output through."""
  import subprocess
  import torch
  single = bool(os.environ.get('BSX_BENCH_SINGLE_DEVICE'))
  have = torch.cuda.device_count()
  if have < args.gpus and not single:
    raise SystemExit(f'--gpus {args.gpus} but only {have} HIP device(s) visible '
                     '(BSX_BENCH_SINGLE_DEVICE=1 BSX_BENCH_BACKEND=gloo runs all ranks on cuda:0 for testing)')
  env = dict(os.environ)
  env.setdefault('OMP_NUM_THREADS', '4')
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------ one rank
class Rank:
  """Everything one process (one GPU) measures."""

  def __init__(self, args):
    import torch
    import torch.distributed as dist
    import bsuite_amd
    self.torch, self.dist, self.bsuite_amd, self.args = torch, dist, bsuite_amd, args
    self.world = int(os.environ.get('WORLD_SIZE', '1'))
    self.rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Test hooks (single-GPU boxes): BSX_BENCH_BACKEND=gloo + BSX_BENCH_SINGLE_DEVICE=1 run all ranks
    # on cuda:0 so the multi-rank control flow can be exercised without a multi-GPU node.
    backend = os.environ.get('BSX_BENCH_BACKEND', 'nccl')
    if os.environ.get('BSX_BENCH_SINGLE_DEVICE'):
      local_rank = 0
    # BSX_BENCH_FORCE_PG=1: create the process group even for one rank, so that the RCCL code path (init,
    # barrier, all-reduce, all-gather on device tensors) is exercised on a single-GPU box.
    self.collective = self.world > 1 or bool(os.environ.get('BSX_BENCH_FORCE_PG'))
    if self.collective:
      os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
      if 'MASTER_PORT' not in os.environ:
        os.environ['MASTER_PORT'] = str(_free_port())
      dist.init_process_group(backend, rank=self.rank, world_size=self.world)
    if args.gpus != self.world:
      raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={self.world}')
    torch.cuda.set_device(local_rank)
    self.dev = torch.device('cuda', local_rank)
    self._ceiling = None
    self.pinned_cores = self.pin_to_local_cores(local_rank) if self.world > 1 else None
    self.warm_runtime()

  def pin_to_local_cores(self, local_rank):
    """One rank per GPU: bind this process to host cores of its GPU's NUMA node, the node's cores dealt out evenly among
    the ranks whose GPUs share it (the eager step of the small families costs the host about what the
    kernel costs the GPU, and eight ranks on one node would otherwise migrate over — and contend for — the same cores;
    the reference pins nothing, bsuite/baselines/utils/pool.py:28-54).  Best effort: returns the number of cores bound
    to, or None where the topology cannot be read (BSX_BENCH_NO_PIN=1 switches it off)."""
    if os.environ.get('BSX_BENCH_NO_PIN') or not hasattr(os, 'sched_setaffinity'):
      return None
    try:
      torch = self.torch

      def node_of(idx):
        pr = torch.cuda.get_device_properties(idx)
        bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
        with open(f'/sys/bus/pci/devices/{bdf}/numa_node') as f:
          return int(f.read())

      single = bool(os.environ.get('BSX_BENCH_SINGLE_DEVICE'))
      n_local = int(os.environ.get('LOCAL_WORLD_SIZE', self.world))
      me = int(os.environ.get('LOCAL_RANK', '0'))
      nodes = [node_of(0 if single else r) for r in range(n_local)]
      node = nodes[me]
      if node < 0:
        return None
      with open(f'/sys/devices/system/node/node{node}/cpulist') as f:
        cpus = []
        for part in f.read().strip().split(','):
          lo, _, hi = part.partition('-')
          cpus += list(range(int(lo), int(hi or lo) + 1))
      cpus = sorted(set(cpus) & os.sched_getaffinity(0))
      sharers = [r for r in range(n_local) if nodes[r] == node]
      share = cpus[sharers.index(me)::len(sharers)]              # interleaved: SMT siblings stay with one rank or the other evenly
      if len(share) < 2:
        return None
      os.sched_setaffinity(0, share)
      return len(share)
    except Exception:  # pylint: disable=broad-except
      return None

  def warm_runtime(self, launches=2048):
    """The first time a process has more than ~256 launches in flight on a stream, the HIP runtime stalls once
    for ~37 ms (it grows a pool; tools/step_loop_gpu_vs_host.py shows the one-off).  The host enqueues a step
    in ~5.5 us, the kernels run 6.6-18 us, so a timed region of a few hundred steps would otherwise pay that
    stall once, at a random step.  Take it here, before anything is timed."""
    torch = self.torch
    from bsuite_amd import _native
    with torch.cuda.device(self.dev):
      scratch = torch.zeros(1, dtype=torch.int64, device=self.dev)
      stream_h = torch.cuda.current_stream(self.dev).cuda_stream
      for _ in range(launches):
        _native.lib.bsx_counter_add(scratch.data_ptr(), 0, stream_h)
      torch.cuda.synchronize(self.dev)

  def sync_all(self):
    self.torch.cuda.synchronize(self.dev)
    if self.collective:
      self.dist.barrier()
      self.torch.cuda.synchronize(self.dev)

  def reduce_times(self, *vals):
    """MAX over ranks of host-side floats."""
    if not self.collective:
      return vals
    t = self.torch.tensor(list(vals), dtype=self.torch.float64,
                          device=self.dev if self.dist.get_backend() == 'nccl' else 'cpu')
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
    return tuple(float(x) for x in t.tolist())

  def timed(self, run, steps, warmup):
    """W untimed + exactly K timed steps, bracketed by barrier + synchronize on both sides; returns
    (wall seconds, ms per step by HIP events on the launch stream), MAX over ranks."""
    torch = self.torch
    run(warmup)
    self.sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(steps)
    ev1.record()
    torch.cuda.synchronize(self.dev)
    wall = time.perf_counter() - t0
    self.sync_all()
    return self.reduce_times(wall, ev0.elapsed_time(ev1) / steps)

  # -------------------------------------------------------------------------------------------
  def measure(self, workload, lanes, steps, warmup, mode='eager', chunk=0, observation_mode='dense',
              logging=False, stagger=True):
    """Times `steps` step() calls of `workload` on `lanes` lanes per rank.  mode: 'eager' (one
    entry-point call per step), 'graph' (HIP graph of `chunk` step() launches), 'rollout' (fused
    rollout(actions[chunk, B]))."""
    torch, args = self.torch, self.args
    bsuite_id, family, okw, obs_numel, state_bytes, period = WORKLOADS[workload]
    B = lanes
    delta = observation_mode == 'delta'
    if delta and family not in ('deep_sea', 'catch'):
      raise SystemExit('--observation-mode delta exists for deep_sea and catch only')
    extra = {}
    if family == 'mnist':
      extra['images'], extra['labels'] = _synthetic_mnist()
    env = self.bsuite_amd.load_from_id(bsuite_id, batch=B, device=self.dev, seed=42, lane_offset=self.rank * B,
                                       num_buffers=2, device_step_counter=(mode == 'graph'),
                                       observation_mode=observation_mode, **extra)
    if logging:
      # the Logging wrapper's per-lane bookkeeping + log-spaced snapshot rows, fused
      # into the same kernels (no logger object: rows stay in the device buffer)
      from bsuite_amd.utils import wrappers as _wrappers
      env = _wrappers.Logging(env, None)
    num_actions = env.action_spec().num_values
    n_act = max(32, chunk)
    actions = synthetic_actions(torch, num_actions, n_act, self.rank * B, B, self.dev)
    if stagger and not args.no_stagger and not delta and not logging:
      stagger_phases(env, actions, period)
    else:
      env.step(actions[0])

    if mode == 'graph':
      assert steps % chunk == 0 and warmup % chunk == 0, '--steps/--warmup must be multiples of --graph'
      side = torch.cuda.Stream(device=self.dev)
      side.wait_stream(torch.cuda.current_stream(self.dev))
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
          with _raw(env).step_counter_deferred():        # one call-counter bump per replay, not per step
            for t in range(chunk):
              env.step(actions[t])
      torch.cuda.current_stream(self.dev).wait_stream(side)

      def run(n_steps):
        for _ in range(n_steps // chunk):
          graph.replay()
    elif mode == 'rollout':
      assert steps % chunk == 0 and warmup % chunk == 0, '--steps/--warmup must be multiples of --rollout'

      def run(n_steps):
        for _ in range(n_steps // chunk):
          env.rollout(actions[:chunk])
    else:
      def run(n_steps):
        for t in range(n_steps):
          env.step(actions[t % n_act])

    run(warmup)                                   # (untimed; the LAST / FIRST counts below are those of the timed steps only)
    torch.cuda.synchronize(self.dev)
    before = _raw(env).episode_counters().clone()
    wall, kernel_ms = self.timed(run, steps, 0)
    ended = (_raw(env).episode_counters() - before).to(torch.float64)

    # end-of-rollout summary: the only collective on the path (RCCL all-gather over xGMI)
    from bsuite_amd import distributed as bdist
    vec, names = bdist.local_summary(env)
    vec = torch.cat([vec, ended])
    summary = bdist.reduce_summary(bdist.all_gather_summary(vec), names + ('timed_last', 'timed_first'))
    # state in/out counts once per T fused steps; only the small-observation families
    # fuse a rollout into one launch (their state then stays in L2 between the T steps)
    fused_T = chunk if (mode == 'rollout' and family not in ('deep_sea', 'catch', 'mnist')) else 1
    bytes_per_step = algorithmic_bytes_per_step(obs_numel, state_bytes / fused_T)
    if delta:
      # ACTUAL bytes of the delta mode (reported separately, never against the dense
      # contract): scalars + state in/out + paint column in/out + the 4-byte cell stores
      # (deep_sea: clear 1 + set 1; catch: up to 2 + 2).
      bytes_per_step = 13 + state_bytes + 8 + 4 * (2 if family == 'deep_sea' else 4)
    achieved = bytes_per_step * B / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(workload + ('_delta' if delta else ''), B, mode, chunk, logging)
    del env, actions
    torch.cuda.empty_cache()
    calls = steps * B * self.world
    return dict(workload=workload, bsuite_id=bsuite_id, family=family, okw=okw, num_actions=num_actions, lanes=B,
                steps=steps, wall=wall, kernel_ms=kernel_ms, value=calls / wall, bytes_per_step=bytes_per_step,
                achieved=achieved, traffic=traffic, traffic_src=traffic_src, mode=mode, chunk=chunk,
                episodes_finished=summary['episodes_finished'],
                info_sums={k: v for k, v in summary.items()
                           if k not in ('lanes', 'episodes_finished', 'episodes_started', 'timed_last', 'timed_first')},
                timed_mix=dict(last=summary['timed_last'] / calls, first=summary['timed_first'] / calls))

  def measure_sustained(self, workload, lanes, seconds, window_s=0.5):
    """Eager step() calls for at least `seconds` of GPU time, a HIP event on the launch stream every ~window_s:
    ms per step of every window -> min / median / max.  Says whether the 20-step figure of the headline (12 ms of
    GPU time) holds once clocks, power state and the memory system have seen seconds of the same work; it also puts
    seconds of GPU activity into the run for whoever samples it from outside."""
    torch = self.torch
    bsuite_id, family, okw, obs_numel, state_bytes, period = WORKLOADS[workload]
    env = self.bsuite_amd.load_from_id(bsuite_id, batch=lanes, device=self.dev, seed=42, lane_offset=self.rank * lanes, num_buffers=2)
    num_actions = env.action_spec().num_values
    actions = synthetic_actions(torch, num_actions, 32, self.rank * lanes, lanes, self.dev)
    stagger_phases(env, actions, period)
    for t in range(20):
      env.step(actions[t % 32])
    torch.cuda.synchronize(self.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(20):
      env.step(actions[t % 32])
    e1.record()
    torch.cuda.synchronize(self.dev)
    per_window = max(20, int(window_s * 1e3 / (e0.elapsed_time(e1) / 20)))
    n_windows = max(4, int(round(seconds / window_s)))
    events = [torch.cuda.Event(enable_timing=True) for _ in range(n_windows + 1)]
    t0 = time.perf_counter()
    events[0].record()
    t = 0
    for w in range(n_windows):
      for _ in range(per_window):
        env.step(actions[t & 31])
        t += 1
      events[w + 1].record()
      if w >= 2:
        events[w - 1].synchronize()            # the host stays at most two windows ahead of the GPU
    torch.cuda.synchronize(self.dev)
    wall = time.perf_counter() - t0
    ms = sorted(events[w].elapsed_time(events[w + 1]) / per_window for w in range(n_windows))
    med = ms[len(ms) // 2]
    bytes_per_step = algorithmic_bytes_per_step(obs_numel, state_bytes)
    del env, actions
    torch.cuda.empty_cache()
    return {'steps': n_windows * per_window, 'seconds': wall, 'windows': n_windows, 'steps_per_window': per_window,
            'ms_per_step': {'min': ms[0], 'median': med, 'max': ms[-1]}, 'value': lanes * self.world / (med * 1e-3),
            'roofline_frac_median': bytes_per_step * lanes / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'mode': 'e', 'lanes_per_gpu': lanes}

  def store_ceiling(self):
    """Pure-store ceiling of THIS box (one 16-B store per thread over 2 GiB, no other work): context for
    the roofline fraction of the store-bound families.  Not part of any timed region."""
    if self._ceiling is None:
      torch = self.torch
      from bsuite_amd import _native
      scratch = torch.empty(1 << 31, dtype=torch.uint8, device=self.dev)   # 2 GiB: far beyond L2 + Infinity Cache
      nbytes = scratch.numel()
      stream_h = torch.cuda.current_stream(self.dev).cuda_stream
      for _ in range(3):
        _native.lib.bsx_calib_fill(scratch.data_ptr(), nbytes, 0, stream_h)
      c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      c0.record()
      for _ in range(10):
        _native.lib.bsx_calib_fill(scratch.data_ptr(), nbytes, 0, stream_h)
      c1.record()
      torch.cuda.synchronize(self.dev)
      self._ceiling = nbytes * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
      del scratch
      torch.cuda.empty_cache()
    return self._ceiling

  def copy_ceiling(self):
    """Copy ceiling of THIS box for the small-observation families' access mix (a third read, two thirds written:
    bsx_calib_copy, 1 GiB in, 2 GiB out, one 16-byte load + two 16-byte stores per thread): (R + W) bytes per second.
    A kernel that mixes reads into its writes runs against this line, not against the pure-fill rate."""
    if getattr(self, '_copy_ceiling', None) is None:
      torch = self.torch
      from bsuite_amd import _native
      n = 1 << 30
      src = torch.empty(n, dtype=torch.uint8, device=self.dev)
      dst = torch.empty(2 * n, dtype=torch.uint8, device=self.dev)
      stream_h = torch.cuda.current_stream(self.dev).cuda_stream
      for _ in range(3):
        _native.lib.bsx_calib_copy(dst.data_ptr(), src.data_ptr(), n, 2, stream_h)
      c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      c0.record()
      for _ in range(10):
        _native.lib.bsx_calib_copy(dst.data_ptr(), src.data_ptr(), n, 2, stream_h)
      c1.record()
      torch.cuda.synchronize(self.dev)
      self._copy_ceiling = 3 * n * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
      del src, dst
      torch.cuda.empty_cache()
    return self._copy_ceiling

  def host_step_us(self, calls=4000):
    """What ONE env.step() costs the host on this box (Python + ctypes + hipLaunchKernel), in microseconds: eager calls
    of a 256-lane bandit, whose kernel the GPU finishes faster than any host can enqueue it.  An eager record whose
    kernel time is not well above this figure is bound by the host, not by the GPU (the same
    mountain_car step measured 9.9 us on one box and 11.6 on another) — such records carry `host_bound`."""
    if getattr(self, '_host_step_us', None) is None:
      torch = self.torch
      env = self.bsuite_amd.load_from_id('bandit/0', batch=256, device=self.dev, seed=1, num_buffers=2)
      a = torch.zeros(256, dtype=torch.int32, device=self.dev)
      for _ in range(200):
        env.step(a)
      torch.cuda.synchronize(self.dev)
      t0 = time.perf_counter()
      for _ in range(calls):
        env.step(a)
      dt = time.perf_counter() - t0
      torch.cuda.synchronize(self.dev)
      self._host_step_us = dt / calls * 1e6
    return self._host_step_us

  def roofline(self, r, full=False):
    """HBM roofline of a record: algorithmic bytes per launch / launch time by HIP events.  Fused
    rollouts of the physics families also carry the VALU-issue roofline (`valu`: wave-instructions of the committed
    SQ pass / launch time against one wave64 instruction per 2 cycles and SIMD): their state stays in registers for
    T steps, and neither ceiling is close — the record's `bound` is the nearer one (DESIGN §3.3 has the ablations:
    cartpole 7.1 us of arithmetic + 5.1 us of stores at the fill rate, 10.6-11 us measured)."""
    out = {'bound': 'hbm', 'achieved': r['achieved'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
           'frac': r['achieved'] / HBM_PEAK_GBPS, 'traffic': r['traffic']}
    if full:
      ceiling = self.store_ceiling()
      out.update(traffic_src=r['traffic_src'], alg_bytes=int(round(r['bytes_per_step'] * r['lanes'])), kernel_ms=r['kernel_ms'],
                 box_fill_GBps=ceiling, frac_of_box_fill=r['achieved'] / ceiling)
    if r['mode'] == 'rollout' and r['family'] in ('cartpole', 'mountain_car'):
      v = pmc_valu(r['workload'], f"rollout{r['chunk']}", r['lanes'])
      if v is not None:
        ginstr = v['insts'] / (r['kernel_ms'] * r['chunk'] * 1e-3) / 1e9      # the launch runs `chunk` steps
        valu = {'achieved': ginstr, 'peak': VALU_PEAK_GINSTR, 'unit': 'Gwave-instr/s', 'frac': ginstr / VALU_PEAK_GINSTR,
                'valu_busy': v['busy'], 'src': v['src']}
        if valu['frac'] > out['frac']:                                        # the nearer ceiling names the bound
          out = dict(valu, bound='valu', hbm={'achieved': out['achieved'], 'frac': out['frac']})
        else:
          out['valu'] = valu
    return out

  def sub_record(self, r):
    """Compact sub-record.  `mode`: e = eager step() per call; gT = hipGraph of T step() launches; rT = rollout(T)
    per call — deep_sea / catch: software-pipelined, T+1 launches (bsx_call_t.state_alt); the other families: one
    fused T-step kernel."""
    mode = 'e' if r['mode'] == 'eager' else f"g{r['chunk']}" if r['mode'] == 'graph' else f"r{r['chunk']}"
    roof = self.roofline(r)
    # (compact: `steps`, `lanes_per_gpu` and the roofline's peak / unit are the same for every sub-record of a line and
    # stand once in `also_common`; a record that differs carries its own)
    for k in ('peak', 'unit'):
      roof.pop(k, None)
    out = {'value': r['value'], 'ms_per_step': r['wall'] / r['steps'] * 1e3, 'mode': mode,
           'bytes_per_env_step': r['bytes_per_step'], 'roofline': roof}
    common = getattr(self, '_also_common', None)
    if common is None or r['steps'] != common['steps']:
      out['steps'] = r['steps']
    if common is None or r['lanes'] != common['lanes_per_gpu']:
      out['lanes_per_gpu'] = r['lanes']
    if r['family'] == 'bandit' and r['mode'] == 'eager':
      # SURVEY §8(d) leaves the info accumulators out ("touched only on LAST steps: amortised < 1 B") — not so for a ONE-step
      # episode: every second call of every lane is LAST, and the order-dependent f64 total_regret column (bandit.py:62) is a
      # whole cache line read and written per 16 lanes whichever half of them updates: + 8 B in + 8 B out per env step.  The
      # line's `frac` stays on the survey's 25 B; this is what the kernel really has to move (PMC: profiles/*/bandit_pmc_traffic.json)
      out['bytes_actual'] = r['bytes_per_step'] + 16
      roof['frac_actual_bytes'] = roof['frac'] * (r['bytes_per_step'] + 16) / r['bytes_per_step']
    if r['mode'] == 'eager' and r['family'] not in ('deep_sea', 'catch', 'mnist') and self.world == 1:
      # the small-observation families read a third of what they write: their line is this box's COPY rate
      roof['frac_of_box_copy'] = r['achieved'] / self.copy_ceiling()
      if r['kernel_ms'] * 1e3 < 1.25 * self.host_step_us():
        out['host_bound'] = True           # one launch per call: the host enqueues about as slowly as the kernel runs
    return out

  # -------------------------------------------------------------------------------------------
  def measure_sweep(self, lanes, steps, warmup, ring=16):
    """BASELINE config 5: all 468 bsuite_ids as lane segments (`lanes` in total, split evenly per id, whole
    segments bin-packed over the ranks), the whole sweep as ONE launch group.  Random actions over time: every
    segment reads row (sweep step mod `ring`) of a pre-generated [ring, lanes] action ring on the device
    (bsx_call_t.action_ring; SURVEY §8d actions [T,B]).  Two schedules are timed on the same batch:
      closed-loop (the record's `value`): two launches per sweep step — phase 0 advances every lane and bumps the
        call counter, phase 1 is the observation store stream; the TimeSteps of step s are complete before the
        actions of step s+1 are needed;
      open-loop (`pipelined`): one launch per step — the store stream of step s beside the lane advance of step
        s+1 — valid because the actions do not depend on the observations; not reachable by a closed-loop agent."""
    import tempfile
    import numpy as np
    torch = self.torch
    from bsuite_amd import sweep_batch as sb
    from bsuite_amd.utils import datasets
    from bsuite_amd import distributed as bdist
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
    tmp = tempfile.mkdtemp(prefix='bsx_mnist_')
    datasets.write_idx_files(tmp, d['images_u8'], d['labels'])      # synthetic stand-in (no network)
    mn = dict(data_dir=tmp)
    batch = sb.SweepBatch(None, lanes, device=self.dev, seed=42, rank=self.rank, world_size=self.world,
                          env_kwargs=dict(mnist=mn, mnist_noise=mn, mnist_scale=mn))
    acts = batch.random_actions(seed=1, ring=ring)   # keyed by segment: independent of the rank assignment
    local_bytes = float(sum(l * sb.bytes_per_step(int(np.prod(e.observation_spec().shape)))
                            for e, (_, _, l) in zip(batch.envs, batch.segments)))
    timed = {}
    only = getattr(self.args, 'sweep_schedule', 'all')            # (profiling passes time ONE schedule: its kernels' dispatches then are its own)
    for name, pipelined in (('closed', False), ('split', False), ('pipelined', True)):
      if only not in ('all', name):
        continue
      batch.prepare_groups(acts, pipelined=pipelined, rows_in_stream=(None if self.args.row_path == 'auto' else self.args.row_path == 'on'),
                           split=(name == 'split'))

      def run(n):
        for _ in range(n):
          batch.step_grouped()

      timed[name] = self.timed(run, steps, warmup)
      batch.release_groups()
    # the only collective: all-gather of the per-rank summaries (here: bytes + lanes + episode counters)
    summ = batch.summary()
    vec = torch.tensor([local_bytes, float(batch.lanes()), float(len(batch.envs)),
                        sum(v['episodes_finished'] for v in summ.values())], dtype=torch.float64, device=self.dev)
    g = bdist.all_gather_summary(vec)
    total_lanes, max_rank_bytes = float(g[:, 1].sum()), float(g[:, 0].max())

    def roof(step_ms, traffic=None):
      achieved = max_rank_bytes / (step_ms * 1e-3) / 1e9          # the busiest rank's stream
      return {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
              'traffic': traffic}

    tr = {k: (pmc_traffic(k, int(total_lanes))[0] if self.world == 1 else None) for k in ('sweep_closed', 'sweep_split', 'sweep_pipelined')}
    # `value`: the closed-loop schedule step_grouped() runs by default (sweep_batch.DEFAULT_SPLIT); the other closed-loop
    # cut of the same two launches beside it
    main, other = ('split', 'closed') if sb.DEFAULT_SPLIT else ('closed', 'split')
    modes = {'closed': 'closed-loop, 2 launches/step: phase 0 | store stream',
             'split': 'closed-loop, 2 launches/step: advance of the stream families | store stream + small families'}
    if only != 'all':
      wall, step_ms = timed[only]
      rec = {'value': total_lanes * steps / wall, 'ms_per_step': wall / steps * 1e3, 'steps': steps,
             'mode': modes.get(only, 'open-loop, 1 launch/step'), 'global_lanes': int(total_lanes),
             'segments_per_rank': [int(x) for x in g[:, 2].tolist()], 'action_ring': ring, 'alg_bytes_busiest_rank': max_rank_bytes,
             'roofline': roof(step_ms, tr['sweep_' + only]), 'pipelined': {'open_loop': True}, 'episodes_finished': float(g[:, 3].sum())}
      del batch, acts
      torch.cuda.empty_cache()
      return rec
    (wall, step_ms), (wall_o, step_ms_o), (wall_p, step_ms_p) = timed[main], timed[other], timed['pipelined']
    rec = {'value': total_lanes * steps / wall, 'ms_per_step': wall / steps * 1e3, 'steps': steps, 'mode': modes[main],
           'global_lanes': int(total_lanes), 'segments_per_rank': [int(x) for x in g[:, 2].tolist()], 'action_ring': ring,
           'alg_bytes_busiest_rank': max_rank_bytes, 'roofline': roof(step_ms, tr['sweep_' + main]),
           other: {'ms_per_step': wall_o / steps * 1e3, 'frac': roof(step_ms_o)['frac']},
           'pipelined': {'value': total_lanes * steps / wall_p, 'ms_per_step': wall_p / steps * 1e3, 'open_loop': True,
                         'mode': '1 launch/step', 'roofline': roof(step_ms_p, tr['sweep_pipelined'])},
           'episodes_finished': float(g[:, 3].sum())}
    del batch, acts
    torch.cuda.empty_cache()
    return rec

  # -------------------------------------------------------------------------------------------
  def guarded(self, name, fn):
    """An auxiliary measurement must never cost the headline line: record the error instead."""
    try:
      return fn()
    except SystemExit:
      raise
    except Exception as e:  # pylint: disable=broad-except
      if self.world > 1:
        raise                      # ranks must stay in step: a lone rank skipping a collective would hang
      return {'error': f'{type(e).__name__}: {e}'[:300]}

  def run(self):
    args, world = self.args, self.world
    if args.row_path != 'auto':
      # A/B of the chains' wide rows: lane advance + store stream (bsx_call_t.row_scratch) for every batch / never
      from bsuite_amd.environments import base as _base
      _base.Environment.row_path_min_bytes = 0 if args.row_path == 'on' else None
    if args.workload == 'sweep':
      rec = self.measure_sweep(args.lanes, args.steps, args.warmup)
      if self.rank == 0:
        line = {'metric': 'env-steps/sec', 'value': rec['value'], 'unit': 'env-steps/s', 'n_gpus': world,
                'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': rec['ms_per_step'],
                'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int32+f32',
                'data': 'synthetic (MNIST ids on a synthetic stand-in dataset)',
                'config': {'workload': 'sweep.SWEEP: 468 bsuite_ids as lane segments, random-action ring, dense TimeStep',
                           'global_lanes': rec['global_lanes'], 'segments_per_rank': rec['segments_per_rank'],
                           'sharding': f'whole segments bin-packed over {world} rank(s) by lanes x bytes/step'},
                'roofline': rec['roofline'], 'launch': rec['mode'], 'pipelined': rec['pipelined'],
                **{k: rec[k] for k in ('closed', 'split') if k in rec},
                'episodes_finished': rec['episodes_finished']}
        print(json.dumps(sig(line)), flush=True)
      return

    # --lanes is the GLOBAL batch (BASELINE.json: batch = 2^20 at 1, 2, 4 and 8 GPUs): strong scaling unless --weak
    strong = not args.weak
    lanes = args.lanes
    if strong:
      if lanes % world:
        raise SystemExit('--lanes must be divisible by the number of ranks (strong scaling: the global batch is split evenly)')
      lanes //= world

    mode, chunk = ('graph', args.graph) if args.graph else ('rollout', args.rollout) if args.rollout else ('eager', 0)
    m = self.measure(args.workload, lanes, args.steps, args.warmup, mode, chunk, args.observation_mode, args.logging)
    also = {}
    headline = (args.workload == 'deep_sea' and mode == 'eager' and args.observation_mode == 'dense'
                and not args.logging and not args.no_also)
    if headline:
      # The sub-records are extras of the line, not the contract's K timed steps: their kernels run 10-45 us,
      # so they get at least 200 timed / 40 warm-up steps (a 20-step region of a 10 us kernel measures the
      # synchronisation around it), and say so in their `steps` field.
      K, W = max(args.steps, 200), max(args.warmup, 40)
      K16, W16 = (K + 15) // 16 * 16, (W + 15) // 16 * 16

      self._also_common = {'steps': K, 'lanes_per_gpu': lanes, 'roofline': {'bound': 'hbm', 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s'}}

      def sub(workload, n_lanes, md='eager', ch=0, k=K, w=W):
        return self.guarded(workload, lambda: self.sub_record(self.measure(workload, n_lanes, k, w, md, ch)))

      also['catch/0'] = sub('catch', lanes)                      # the other half of BASELINE.json's metric
      if world == 1:
        # seconds, not milliseconds, of the two headline workloads (HIP-event time per half-second window); they also sit
        # between the short timed regions above and the CPU legs below, so that the run's GPU activity is not one burst
        also['deep_sea/10 sustained'] = self.guarded('sustained', lambda: self.measure_sustained('deep_sea', lanes, args.sustained))
        also['catch/0 sustained'] = self.guarded('sustained', lambda: self.measure_sustained('catch', lanes, args.sustained * 0.4))
        # the wrapped path (every *_noise id, every recorded run): the non-lean kernel instantiations
        also['catch_noise/0'] = sub('catch_noise', lanes)
        also['deep_sea/10 logging'] = self.guarded('logging', lambda: self.sub_record(self.measure('deep_sea', lanes, K, W, 'eager', 0, 'dense', True)))
        # rollout(actions[T,B]) of the two-kernel families: the open-loop form of the same metric, pipelined
        K32 = (K + 31) // 32 * 32
        also['catch/0 r32'] = sub('catch', lanes, 'rollout', 32, K32, 32)
        also['deep_sea/10 r16'] = sub('deep_sea', lanes, 'rollout', 16, 48, 16)
        for w_ in ('cartpole', 'mountain_car'):                  # BASELINE configs[3]
          bid = WORKLOADS[w_][0]
          also[bid] = sub(w_, lanes)
          if w_ == 'mountain_car':                               # (the graph record: the same launches without the host between them)
            also[bid + ' g16'] = sub(w_, lanes, 'graph', 16, K16, W16)
          also[bid + ' r16'] = sub(w_, lanes, 'rollout', 16, K16, W16)
        # the other families north_star names (bandit, memory_chain, umbrella_chain, discounting_chain) and the mnist
        # bandit of the sweep: eager step() and — where a rollout is one fused launch — rollout(16)
        for w_ in ('bandit', 'discounting_chain', 'memory_len', 'umbrella_length', 'mnist'):
          bid = WORKLOADS[w_][0]
          also[bid] = sub(w_, lanes)
          if w_ != 'mnist':
            also[bid + ' r16'] = sub(w_, lanes, 'rollout', 16, K16, W16)
      elif strong:
        # weak scaling as information: the full batch on every GPU
        also['weak'] = dict(self.guarded('weak', lambda: self.sub_record(
            self.measure('deep_sea', args.lanes, args.steps, args.warmup))), scaling='weak', global_lanes=args.lanes * world)
      # BASELINE configs[4]: the heterogeneous sweep, sharded over the ranks by whole segments
      also['sweep'] = self.guarded('sweep', lambda: self.measure_sweep(args.lanes, max(100, K // 2), max(10, W // 2)))

    if self.rank == 0:
      B = lanes
      line = {
          'metric': 'env-steps/sec', 'value': m['value'], 'unit': 'env-steps/s',
          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
          'ms_per_step': m['wall'] / args.steps * 1e3, 'higher_is_better': True,
          'scaling': 'strong' if strong else 'weak',
          'vs_baseline': None, 'dtype': 'f32' if m['family'] in ('cartpole', 'mountain_car') else 'int32',
          'data': 'synthetic',
          'config': {'workload': f"{m['bsuite_id']} {m['family']} {m['okw']} random actions, "
                                 + ('dense TimeStep' if args.observation_mode == 'dense' else
                                    'DELTA observation mode (persistent buffers patched in place; not the dense contract)')
                                 + (', Logging wrapper' if args.logging else ''),
                     'mode': (f'g{args.graph}' if args.graph else f'r{args.rollout}' if args.rollout else 'e'),
                     'lanes_per_gpu': B, 'global_lanes': B * world, 'bytes_per_env_step': m['bytes_per_step'],
                     **({'pinned_cores_per_rank': self.pinned_cores} if world > 1 else {}),
                     'episode_phases': 'lock-step' if args.no_stagger else 'staggered'},
          'roofline': self.roofline(m, full=True),
      }
      if also:
        line['also'] = also
        if getattr(self, '_also_common', None):
          line['also_common'] = self._also_common
      if world == 1 and headline:
        line['host'] = {'step_us': self.host_step_us(), 'box_copy_GBps': self.copy_ceiling()}
      line.update(episodes_finished=m['episodes_finished'], bsuite_info_sums=m['info_sums'], timed_mix=m['timed_mix'])
      if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(m['bsuite_id'], m['family'], m['okw'], m['num_actions'])
      text = json.dumps(sig(line))
      if len(text) >= 7800:                 # the driver keeps an 8000-character tail: never let the line outgrow it
        for k in ('bsuite_info_sums', 'timed_mix'):
          line.pop(k, None)
        text = json.dumps(sig(line, 4))
      print(text, flush=True)

  def close(self):
    if self.collective:
      self.dist.destroy_process_group()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--workload', default='deep_sea', choices=sorted(WORKLOADS) + ['sweep'])
  ap.add_argument('--lanes', type=int, default=1 << 20,
                  help='GLOBAL lane count (BASELINE.json: batch = 2^20), split evenly over the ranks (strong scaling)')
  ap.add_argument('--weak', action='store_true', help='weak scaling: --lanes per GPU')
  ap.add_argument('--strong', action='store_true', help='(the default; kept for older command lines)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--sustained', type=float, default=5.0,
                  help='seconds of eager deep_sea steps in the `also["deep_sea/10 sustained"]` record (catch: 0.4 x)')
  ap.add_argument('--no-also', action='store_true', help='only the main workload (no catch / cartpole / sweep sub-records)')
  ap.add_argument('--no-stagger', action='store_true', help='start all lanes in lock-step (fresh lanes, first call = reset)')
  ap.add_argument('--logging', action='store_true',
                  help='wrap the environment in the batched Logging wrapper (bookkeeping fused into the kernels)')
  ap.add_argument('--observation-mode', default='dense', choices=['dense', 'delta'],
                  help="'delta' (deep_sea, catch): persistent observation buffers patched in place; a "
                       'separate mode with its own byte accounting, NOT the dense contract of the headline')
  ap.add_argument('--sweep-schedule', default='all', choices=['all', 'closed', 'split', 'pipelined'],
                  help='--workload sweep: time this schedule only (profiling passes); the default times all three')
  ap.add_argument('--row-path', default='auto', choices=['auto', 'on', 'off'],
                  help="memory_chain / umbrella_chain rows of more than 8 floats: 'on' = lane advance + wide-row store stream "
                       "(bsx_call_t.row_scratch), also for the sweep group's segments; 'auto' / 'off' = the product default, "
                       'the one-launch LDS bit planes (A/B: measured faster)')
  ap.add_argument('--rollout', type=int, default=0,
                  help='advance this many steps per entry-point call with env.rollout(actions[T,B]) '
                       '(fused T-step kernel for the small-observation families)')
  ap.add_argument('--graph', type=int, default=0,
                  help='capture this many consecutive step() launches into one HIP graph and replay it '
                       '(for the tiny families whose per-step kernel is shorter than a host launch)')
  args = ap.parse_args()

  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(self_launch(args))

  r = Rank(args)
  try:
    r.run()
  finally:
    r.close()


if __name__ == '__main__':
  main()
