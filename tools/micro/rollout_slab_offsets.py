"""Does the placement of the four [T, B] output slabs of a fused rollout relative to each other matter (HBM channel
phase)?  All four carved out of ONE allocation with chosen byte offsets between them; us per step of mountain_car /
cartpole rollouts of T = 16 at 2^20 lanes for several offsets, each measured three times.

  python tools/micro/rollout_slab_offsets.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bsuite_amd  # noqa: E402
from bsuite_amd import _native  # noqa: E402

B, T = 1 << 20, 16
STAGGER = '--stagger' in sys.argv
PRE_STEPS = int(sys.argv[sys.argv.index('--pre-steps') + 1]) if '--pre-steps' in sys.argv else 0
ROUNDTRIP = '--roundtrip' in sys.argv
PERIOD = int(sys.argv[sys.argv.index('--period') + 1]) if '--period' in sys.argv else 0   # lanes at phases i % PERIOD (0: the episode length)
OFFS = [int(x) for x in sys.argv[sys.argv.index('--offsets') + 1].split(',')] if '--offsets' in sys.argv else [0, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 1024, (1 << 20) + 4096]


def carve(pool, at, shape, dtype):
  n = 1
  for s in shape:
    n *= s
  nbytes = n * torch.tensor([], dtype=dtype).element_size()
  t = pool[at:at + nbytes].view(dtype).view(shape)
  return t, at + nbytes


def run(bid, off):
  env = bsuite_amd.load_from_id(bid, batch=B, seed=1, num_buffers=2)
  raw = env.raw_env if hasattr(env, 'raw_env') else env
  numel = 1
  for s in raw._obs_shape:
    numel *= s
  acts = torch.randint(0, env.action_spec().num_values, (32, B), dtype=torch.int32, device='cuda')
  if STAGGER:
    import bench                                                # lanes at staggered episode phases, as bench.py times them
    bench.stagger_phases(env, acts, PERIOD if PERIOD else bench.WORKLOADS[bid.split('/')[0]][5])
  for k in range(PRE_STEPS):                                    # eager step() calls before the rollouts
    env.step(acts[k % 32])
  if ROUNDTRIP:
    raw.load_state_dict(raw.state_dict())
  acts = acts[:T].contiguous()
  env.rollout(acts)
  if off is not None:
    pool = torch.empty(T * B * (9 + 4 * numel) + 8 * (off + 4096) + (1 << 22), dtype=torch.uint8, device='cuda')
    base = (-pool.data_ptr()) % (1 << 21)                         # 2 MiB aligned start
    at = base
    r, at = carve(pool, at, (T, B), torch.float32); at += off
    d, at = carve(pool, at, (T, B), torch.float32); at += off
    o, at = carve(pool, at, (T, B) + tuple(raw._obs_shape), torch.float32); at += off
    s, at = carve(pool, at, (T, B), torch.int8)
    out = dict(reward=r, discount=d, step_type=s, observation=o)
    raw._rollout_out[T] = (out, _native.TimeStepPtrs(r.data_ptr(), d.data_ptr(), s.data_ptr(), o.data_ptr()))
  for _ in range(4):
    env.rollout(acts)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  res = []
  for _ in range(3):
    e0.record()
    for _ in range(24):
      env.rollout(acts)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / (24 * T) * 1e3)
  del env
  torch.cuda.empty_cache()
  return res


for bid in ('mountain_car/0', 'cartpole/0'):
  for off in [None] + OFFS:
    r = run(bid, off)
    print(f'{bid:16s} slabs {"as torch allocates them" if off is None else "one pool, +%7d B between" % off}:  ' + '  '.join(f'{x:6.2f}' for x in r) + '  us per step', flush=True)
