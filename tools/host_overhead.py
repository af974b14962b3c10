"""Host cost of one env.step() call (the tiny families' kernels run 5-7 us at 2^20 lanes: a slower host
leaves the GPU idle between launches).  Small batch so that the GPU is never the bottleneck; every number is
microseconds per call over N calls, wall clock, with one synchronisation at the end.

  python tools/host_overhead.py [--n 20000]
"""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bsuite_amd  # noqa: E402
from bsuite_amd import _native  # noqa: E402


def per_call(fn, n):
  fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=20000)
  ap.add_argument('--batch', type=int, default=256)
  args = ap.parse_args()
  n = args.n
  dev = torch.device('cuda:0')
  out = {}
  for bid in ('bandit/0', 'mountain_car/0', 'catch/0'):
    env = bsuite_amd.load_from_id(bid, batch=args.batch, num_buffers=2)
    na = env.action_spec().num_values
    acts = torch.randint(na, (64, args.batch), device=dev, dtype=torch.int32)
    a0 = acts[0].contiguous()
    env.step(a0)
    k = [0]

    def full():
      k[0] = (k[0] + 1) & 63
      env.step(acts[k[0]])
    out[f'{bid} env.step(actions[t])'] = per_call(full, n)
    out[f'{bid} env.step(a0)'] = per_call(lambda: env.step(a0), n)
    raw = env.raw_env if hasattr(env, 'raw_env') else env
    ptr = a0.data_ptr()
    out[f'{bid} env._call(ptr, False)'] = per_call(lambda: raw._call(ptr, False), n)
    call, outp = raw._call_desc, raw._out_ptrs[0]
    call.hip_stream = torch.cuda.current_stream(dev).cuda_stream
    call.stream.step_index = 5
    fn = getattr(_native.lib, f'bsx_{raw._abi_name}_step')
    argv = raw._native_args(call, ptr, outp)
    out[f'{bid} bsx_{raw._abi_name}_step(*prebuilt args)'] = per_call(lambda: fn(*argv), n)
    out[f'{bid} _native_args + C call'] = per_call(lambda: fn(*raw._native_args(call, ptr, outp)), n)
  out['acts[k] (tensor indexing)'] = per_call(lambda: acts[3], n)
  out['torch.cuda.current_stream(dev).cuda_stream'] = per_call(lambda: torch.cuda.current_stream(dev).cuda_stream, n)
  out['lib.bsx_abi_version() (ctypes round trip)'] = per_call(lambda: _native.lib.bsx_abi_version(), n)
  x = torch.zeros(256, device=dev)
  out['torch x.add_(1) (one tiny kernel through torch)'] = per_call(lambda: x.add_(1), n)
  for k_, (host, total) in out.items():
    print(f'{k_:58s} host {host:6.2f} us/call   incl. final sync {total:6.2f}')


if __name__ == '__main__':
  main()
