import os, sys, time, ctypes, torch
sys.path.insert(0, os.getcwd())
import bsuite_amd
from bsuite_amd import _native
dev = torch.device('cuda:0')
B = 1 << 20
for bid in ('bandit/0', 'discounting_chain/0', 'mountain_car/0', 'cartpole/0'):
  env = bsuite_amd.load_from_id(bid, batch=B, num_buffers=2, seed=42)
  na = env.action_spec().num_values
  acts = torch.randint(na, (32, B), device=dev, dtype=torch.int32)
  for t in range(40):
    env.step(acts[t % 32])
  raw = env
  call, outp = raw._call_desc, raw._out_ptrs[0]
  fn = getattr(_native.lib, f'bsx_{raw._abi_name}_step')
  argv = list(raw._native_args(call, acts[0].data_ptr(), outp))
  def c_loop(n):
    for _ in range(n):
      fn(*argv)
  def py_loop(n):
    for t in range(n):
      env.step(acts[t % 32])
  for name, loop in (('C entry point in a loop', c_loop), ('env.step(actions[t])', py_loop)):
    res = []
    for rep in range(3):
      loop(50)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      t0 = time.perf_counter(); e0.record(); loop(400); th = time.perf_counter() - t0; e1.record(); torch.cuda.synchronize()
      tw = time.perf_counter() - t0
      res.append((e0.elapsed_time(e1) / 400 * 1e3, th / 400 * 1e6, tw / 400 * 1e6))
    print(f'{bid:22s} {name:26s} GPU us/step {[round(r[0], 2) for r in res]}  host enqueue us/step {[round(r[1], 2) for r in res]}  wall {[round(r[2], 2) for r in res]}')
