"""Is torch's fill kernel really faster than ours?  Interleaved event timings + (optionally under
rocprofv3) per-kernel durations."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsuite_amd import _native
n = (1 << 20) * 3600
b8 = torch.empty(n, dtype=torch.uint8, device='cuda')
b32 = b8.view(torch.float32)
st = torch.cuda.current_stream().cuda_stream
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tests = {
  'torch_u8_zero': lambda: b8.zero_(),
  'torch_u8_fill7': lambda: b8.fill_(7),
  'torch_f32_zero': lambda: b32.zero_(),
  'torch_f32_fill1.5': lambda: b32.fill_(1.5),
  'bsx_calib_fill': lambda: _native.lib.bsx_calib_fill(b8.data_ptr(), n, 0, st),
  'bsx_calib_fill_nt': lambda: _native.lib.bsx_calib_fill(b8.data_ptr(), n, 1, st),
}
for rep in range(3):
    for k, fn in tests.items():
        ms = ev(fn)
        print(json.dumps({'rep': rep, 'name': k, 'ms': round(ms, 4), 'TBps': round(n / ms / 1e9, 3)}), flush=True)
# single isolated launches (idle gaps between them)
import time
for k, fn in tests.items():
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); time.sleep(0.05)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1), 4))
    print(json.dumps({'isolated': k, 'ms': ts}), flush=True)
