import ctypes, json, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from bsuite_amd import _native
so = os.path.join(here, 'calib_store4.so')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, os.path.join(here, 'calib_store4.hip')])
lib = ctypes.CDLL(so)
lib.calib4.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
n = (1 << 20) * 3600
b8 = torch.empty(n, dtype=torch.uint8, device='cuda'); b32 = b8.view(torch.float32)
hot = torch.randint(0, 900, (1 << 20,), dtype=torch.int32, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tests = {'torch_f32_fill': lambda: b32.fill_(1.5), 'bsx_calib_fill': lambda: _native.lib.bsx_calib_fill(b8.data_ptr(), n, 0, st)}
for v, name in enumerate(['noloop_k1', 'noloop_k2', 'noloop_k4', 'noloop_k8', 'noloop32_k2', 'noloop32_k4', 'noloop_hot_k2', 'noloop_hot_k4']):
    tests[name] = (lambda v=v: lib.calib4(b8.data_ptr(), n, v, hot.data_ptr(), st))
for rep in range(3):
    for k, fn in tests.items():
        ms = ev(fn)
        print(json.dumps({'rep': rep, 'name': k, 'ms': round(ms, 4), 'TBps': round(n / ms / 1e9, 3)}), flush=True)
