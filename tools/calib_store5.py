import ctypes, json, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
so = os.path.join(here, 'calib_store5.so')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, os.path.join(here, 'calib_store5.hip')])
lib = ctypes.CDLL(so)
lib.calib5.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
n = (1 << 20) * 3600
b8 = torch.empty(n, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
names = ['plain', 'sc0', 'sc1', 'sc0 sc1', 'nt', 'sc0 nt', 'sc1 nt', 'sc0 sc1 nt']
for rep in range(3):
    for v, name in enumerate(names):
        ms = ev(lambda v=v: lib.calib5(b8.data_ptr(), n, v, st))
        print(json.dumps({'rep': rep, 'store_bits': name, 'ms': round(ms, 4), 'TBps': round(n / ms / 1e9, 3)}), flush=True)
