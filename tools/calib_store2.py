import ctypes, os, json, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "calib_store2.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "calib_store2.hip")])
lib = ctypes.CDLL(so)
lib.calib2.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
lib.calib2_memset.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
nbytes = (1 << 20) * 3600
buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = ["k1_b256","k2_b256","k4_b256","k8_b256","s4_b256","s8_b256","s16_b256","s4_b512","s4_b1024","s4_b64","s4_b128","k4_b256_nt","s8_b256_nt","k4_b1024","k4_b64","dword_b256"]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = []
for v, n in enumerate(names):
    for grid in (1024, 2048, 4096, 16384, 65536, 262144):
        ms = timeit(lambda: lib.calib2(buf.data_ptr(), nbytes, v, grid, st))
        res.append((round(nbytes / ms / 1e9, 3), n, grid))
        print(json.dumps({"variant": n, "grid": grid, "ms": round(ms, 4), "TBps": round(nbytes / ms / 1e9, 3)}), flush=True)
ms = timeit(lambda: lib.calib2_memset(buf.data_ptr(), nbytes, st))
print(json.dumps({"variant": "hipMemsetAsync", "ms": round(ms, 4), "TBps": round(nbytes / ms / 1e9, 3)}))
print("TOP", sorted(res, reverse=True)[:10])
