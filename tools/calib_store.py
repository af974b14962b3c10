"""Pure-store bandwidth calibration on the GPU box (run through gpurun)."""
import ctypes, os, sys, json, subprocess
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "calib_store.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "calib_store.hip")])
lib = ctypes.CDLL(so)
lib.calib_fill.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
lib.calib_fill.restype = ctypes.c_int
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).total_memory / 2**30, "GiB")
nbytes = (1 << 20) * 3600
buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(name, variant, grid=2048, tile=0, reps=20, moved=nbytes):
    for _ in range(3):
        rc = lib.calib_fill(buf.data_ptr(), nbytes, variant, grid, tile, st); assert rc == 0, rc
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.calib_fill(buf.data_ptr(), nbytes, variant, grid, tile, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"name": name, "ms": round(ms, 4), "TBps": round(moved / ms / 1e9, 3)}), flush=True)
for g in (1024, 2048, 4096, 8192, 16384):
    run(f"fill_gs grid={g}", 0, grid=g)
for g in (2048, 8192):
    run(f"fill_gs_nt grid={g}", 1, grid=g)
for t in (3600 * 16, 3600 * 64, 3600 * 256, 3600 * 1024):
    run(f"fill_tile tile={t}", 2, tile=t)
    run(f"fill_tile_nt tile={t}", 3, tile=t)
    run(f"fill_tile_u4 tile={t}", 4, tile=t)
for g in (2048, 8192):
    run(f"copy_gs grid={g} (rd+wr bytes)", 5, grid=g, moved=nbytes)
# torch memset for comparison
for _ in range(3): buf.zero_()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): buf.zero_()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(json.dumps({"name": "torch.zero_", "ms": round(ms, 4), "TBps": round(nbytes / ms / 1e9, 3)}))
