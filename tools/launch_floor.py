"""How long does the smallest useful kernel take?  Pure-store fill kernels (bsx_calib_fill: one 16-byte store
per thread, no loop) of growing size, timed back to back with HIP events — the floor under every
"one launch per step()" family: time = fixed launch/drain cost + bytes / bandwidth.

  python tools/launch_floor.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bsuite_amd import _native  # noqa: E402


def main():
  dev = torch.device('cuda:0')
  buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
  st = torch.cuda.current_stream(dev).cuda_stream
  for mb in (0.25, 1, 4, 13, 26, 51, 89, 128, 232, 512, 1024):
    n = int(mb * (1 << 20)) // 16 * 16
    for _ in range(20):
      _native.lib.bsx_calib_fill(buf.data_ptr(), n, 0, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 400
    for _ in range(reps):
      _native.lib.bsx_calib_fill(buf.data_ptr(), n, 0, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(json.dumps(dict(MiB=mb, us_per_launch=round(us, 2), GBps=round(n / us / 1e3, 1))), flush=True)


if __name__ == '__main__':
  main()
