"""Times bsx_image_observation (ImageObservation adapter) as a store stream: achieved HBM GB/s =
image bytes written / kernel time (HIP events on the launch stream).  Run on the GPU box."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsuite_amd.utils import wrappers  # noqa: E402


def run(obs_shape, shape, lanes, reps=30):
  obs = torch.rand((lanes,) + obs_shape, device='cuda')
  out = torch.empty((lanes,) + shape, device='cuda')
  for _ in range(5):
    wrappers.to_image(shape, obs, out=out)
  ms = float('inf')
  for _ in range(3):                               # best of 3 rounds (first-touch / clock ramp noise)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
      wrappers.to_image(shape, obs, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = min(ms, e0.elapsed_time(e1) / reps)
  gb = out.numel() * 4 / 1e9
  return dict(obs_shape=list(obs_shape), shape=list(shape), lanes=lanes, ms=round(ms, 4),
              GBps=round(gb / (ms / 1e3), 1), frac_of_8TBps=round(gb / (ms / 1e3) / 8000, 3),
              images_per_s=round(lanes / (ms / 1e3)))


if __name__ == '__main__':
  for obs_shape, shape, lanes in [((10, 5), (84, 84, 4), 4096), ((30, 30), (84, 84, 4), 4096),
                                  ((1, 6), (84, 84, 4), 4096), ((1, 3), (84, 84, 4), 4096),
                                  ((10, 5), (84, 84), 16384), ((28, 28), (84, 84, 3), 4096)]:
    print(json.dumps(run(obs_shape, shape, lanes)), flush=True)
