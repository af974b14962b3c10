"""Times the mnist bandit step (advance + observe pair) on the synthetic stand-in dataset."""
import json, os, sys, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bsuite_amd.environments import mnist  # noqa: E402

d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
B = 1 << 17
env = mnist.MNISTBandit(images=d['images_u8'].view(np.int8), labels=d['labels'], seed=1, batch=B, num_buffers=1)
acts = torch.randint(10, (B,), device='cuda', dtype=torch.int32)
for _ in range(10):
  env.step(acts)
best = 1e9
for _ in range(3):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(100):
    env.step(acts)
  e1.record(); torch.cuda.synchronize()
  best = min(best, e0.elapsed_time(e1) / 100)
gb = B * (13 + 3136 + 8) / 1e9
print(json.dumps(dict(variant=os.environ.get('BSX_MNIST_VARIANT', '0'), lanes=B, ms=round(best, 4), GBps=round(gb / best * 1e3, 1),
                      frac=round(gb / best * 1e3 / 8000, 3))))
