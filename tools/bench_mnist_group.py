"""Times the grouped mnist launch pair of a sweep made only of the 60 mnist ids (2240 lanes each)."""
import json, os, sys, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bsuite_amd import sweep, sweep_batch as sb  # noqa: E402
from bsuite_amd.utils import datasets  # noqa: E402

d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
tmp = tempfile.mkdtemp(prefix='bsx_mnist_')
datasets.write_idx_files(tmp, d['images_u8'], d['labels'])
mn = dict(data_dir=tmp)
ids = [i for i in sweep.SWEEP if i.startswith('mnist')]
seg = int(os.environ.get('SEG_LANES', '2240'))
batch = sb.SweepBatch(ids, seg * len(ids), seed=42, env_kwargs=dict(mnist=mn, mnist_noise=mn, mnist_scale=mn))
acts = batch.random_actions(seed=1)
batch.prepare_groups(acts)
for _ in range(10):
  batch.step_grouped()
best = 1e9
for _ in range(3):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(100):
    batch.step_grouped()
  e1.record(); torch.cuda.synchronize()
  best = min(best, e0.elapsed_time(e1) / 100)
gb = batch.lanes() * (13 + 3136 + 8) / 1e9
print(json.dumps(dict(variant=os.environ.get('BSX_MNIST_VARIANT', '0'), map=os.environ.get('BSX_GROUP_MAP', '1'), seg_lanes=seg,
                      lanes=batch.lanes(), ms=round(best, 4), GBps=round(gb / best * 1e3, 1))))
