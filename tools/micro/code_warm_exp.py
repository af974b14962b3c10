"""Is the +10 us that the sweep's phase 0 pays behind the store stream instruction fetch?  Mode B puts a MINI phase 0
(468 ids x 64 lanes: every family's code path, 468 workgroups) between the big stream and the big phase 0: the code is
then in every XCD's L2 when the big launch starts, its data is as cold as before.  Run under rocprofv3 --kernel-trace."""
import os, sys, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/micro/ -> repo root
sys.path.insert(0, ROOT)
from bsuite_amd import sweep_batch as sb, _native
from bsuite_amd.utils import datasets

mode = sys.argv[1]
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'mnist_synthetic_dataset.npz'))
tmp = tempfile.mkdtemp(prefix='bsx_mnist_')
datasets.write_idx_files(tmp, d['images_u8'], d['labels'])
mn = dict(data_dir=tmp)
kw = dict(mnist=mn, mnist_noise=mn, mnist_scale=mn)
big = sb.SweepBatch(None, 1 << 20, seed=42, env_kwargs=kw)
acts = big.random_actions(seed=1, ring=16)
big.prepare_groups(acts, pipelined=False)
mini = sb.SweepBatch(None, 468 * 64, seed=43, env_kwargs=kw)
macts = mini.random_actions(seed=2, ring=16)
mini.prepare_groups(macts, pipelined=False)
st = torch.cuda.current_stream().cuda_stream
phase = _native.lib.bsx_group_step_phase
hb, hm = big._groups[0], mini._groups[0]
for it in range(80):
  _native.check(phase(hb, 0, st), 'p0')
  _native.check(phase(hb, 1, st), 'p1')
  if mode == 'B':
    _native.check(phase(hm, 0, st), 'mini p0')
  elif mode == 'C':                          # control: the mini launch BEFORE the stream (no warming effect left after it)
    pass
torch.cuda.synchronize()
print('done', mode)
