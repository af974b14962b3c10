"""GPU: the device draw stream (include/bsx_stream.h, compiled by hipcc) against the oracle's two
independent restatements (numpy: oracle/stream.py, C: oracle/oracle.c) — bit for bit, including the
normal transform (no fused multiply-add, IEEE sqrt/div only)."""
import numpy as np
import pytest
import torch

from oracle import coracle
from oracle import stream as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed,lane0,step,stream_id', [
    (42, 0, 0, 0), (0xDEADBEEFCAFE, (1 << 33) + 5, (1 << 34) + 77, 1), (7, 1000003, 12345, 0)])
def test_device_words_and_normals(seed, lane0, step, stream_id):
  from bsuite_amd import _native
  n_lanes, n_words = 4096, 64
  words = torch.zeros((n_lanes, n_words), dtype=torch.int32, device='cuda')
  normals = torch.zeros((n_lanes, n_words // 2), dtype=torch.float64, device='cuda')
  rc = _native.lib.bsx_stream_dump(seed, lane0, n_lanes, step, stream_id, n_words, words.data_ptr(),
                                   normals.data_ptr(), torch.cuda.current_stream().cuda_stream)
  assert rc == 0
  w = words.cpu().numpy().view(np.uint32)
  ref = S.words(seed, np.arange(lane0, lane0 + n_lanes, dtype=np.uint64), step, stream_id, n_words)
  np.testing.assert_array_equal(w, ref)
  np.testing.assert_array_equal(w[5], coracle.stream_words(seed, lane0 + 5, step, stream_id, n_words))
  k = S.k53(ref[:, 0::2], ref[:, 1::2]).reshape(-1)
  z_np = S.normal_from_k53(k)
  z_c = coracle.normals(k)
  z_dev = normals.cpu().numpy().reshape(-1)
  np.testing.assert_array_equal(z_np.view(np.uint64), z_c.view(np.uint64))
  np.testing.assert_array_equal(z_dev.view(np.uint64), z_c.view(np.uint64))
  assert abs(z_dev.mean()) < 0.02 and abs(z_dev.std() - 1) < 0.02
