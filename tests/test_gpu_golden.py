"""GPU: the HIP kernels, through the C ABI, against outputs of the unmodified reference
(tests/golden/*.npz).  Integer / grid families: bit-exact (f32 reward == np.float32(reference f64
reward), observation bitwise).  Physics families: per-step teacher-forced, |a-b| <= 1e-6*max(1,|b|)
(BASELINE.json north_star tolerance)."""
import numpy as np
import pytest
import torch

from tests import engine_util as eu
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
TOL = 1e-6



@pytest.mark.parametrize('name', gu.replay_case_names())
def test_engine_matches_reference(name):
  meta, g = gu.load_case(name)
  eu.check_against_case(name, meta, g)
