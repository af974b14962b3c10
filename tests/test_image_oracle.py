"""CPU: the to_image restatement (oracle/image_oracle.py) is pinned (a) bit for bit against
scipy.ndimage.zoom — the routine skimage >= 0.19 `resize` delegates to — and (b) against fixtures
written by the reference's own `to_image` (bsuite/utils/wrappers.py:222-247) in the build container;
plus the argument checks of bsx_image_observation that return before any launch."""
import ctypes

import numpy as np
import pytest
import torch  # noqa: F401  (the HIP runtime the library links against)

from bsuite_amd import _native
from oracle import image_oracle as io
from tests import golden_util as gu


@pytest.mark.parametrize('dims', [(1, 6, 84, 84), (10, 5, 84, 84), (30, 30, 84, 84), (1, 23, 84, 84),
                                  (28, 28, 84, 84), (50, 50, 84, 84), (1, 3, 7, 9), (2, 2, 5, 5),
                                  (3, 7, 3, 7), (10, 5, 11, 6), (1, 103, 84, 120), (5, 1, 9, 4)])
def test_bilinear_restatement_equals_scipy_zoom(dims):
  ndi = pytest.importorskip('scipy.ndimage')
  ih, iw, oh, ow = dims
  rng = np.random.default_rng(ih * 1000 + iw)
  for trial in range(3):
    a = (rng.standard_normal((ih, iw)) if trial else (rng.random((ih, iw)) > 0.7)).astype(np.float32)
    ref = ndi.zoom(a, [oh / ih, ow / iw], order=1, mode='mirror', grid_mode=True)
    got = io.resize_bilinear(a, (oh, ow))
    assert ref.shape == got.shape == (oh, ow) and got.dtype == np.float32
    np.testing.assert_array_equal(ref.view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize('dims', [(10, 5, 6, 4), (10, 5, 8, 8), (30, 30, 12, 12), (1, 103, 84, 84), (28, 28, 7, 9),
                                  (50, 50, 10, 84), (1, 412, 84, 84), (64, 64, 3, 2), (9, 200, 9, 20), (7, 7, 6, 6)])
def test_downscaling_restatement_equals_scipy_gaussian_then_zoom(dims):
  """An axis that shrinks is pre-filtered (skimage anti_aliasing=None -> sigma=(in/out-1)/2) before the zoom."""
  ndi = pytest.importorskip('scipy.ndimage')
  ih, iw, oh, ow = dims
  rng = np.random.default_rng(ih * 1000 + iw + oh)
  for trial in range(3):
    a = (rng.standard_normal((ih, iw)) if trial else (rng.random((ih, iw)) > 0.7)).astype(np.float32)
    factors = np.divide((ih, iw), (oh, ow))
    filtered = ndi.gaussian_filter(a, np.maximum(0, (factors - 1) / 2), cval=0, mode='mirror')
    assert filtered.dtype == np.float32
    ref = ndi.zoom(filtered, [1 / f for f in factors], order=1, mode='mirror', cval=0, grid_mode=True)
    got = io.resize_bilinear(a, (oh, ow))
    assert ref.shape == got.shape == (oh, ow) and got.dtype == np.float32
    np.testing.assert_array_equal(ref.view(np.uint32), got.view(np.uint32))
    assert got.min() >= a.min() and got.max() <= a.max()             # skimage's clip=True never bites


@pytest.mark.parametrize('case', gu.image_adapter_cases(), ids=lambda c: c[0])
def test_oracle_reproduces_reference_to_image(case):
  _, shape, obs, image = case
  got = io.to_image(shape, obs, batched=True)
  assert got.dtype == image.dtype == np.float32
  np.testing.assert_array_equal(got.view(np.uint32), image.view(np.uint32))
  one = io.to_image(shape, obs[0])
  np.testing.assert_array_equal(one.view(np.uint32), image[0].view(np.uint32))


def test_reference_unit_test_properties():
  # wrappers_test.py:135-142: shape and value set are preserved
  for shape, ob in (((84, 84, 4), np.array([1, 2], np.float32)), ((70, 90), np.array([[1, 0, 2, 3]], np.float32))):
    img = io.to_image(shape, ob)
    assert img.shape == shape
    assert sorted(np.unique(img)) == sorted(np.unique(ob))
  with pytest.raises(ValueError):
    io.to_image((8, 8), np.zeros((2, 3, 4), np.float32))


def _call(cfg, n, obs=0, image=0):
  return _native.lib.bsx_image_observation(ctypes.byref(cfg) if cfg is not None else None, n, obs, image, None)


def test_entry_point_argument_errors_without_a_device():
  ok = _native.ImageCfg(_native.IMAGE_BILINEAR, 10, 5, 84, 84, 4)
  assert _call(None, 1) == _native.BSX_ENULL
  assert _call(ok, 0) == 0                                             # empty batch: nothing to launch
  assert _call(ok, -1) == _native.BSX_EINVAL
  assert _call(ok, 4, 0, 0) == _native.BSX_ENULL
  assert _call(ok, 4, 64, 72) == _native.BSX_EALIGN                    # image must be 16-byte aligned
  assert _call(_native.ImageCfg(7, 1, 2, 8, 8, 1), 1) == _native.BSX_EINVAL
  assert _call(_native.ImageCfg(_native.IMAGE_SMALL, 1, 5, 8, 8, 1), 1) == _native.BSX_ERANGE      # size > 4
  assert _call(_native.ImageCfg(_native.IMAGE_BILINEAR, 10, 5, 8, 8, 1, 65), 1) == _native.BSX_ERANGE  # filter radius
  assert _call(_native.ImageCfg(_native.IMAGE_BILINEAR, 2, 2, 2000, 8, 1), 1) == _native.BSX_ERANGE
  assert _call(_native.ImageCfg(_native.IMAGE_BILINEAR, 2, 2, 1024, 1024, 1), 1) == _native.BSX_ERANGE  # >= 2^20 floats


def test_host_rule_selection_matches_to_image():
  from bsuite_amd.utils import wrappers as w
  assert w._image_cfg((84, 84, 4), (1, 3)).mode == _native.IMAGE_SMALL
  c = w._image_cfg((84, 84, 4), (10, 5))
  assert (c.mode, c.in_rows, c.in_cols, c.out_rows, c.out_cols, c.tail) == (_native.IMAGE_BILINEAR, 10, 5, 84, 84, 4)
  c = w._image_cfg((14, 21), (7,))
  assert (c.in_rows, c.in_cols, c.tail) == (1, 7, 1)
  with pytest.raises(ValueError):
    w._image_cfg((8, 8), (2, 3, 4))
  c = w._image_cfg((6, 4), (10, 5))                  # down-scaling: anti-aliasing Gaussian half-kernels
  assert (c.radius_y, c.radius_x) == (1, 1)
  np.testing.assert_array_equal(np.array(c.gauss_y[:2]), io.gaussian_half_kernel(10, 6))
  np.testing.assert_array_equal(np.array(c.gauss_x[:2]), io.gaussian_half_kernel(5, 4))
  c = w._image_cfg((84, 84), (1, 103))               # sigma 0.113: a one-tap kernel is the identity
  assert (c.radius_y, c.radius_x) == (0, 0)
  with pytest.raises(AssertionError):
    w._image_cfg((8,), (1, 2))


def test_host_half_kernels_are_scipys():
  """The anti-aliasing weights handed to the kernel == scipy.ndimage's own Gaussian kernel (truncate 4) for
  skimage's sigma = (in/out - 1)/2, element for element; an axis scipy would skip gets no kernel."""
  filters = pytest.importorskip('scipy.ndimage._filters')
  from bsuite_amd.utils import wrappers as w
  for n_in, n_out in [(10, 6), (5, 4), (30, 12), (103, 84), (28, 7), (28, 9), (50, 10), (412, 84), (64, 3), (64, 2), (200, 20),
                      (7, 6), (9, 9), (5, 84)]:
    half = w._gaussian_half_kernel(n_in, n_out)
    sigma = max(0.0, (n_in / n_out - 1) / 2)
    radius = int(4.0 * sigma + 0.5)
    if sigma <= 1e-15 or radius == 0:
      assert half is None
      continue
    ref = filters._gaussian_kernel1d(sigma, 0, radius)
    np.testing.assert_array_equal(half, ref[radius:])
    np.testing.assert_array_equal(ref[:radius][::-1], ref[radius + 1:])      # symmetric: one half suffices
    np.testing.assert_array_equal(half, io.gaussian_half_kernel(n_in, n_out))
