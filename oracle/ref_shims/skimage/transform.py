"""Stand-in for `skimage.transform.resize` (scikit-image is not installed and cannot be fetched).

TEST INFRASTRUCTURE ONLY.  Follows the published algorithm of scikit-image >= 0.19 for the only
call the reference makes (bsuite/utils/wrappers.py:216: `resize(observation, shape[:2],
preserve_range=True)`, i.e. order=1, mode='reflect', clip=True, anti_aliasing=None):

  * float32 input stays float32 (`convert_to_float(..., preserve_range=True)`);
  * anti_aliasing=None means: Gaussian pre-filter iff some output dimension is smaller than the
    input's (float images), sigma = max(0, (in/out - 1) / 2) per axis, applied with
    `scipy.ndimage.gaussian_filter(image, sigma, cval=cval, mode='mirror')`;
  * `scipy.ndimage.zoom(image, 1/factors, order=1, mode='mirror', grid_mode=True)`
    ('reflect' in numpy.pad vocabulary is ndimage's 'mirror');
  * the result is clipped to the input's [min, max].

Parity with a real skimage is UNPINNED — only scipy.ndimage, which skimage delegates to, is present
here; the control flow above is transcribed from skimage/transform/_warps.py (resize, 0.19-0.22).
"""
import numpy as np
from scipy import ndimage as ndi


def resize(image, output_shape, order=None, mode='reflect', cval=0, clip=True, preserve_range=False,
           anti_aliasing=None, anti_aliasing_sigma=None):
  image = np.asarray(image)
  output_shape = tuple(int(s) for s in output_shape)
  if order not in (None, 1) or mode != 'reflect' or not preserve_range or anti_aliasing is not None:
    raise NotImplementedError('skimage stand-in: only resize(obs, shape, preserve_range=True)')
  if len(output_shape) != image.ndim:
    raise NotImplementedError('skimage stand-in: output rank must equal input rank')
  if image.dtype == np.float16:
    image = image.astype(np.float32)
  if image.dtype.char not in 'df':
    image = image.astype(float)
  factors = np.divide(image.shape, output_shape)
  if anti_aliasing is None:
    anti_aliasing = any(o < i for o, i in zip(output_shape, image.shape))
  if anti_aliasing:
    if anti_aliasing_sigma is None:
      anti_aliasing_sigma = np.maximum(0, (factors - 1) / 2)
    filtered = ndi.gaussian_filter(image, anti_aliasing_sigma, cval=cval, mode='mirror')
  else:
    filtered = image
  zoom_factors = [1 / f for f in factors]
  out = ndi.zoom(filtered, zoom_factors, order=1, mode='mirror', cval=cval, grid_mode=True)
  if clip:
    np.clip(out, np.min(image), np.max(image), out=out)
  return out
