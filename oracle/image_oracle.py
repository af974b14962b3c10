"""ORACLE (test infrastructure, never imported by the product): numpy restatement of
`to_image` — bsuite/utils/wrappers.py:178-247.

`_small_state_to_image` (:178-204) is restated literally.  `_interpolate_to_image` (:207-219)
calls skimage.transform.resize, a third-party dependency that is NOT under /root/reference and not
installed (setup.py lists `scikit-image` unpinned); for scikit-image >= 0.19 that call is
scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True) — preceded, when an output dimension is
smaller than the input's, by the anti-aliasing scipy.ndimage.gaussian_filter(sigma=(in/out-1)/2,
mode='mirror') — whose arithmetic `resize_bilinear` restates operation by operation in f64 (ni_interpolation.c
NI_ZoomShift: cc = (k+0.5)*zoom-0.5, map_coordinate mirror, weights (1-f, f), 4-term sum in C
order).  Pinned: bit for bit against scipy.ndimage.zoom (tests/test_image_oracle.py) and against
fixtures written by the reference's own to_image running over that scipy call
(tests/golden/image_*.npz, oracle/make_golden.py).  Parity against a real skimage: UNPINNED.
"""
import numpy as np


def _mirror_index(i, n):
  if n <= 1:
    return 0
  s2 = 2 * n - 2
  if i < 0:
    i = s2 * (-i // s2) + i
    return i + s2 if i <= 1 - n else -i
  if i >= n:
    i -= s2 * (i // s2)
    if i >= n:
      i = s2 - i
  return i


def _mirror_coord(c, n):
  if c < 0:
    if n <= 1:
      return 0.0
    s2 = 2 * n - 2
    c = s2 * int(-c / s2) + c
    return c + s2 if c <= 1 - n else -c
  if c > n - 1:
    if n <= 1:
      return 0.0
    s2 = 2 * n - 2
    c -= s2 * int(c / s2)
    if c >= n:
      c = s2 - c
  return c


def axis_table(n_in, n_out):
  """Source index pair and weight pair of every output coordinate along one axis."""
  zoom = np.float64(n_in) / np.float64(n_out)
  i0 = np.zeros(n_out, np.int64)
  i1 = np.zeros(n_out, np.int64)
  w0 = np.zeros(n_out)
  w1 = np.zeros(n_out)
  for k in range(n_out):
    cc = np.float64(k)
    cc = cc + 0.5
    cc = cc * zoom
    cc = cc - 0.5
    cc = np.float64(_mirror_coord(cc, n_in))
    fl = np.floor(cc)
    t = cc - fl
    w0[k] = 1.0 - t
    w1[k] = t
    i0[k] = _mirror_index(int(fl), n_in)
    i1[k] = _mirror_index(int(fl) + 1, n_in)
  return i0, i1, w0, w1


def gaussian_half_kernel(n_in, n_out):
  """skimage's anti-aliasing Gaussian along one axis as scipy builds it: sigma = max(0, (in/out-1)/2)
  (skimage/transform/_warps.py resize), radius = int(4*sigma + 0.5) and normalised weights
  exp(-0.5/sigma^2 * x^2) / sum (scipy.ndimage._filters._gaussian_kernel1d).  Returns the weights at
  distance 0..radius, or None when the axis is not filtered (sigma <= 1e-15, or a 1-tap kernel)."""
  sigma = max(0.0, (np.float64(n_in) / np.float64(n_out) - 1) / 2)
  if not sigma > 1e-15:
    return None
  radius = int(4.0 * float(sigma) + 0.5)
  if radius == 0:
    return None                                     # weights [1.0]: exact identity
  sigma2 = sigma * sigma
  x = np.arange(-radius, radius + 1)
  phi = np.exp(-0.5 / sigma2 * x ** 2)
  phi = phi / phi.sum()
  return phi[radius:]


def gaussian_prefilter_axis(a, half, axis):
  """scipy NI_Correlate1D, symmetric branch, mode 'mirror', on a float32 array: per output element
  t = in[0]*w[0]; for j = radius .. 1: t += (in[-j] + in[+j]) * w[j]  (f64), rounded to f32."""
  a = np.moveaxis(np.asarray(a, np.float32), axis, -1)
  n = a.shape[-1]
  r = len(half) - 1
  idx = np.array([[_mirror_index(i + j, n) for i in range(n)] for j in range(-r, r + 1)])   # [2r+1, n]
  d = a.astype(np.float64)
  t = d * half[0]
  for j in range(r, 0, -1):
    t = t + (d[..., idx[r - j]] + d[..., idx[r + j]]) * half[j]
  return np.moveaxis(t.astype(np.float32), -1, axis)


def resize_bilinear(obs, out_shape):
  """[..., h, w] -> [..., H, W]; leading dimensions are independent images.  When an axis shrinks,
  skimage >= 0.19 first applies its anti-aliasing Gaussian (scipy.ndimage.gaussian_filter, axis by
  axis, each pass rounded to the float32 input dtype), then the same order-1 zoom."""
  obs = np.asarray(obs)
  H, W = out_shape
  lo = obs.min(axis=(-2, -1), keepdims=True)
  hi = obs.max(axis=(-2, -1), keepdims=True)
  if H < obs.shape[-2] or W < obs.shape[-1]:
    for axis, n_out in ((-2, H), (-1, W)):
      half = gaussian_half_kernel(obs.shape[axis], n_out)
      if half is not None:
        obs = gaussian_prefilter_axis(obs, half, axis)
  y0, y1, wy0, wy1 = axis_table(obs.shape[-2], H)
  x0, x1, wx0, wx1 = axis_table(obs.shape[-1], W)
  a = obs.astype(np.float64)
  wy0, wy1 = wy0[:, None], wy1[:, None]
  t = np.zeros(obs.shape[:-2] + (H, W))
  t = t + (a[..., y0[:, None], x0[None, :]] * wy0) * wx0
  t = t + (a[..., y0[:, None], x1[None, :]] * wy0) * wx1
  t = t + (a[..., y1[:, None], x0[None, :]] * wy1) * wx0
  t = t + (a[..., y1[:, None], x1[None, :]] * wy1) * wx1
  out = t.astype(obs.dtype)
  return np.clip(out, lo, hi)                    # skimage's clip=True (range of the ORIGINAL image); a no-op:
                                                 # every stage is a convex combination rounded to nearest


def small_state_to_image(shape, obs):
  """wrappers.py:178-204; obs [..., size] with size <= 4 -> [..., *shape]."""
  obs = np.asarray(obs)
  size = obs.shape[-1]
  lead = obs.shape[:-1]
  result = np.empty(lead + tuple(shape), dtype=obs.dtype)
  r = result.reshape((-1,) + tuple(shape))                     # view: [lanes, *shape]
  f = obs.reshape((-1, size))
  v = lambda j: f[:, j].reshape((-1,) + (1,) * len(shape))
  if size == 1:
    r[:] = v(0)
  elif size == 2:
    r[:, :, :shape[1] // 2] = v(0)
    r[:, :, shape[1] // 2:] = v(1)
  elif size in (3, 4):
    r[:, :shape[0] // 2, :shape[1] // 2] = v(0)               # "top-left"
    r[:, shape[0] // 2:, :shape[1] // 2] = v(1)               # labelled top-right, is bottom-left (:193-194)
    r[:, :shape[0] // 2, shape[1] // 2:] = v(2)
    r[:, shape[0] // 2:, shape[1] // 2:] = v(size - 1)        # flattened[-1]
  else:
    raise ValueError('Hand-crafted rule only for small state observation.')
  return result


def to_image(shape, obs, batched=False):
  """wrappers.py:222-247.  batched=True: obs carries one leading lane dimension."""
  shape = tuple(shape)
  assert len(shape) >= 2
  obs = np.asarray(obs)
  one = obs[0] if batched else obs
  if one.size <= 4:
    flat = obs.reshape((obs.shape[0], -1)) if batched else obs.reshape(-1)
    return small_state_to_image(shape, flat)
  if one.ndim <= 2:
    if one.ndim == 1:
      obs = np.expand_dims(obs, -2)
    plane = resize_bilinear(obs, shape[:2])
    while plane.ndim - (1 if batched else 0) < len(shape):
      plane = np.expand_dims(plane, -1)
    return np.broadcast_to(plane, ((obs.shape[0],) if batched else ()) + shape).copy()
  raise ValueError('Cannot convert observation shape {} to desired shape {}'.format(one.shape, shape))
