"""ORACLE (test infrastructure, never imported by the product): numpy restatement of
`to_image` — bsuite/utils/wrappers.py:178-247.

`_small_state_to_image` (:178-204) is restated literally.  `_interpolate_to_image` (:207-219)
calls skimage.transform.resize, a third-party dependency that is NOT under /root/reference and not
installed (setup.py lists `scikit-image` unpinned); for scikit-image >= 0.19 and an output no smaller
than the input that call is scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True), whose
arithmetic `resize_bilinear` restates operation by operation in f64 (ni_interpolation.c
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


def resize_bilinear(obs, out_shape):
  """[..., h, w] -> [..., H, W]; leading dimensions are independent images."""
  obs = np.asarray(obs)
  H, W = out_shape
  if H < obs.shape[-2] or W < obs.shape[-1]:
    raise NotImplementedError('down-scaling needs skimage\'s anti-aliasing filter')
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
  lo = obs.min(axis=(-2, -1), keepdims=True)
  hi = obs.max(axis=(-2, -1), keepdims=True)
  return np.clip(out, lo, hi)                    # skimage's clip=True; a no-op after the f32 cast


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
