"""ORACLE (test infrastructure, never shipped / never on the product path).

Independent numpy restatement of the "bsx stream v1" draw stream specified in
include/bsx_stream.h (it deliberately does not share code with that header).  Philox4x32-10 is
pinned by the published Random123 known-answer vectors (tests/test_stream_oracle.py); the normal
transform (Wichura AS241 PPND16) is pinned against scipy.special.ndtri.

The reference's randomness sources this stands in for: np.random.RandomState members
rand / randn / binomial(1,.5) / randint / uniform used at deep_sea.py:126,130, catch.py:71,
memory_chain.py:94-95, umbrella_chain.py:65,83,89-90, cartpole.py:91-92, mountain_car.py:69,
utils/wrappers.py:278.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

STREAM_ENV = 0
STREAM_WRAP = 1


def philox4x32_10(ctr, key):
  """ctr: uint32 array [..., 4]; key: (k0, k1) python ints.  Returns uint32 [..., 4]."""
  c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
  k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
  for _ in range(10):
    p0 = M0 * c[0]
    p1 = M1 * c[2]
    hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
    hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
    c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
    k0 = (k0 + W0) & 0xFFFFFFFF
    k1 = (k1 + W1) & 0xFFFFFFFF
  return np.stack(c, axis=-1).astype(np.uint32)


def words(seed, lanes, step, stream_id, n_words):
  """uint32 [len(lanes), n_words]: the first n_words words of each lane's (step, stream) triple."""
  lanes = np.atleast_1d(np.asarray(lanes, dtype=np.uint64))
  n_blocks = (n_words + 3) // 4
  ctr = np.zeros((lanes.shape[0], n_blocks, 4), dtype=np.uint32)
  ctr[..., 0] = (lanes & MASK32).astype(np.uint32)[:, None]
  ctr[..., 1] = (lanes >> np.uint64(32)).astype(np.uint32)[:, None]
  ctr[..., 2] = np.uint32(step & 0xFFFFFFFF)
  hi = (((step >> 32) & 0xFFFF) << 16) | ((stream_id & 0xFF) << 8)
  ctr[..., 3] = (np.uint32(hi) | np.arange(n_blocks, dtype=np.uint32))[None, :]
  out = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
  return out.reshape(lanes.shape[0], n_blocks * 4)[:, :n_words]


def k53(a, b):
  a = np.asarray(a, dtype=np.uint64)
  b = np.asarray(b, dtype=np.uint64)
  return ((a >> np.uint64(5)) << np.uint64(26)) | (b >> np.uint64(6))


def _log(x):
  """Bit-level natural log restated from the spec: frexp to [~.707,1.414], atanh series."""
  x = np.asarray(x, dtype=np.float64)
  u = x.view(np.uint64)
  e = ((u >> np.uint64(52)) & np.uint64(0x7FF)).astype(np.int64) - 1023
  m = ((u & np.uint64(0x000FFFFFFFFFFFFF)) | np.uint64(0x3FF0000000000000)).view(np.float64)
  big = m > 1.4142135623730951
  m = np.where(big, m * 0.5, m)
  e = np.where(big, e + 1, e)
  s = (m - 1.0) / (m + 1.0)
  s2 = s * s
  p = np.full_like(s, 1.0 / 25.0)
  for d in (23.0, 21.0, 19.0, 17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
    p = p * s2 + 1.0 / d
  p = p * s2 + 1.0
  return e.astype(np.float64) * 0.6931471805599453 + 2.0 * s * p


_A = (3.3871328727963666080e+0, 1.3314166789178437745e+2, 1.9715909503065514427e+3,
      1.3731693765509461125e+4, 4.5921953931549871457e+4, 6.7265770927008700853e+4,
      3.3430575583588128105e+4, 2.5090809287301226727e+3)
_B = (1.0, 4.2313330701600911252e+1, 6.8718700749205790830e+2, 5.3941960214247511077e+3,
      2.1213794301586595867e+4, 3.9307895800092710610e+4, 2.8729085735721942674e+4,
      5.2264952788528545610e+3)
_C = (1.42343711074968357734e+0, 4.63033784615654529590e+0, 5.76949722146069140550e+0,
      3.64784832476320460504e+0, 1.27045825245236838258e+0, 2.41780725177450611770e-1,
      2.27238449892691845833e-2, 7.74545014278341407640e-4)
_D = (1.0, 2.05319162663775882187e+0, 1.67638483018380384940e+0, 6.89767334985100004550e-1,
      1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4,
      1.05075007164441684324e-9)
_E = (6.65790464350110377720e+0, 5.46378491116411436990e+0, 1.78482653991729133580e+0,
      2.96560571828504891230e-1, 2.65321895265761230930e-2, 1.24266094738807843860e-3,
      2.71155556874348757815e-5, 2.01033439929228813265e-7)
_F = (1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2,
      7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7,
      2.04426310338993978564e-15)


def _horner(coef, r):
  acc = np.full_like(r, coef[7])
  for i in range(6, -1, -1):
    acc = acc * r + coef[i]
  return acc


def normal_from_k53(k):
  k = np.atleast_1d(np.asarray(k, dtype=np.uint64))
  j = 2 * (k.astype(np.int64) - (1 << 52)) + 1
  q = j.astype(np.float64) * 2.0 ** -54
  aq = np.abs(q)
  with np.errstate(all='ignore'):
    r0 = 0.180625 - q * q
    central = q * _horner(_A, r0) / _horner(_B, r0)
    rt = np.where(aq > 0.425, 0.5 - aq, 0.25)
    r = np.sqrt(-_log(rt))
    mid = _horner(_C, r - 1.6) / _horner(_D, r - 1.6)
    far = _horner(_E, r - 5.0) / _horner(_F, r - 5.0)
    tail = np.where(r <= 5.0, mid, far)
    tail = np.where(q < 0, -tail, tail)
  return np.where(aq <= 0.425, central, tail)


class LaneStream:
  """Sequential reader of one lane's words for one (step, stream) triple — scalar, for replay."""

  def __init__(self, seed, lane, stream_id=STREAM_ENV):
    self.seed, self.lane, self.stream_id = int(seed), int(lane), int(stream_id)
    self.begin_step(0)

  def begin_step(self, step):
    self.step = int(step)
    self._buf = np.zeros(0, dtype=np.uint32)
    self.next = 0

  def word(self):
    if self.next >= self._buf.shape[0]:
      n = max(16, 2 * self._buf.shape[0])
      self._buf = words(self.seed, [self.lane], self.step, self.stream_id, n)[0]
    w = int(self._buf[self.next])
    self.next += 1
    return w

  def k53(self):
    a = self.word()
    b = self.word()
    return ((a >> 5) << 26) | (b >> 6)

  def uniform01(self):
    return self.k53() * 2.0 ** -53

  def bern(self):
    return self.word() >> 31

  def bern_vec(self, n):
    ws = [self.word() for _ in range((n + 31) // 32)]
    return np.array([(ws[i // 32] >> (i % 32)) & 1 for i in range(n)], dtype=np.int64)

  def randint(self, n):
    return (self.word() * int(n)) >> 32

  def normal(self):
    return float(normal_from_k53(np.uint64(self.k53()))[0])
