"""CPU: pin the oracle's draw stream.  Philox4x32-10 against the published Random123 known-answer
vectors; the normal transform against scipy's ndtri; numpy and C restatements against each other."""
import numpy as np

from oracle import coracle
from oracle import stream as S


def test_philox_known_answers():
  kat = [
      ([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
      ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
      ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0),
       [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
  ]
  for ctr, key, want in kat:
    got = S.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
    assert [int(x) for x in got] == want


def test_c_and_numpy_words_agree():
  for seed, lane, step, sid in [(42, 0, 0, 0), (2**63 - 1, 2**40 + 3, 2**47 + 9, 1), (1, 2, 3, 0)]:
    a = S.words(seed, [lane], step, sid, 300)[0]
    b = coracle.stream_words(seed, lane, step, sid, 300)
    np.testing.assert_array_equal(a, b)


def test_normal_transform():
  from scipy.special import ndtri
  rng = np.random.default_rng(0)
  k = rng.integers(0, 1 << 53, size=500000, dtype=np.uint64)
  edge = np.array([0, 1, 2, 100, (1 << 52) - 1, 1 << 52, (1 << 53) - 3, (1 << 53) - 2], np.uint64)
  k = np.concatenate([k, edge])
  z = S.normal_from_k53(k)
  np.testing.assert_array_equal(z.view(np.uint64), coracle.normals(k).view(np.uint64))
  # evaluate scipy on the small-tail side so its own input rounding (p near 1) does not dominate
  upper = k >= np.uint64(1 << 52)
  kk = np.where(upper, np.uint64((1 << 53) - 1) - k, k)
  p = (kk.astype(np.float64) + 0.5) * 2.0 ** -53
  ref = np.where(upper, -ndtri(p), ndtri(p))
  assert np.max(np.abs(z - ref) / np.maximum(1.0, np.abs(ref))) < 1e-13
  assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
  lo, hi = S.normal_from_k53(np.array([0, (1 << 53) - 1], np.uint64))
  assert lo == -hi and 8.2 < hi < 8.4


def test_lane_stream_draws():
  s = S.LaneStream(42, 7)
  s.begin_step(3)
  w = S.words(42, [7], 3, 0, 8)[0]
  assert s.bern() == int(w[0]) >> 31
  assert s.randint(5) == (int(w[1]) * 5) >> 32
  assert s.uniform01() == float((((int(w[2]) >> 5) << 26) | (int(w[3]) >> 6)) * 2.0 ** -53)
  v = s.bern_vec(33)
  assert v.shape == (33,) and v[0] == int(w[4]) & 1 and v[32] == int(w[5]) & 1
