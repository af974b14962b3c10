/* bsx_stream.h — "bsx stream v1": the counter-based random-draw stream of the engine.
 *
 * Why it exists: the reference gives every environment object its own np.random.RandomState
 * (MT19937, 2.5 KB of state; e.g. bsuite/environments/deep_sea.py:77, catch.py:58,
 * utils/wrappers.py:267).  At 2^20 lanes that is 2.6 GB of generator state and most sweep seeds
 * are None (not reproducible in the reference either).  Here a lane's draws are a pure function of
 *      (seed, global lane id, step_index, stream_id, word#)
 * so there is no generator state in HBM, shards of any size produce identical trajectories, and a
 * host can replay any lane.  Parity with the reference is defined as "same dynamics given the same
 * draws": the oracle substitutes a replay of this stream for `env._rng` (oracle/replay.py).
 *
 * Specification
 *   generator : Philox4x32-10 (Salmon et al., SC'11), key = (seed[31:0], seed[63:32]),
 *               counter = (lane[31:0], lane[63:32], step[31:0],
 *                          step[47:32]<<16 | stream_id<<8 | block)
 *               word w of a (lane, step, stream) triple = output word (w & 3) of block (w >> 2).
 *   stream_id : 0 = environment dynamics (`env._rng`), 1 = reward wrapper (`RewardNoise._rng`).
 *   draws consume words in order:
 *     U()        2 words a,b : k = (a>>5)<<26 | (b>>6) ; U = k * 2^-53   (numpy legacy rand())
 *     Bern()     1 word      : word >> 31                                 (binomial(1, .5))
 *     BernVec(n) ceil(n/32) words : element i = (word[i/32] >> (i%32)) & 1
 *     RandInt(n) 1 word      : (uint64(word) * n) >> 32                   (randint(n))
 *     Normal()   2 words     : k as in U(); q = (2*(k-2^52)+1) * 2^-54 in (-.5,.5) exactly;
 *                              z = Phi^-1(q + .5) by Wichura's AS241 PPND16 rational forms,
 *                              evaluated with +,-,*,/,sqrt and the bit-level log below only,
 *                              no fused multiply-add, so host and device agree bit for bit.
 *
 * This header is plain C99 + the BSX_HD qualifier so the HIP kernels and host code share one
 * definition.  The oracle does NOT include it: oracle/stream.py and oracle/oracle.c restate the
 * specification independently and the tests compare all three.
 */
#ifndef BSX_STREAM_H_
#define BSX_STREAM_H_

#include <stdint.h>

#include "bsx_libm_log.h"

#if defined(__HIPCC__)
#define BSX_HD __host__ __device__ __forceinline__
#else
#define BSX_HD static inline
#endif

#define BSX_STREAM_ENV 0u
#define BSX_STREAM_WRAP 1u

typedef struct { uint32_t v[4]; } bsx_u32x4;

BSX_HD uint32_t bsx_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

BSX_HD bsx_u32x4 bsx_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                   uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = bsx_mulhi32(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = bsx_mulhi32(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  bsx_u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

/* Sequential word reader for one (lane, step, stream) triple. */
typedef struct {
  uint32_t c0, c1, c2, c3hi, k0, k1;
  uint32_t next;      /* index of the next word to hand out */
  int32_t have;       /* block currently cached in `blk` (-1 none) */
  bsx_u32x4 blk;
  /* MT19937-exact mode (see below): NULL = counter-based stream */
  uint32_t* mt;       /* element k of this lane's 624-word state lives at mt[k * mt_stride]  */
  int64_t mt_stride;
  int32_t mt_pos;     /* numpy's `pos`: 0..624 */
  int32_t mt_has_gauss; /* RandomState's cached second value of the polar pair */
  double mt_gauss;
} bsx_draws;

BSX_HD void bsx_draws_init(bsx_draws* d, uint64_t seed, uint64_t lane, uint64_t step, uint32_t stream_id) {
  d->k0 = (uint32_t)seed; d->k1 = (uint32_t)(seed >> 32);
  d->c0 = (uint32_t)lane; d->c1 = (uint32_t)(lane >> 32);
  d->c2 = (uint32_t)step;
  d->c3hi = (((uint32_t)(step >> 32) & 0xFFFFu) << 16) | ((stream_id & 0xFFu) << 8);
  d->next = 0; d->have = -1;
  d->mt = 0; d->mt_stride = 0; d->mt_pos = 0; d->mt_has_gauss = 0; d->mt_gauss = 0.0;
}

/* ---- MT19937-exact mode ("mode B") ------------------------------------------------------------
 * For seeded small batches the engine can instead carry, per lane, the very generator the
 * reference uses — np.random.RandomState = MT19937 (624-word state + position) with numpy's LEGACY
 * samplers — so that e.g. Catch(seed=0) here and in the reference produce the same trajectory with
 * no replay shim.  The host builds each lane's initial state with numpy itself
 * (RandomState(seed).get_state() after replaying the constructor's draws).  Samplers, restated
 * from numpy's legacy distributions (verified against numpy 2.2 RandomState in the tests):
 *   next_u32     genrand_int32 (Matsumoto-Nishimura reference algorithm, tempering included)
 *   U()          a = next>>5, b = next>>6 ; (a*2^26 + b) / 2^53            (random_sample / rand)
 *   uniform      lo + (hi-lo)*U()
 *   binomial(1,.5)          one U() per draw: int(U > 0.5)  (inversion algorithm with n=1, p=.5)
 *   binomial(1,.5,size=n)   n sequential draws
 *   randint(n)   n==1: 0 without a draw; else mask = next_pow2(n-1)-1, reject next_u32&mask > n-1
 *   randn        legacy_gauss: polar Box-Muller, x = 2U-1 pairs until 0 < r2 < 1,
 *                f = sqrt(-2*log(r2)/r2), returns f*x2 and caches f*x1 (`has_gauss`); `log` is the
 *                host libm's, restated bit for bit in include/bsx_libm_log.h; sqrt and / are IEEE.   */
BSX_HD void bsx_mt_twist(uint32_t* mt, int64_t st) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
  int kk;
  uint32_t y;
  for (kk = 0; kk < 624 - 397; kk++) {
    y = (mt[kk * st] & UPPER) | (mt[(kk + 1) * st] & LOWER);
    mt[kk * st] = mt[(kk + 397) * st] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
  }
  for (; kk < 623; kk++) {
    y = (mt[kk * st] & UPPER) | (mt[(kk + 1) * st] & LOWER);
    mt[kk * st] = mt[(kk + (397 - 624)) * st] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
  }
  y = (mt[623 * st] & UPPER) | (mt[0] & LOWER);
  mt[623 * st] = mt[396 * st] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
}

BSX_HD uint32_t bsx_mt_next(bsx_draws* d) {
  if (d->mt_pos >= 624) { bsx_mt_twist(d->mt, d->mt_stride); d->mt_pos = 0; }
  uint32_t y = d->mt[(int64_t)d->mt_pos * d->mt_stride];
  d->mt_pos++;
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

BSX_HD uint32_t bsx_word(bsx_draws* d) {
  if (d->mt) return bsx_mt_next(d);
  uint32_t w = d->next++;
  int32_t b = (int32_t)(w >> 2);
  if (b != d->have) {
    d->blk = bsx_philox4x32_10(d->c0, d->c1, d->c2, d->c3hi | (uint32_t)b, d->k0, d->k1);
    d->have = b;
  }
  uint32_t i = w & 3u;
  return i == 0 ? d->blk.v[0] : i == 1 ? d->blk.v[1] : i == 2 ? d->blk.v[2] : d->blk.v[3];
}

/* Computes block 0 of the stream NOW (counter-based mode; a no-op in MT19937-exact mode).  For a family whose every path
 * draws from block 0 — umbrella_chain: the reset, a MID step and a LAST step alike — calling this before the paths diverge
 * makes a wave whose lanes are at different episode phases walk the ten Philox rounds once instead of once per path.  The
 * words handed out afterwards are the same words. */
BSX_HD void bsx_draws_prime(bsx_draws* d) {
  if (d->mt) return;
  d->blk = bsx_philox4x32_10(d->c0, d->c1, d->c2, d->c3hi, d->k0, d->k1);
  d->have = 0;
}

BSX_HD uint64_t bsx_k53(bsx_draws* d) {
  uint32_t a = bsx_word(d), b = bsx_word(d);
  return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
}
BSX_HD double bsx_uniform(bsx_draws* d) { return (double)bsx_k53(d) * 0x1p-53; }
BSX_HD uint32_t bsx_bern(bsx_draws* d) {
  if (d->mt) return bsx_uniform(d) > 0.5 ? 1u : 0u;       /* legacy binomial(1, .5): one double */
  return bsx_word(d) >> 31;
}
BSX_HD uint32_t bsx_randint(bsx_draws* d, uint32_t n) {
  if (d->mt) {                                            /* legacy masked rejection on uint32 */
    uint32_t rng = n - 1u, mask, v;
    if (rng == 0u) return 0u;
    mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    do { v = bsx_mt_next(d) & mask; } while (v > rng);
    return v;
  }
  return (uint32_t)(((uint64_t)bsx_word(d) * (uint64_t)n) >> 32);
}
/* Element b of a BernVec(n) being drawn in order b = 0, 1, ...; *w carries the current word. */
BSX_HD uint32_t bsx_bern_vec_bit(bsx_draws* d, int b, uint32_t* w) {
  if (d->mt) return bsx_uniform(d) > 0.5 ? 1u : 0u;
  if ((b & 31) == 0) *w = bsx_word(d);
  return (*w >> (b & 31)) & 1u;
}

/* ---- bit-reproducible natural log for normal doubles in (0, 1] --------------------------- */
BSX_HD double bsx_bits_to_f64(uint64_t u) { union { uint64_t u; double d; } x; x.u = u; return x.d; }
BSX_HD uint64_t bsx_f64_to_bits(double d) { union { uint64_t u; double d; } x; x.d = d; return x.u; }

#if defined(__clang__)
#define BSX_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define BSX_NO_CONTRACT
#endif

/* A floating-point constant of the polynomials below, materialised where it is used (device: an SGPR pair set by
 * s_mov_b64 behind an opaque asm; host: the literal).  Left to itself the compiler hoists all ~60 of them out of the
 * step loop of a fused rollout into VGPR pairs — 120 vector registers, i.e. 2 waves per SIMD for every kernel that can
 * draw a RewardNoise normal (157-218 VGPRs in the NOISE = 1 rollout instantiations).  Same values, same
 * operations: no numerical effect. */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double bsx_k_(double c) { asm volatile("" : "+s"(c)); return c; }
#define BSX_K(c) bsx_k_(c)
#else
#define BSX_K(c) (c)
#endif

BSX_HD double bsx_log(double x) {
  BSX_NO_CONTRACT
  uint64_t u = bsx_f64_to_bits(x);
  int32_t e = (int32_t)((u >> 52) & 0x7FF) - 1023;
  double m = bsx_bits_to_f64((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }                             /* [~.707,1.414] */
  double s = (m - 1.0) / (m + 1.0);
  double s2 = s * s;
  double p = BSX_K(1.0 / 25.0);
  p = p * s2 + BSX_K(1.0 / 23.0);
  p = p * s2 + BSX_K(1.0 / 21.0);
  p = p * s2 + BSX_K(1.0 / 19.0);
  p = p * s2 + BSX_K(1.0 / 17.0);
  p = p * s2 + BSX_K(1.0 / 15.0);
  p = p * s2 + BSX_K(1.0 / 13.0);
  p = p * s2 + BSX_K(1.0 / 11.0);
  p = p * s2 + BSX_K(1.0 / 9.0);
  p = p * s2 + BSX_K(1.0 / 7.0);
  p = p * s2 + BSX_K(1.0 / 5.0);
  p = p * s2 + BSX_K(1.0 / 3.0);
  p = p * s2 + 1.0;
  double lm = 2.0 * s * p;
  return (double)e * 0.6931471805599453 + lm;
}

#if defined(__HIPCC__)
#define BSX_SQRT(x) __builtin_sqrt(x)
#else
#include <math.h>
#define BSX_SQRT(x) sqrt(x)
#endif

/* AS241 PPND16 on q = p - 1/2 (Wichura 1988), k = 53-bit draw. */
BSX_HD double bsx_normal_from_k53(uint64_t k) {
  BSX_NO_CONTRACT
  int64_t j = 2 * ((int64_t)k - (int64_t)(1ull << 52)) + 1;   /* odd, |j| < 2^53 */
  double q = (double)j * 0x1p-54;
  double aq = q < 0 ? -q : q;
  double val;
  if (aq <= 0.425) {
    double r = 0.180625 - q * q;
    double num = BSX_K(2.5090809287301226727e+3);
    num = num * r + BSX_K(3.3430575583588128105e+4);
    num = num * r + BSX_K(6.7265770927008700853e+4);
    num = num * r + BSX_K(4.5921953931549871457e+4);
    num = num * r + BSX_K(1.3731693765509461125e+4);
    num = num * r + BSX_K(1.9715909503065514427e+3);
    num = num * r + BSX_K(1.3314166789178437745e+2);
    num = num * r + BSX_K(3.3871328727963666080e+0);
    double den = BSX_K(5.2264952788528545610e+3);
    den = den * r + BSX_K(2.8729085735721942674e+4);
    den = den * r + BSX_K(3.9307895800092710610e+4);
    den = den * r + BSX_K(2.1213794301586595867e+4);
    den = den * r + BSX_K(5.3941960214247511077e+3);
    den = den * r + BSX_K(6.8718700749205790830e+2);
    den = den * r + BSX_K(4.2313330701600911252e+1);
    den = den * r + 1.0;
    return q * num / den;
  }
  double r = 0.5 - aq;              /* min(p, 1-p), exact, >= 2^-54 */
  r = BSX_SQRT(-bsx_log(r));
  if (r <= 5.0) {
    r = r - 1.6;
    double num = BSX_K(7.74545014278341407640e-4);
    num = num * r + BSX_K(2.27238449892691845833e-2);
    num = num * r + BSX_K(2.41780725177450611770e-1);
    num = num * r + BSX_K(1.27045825245236838258e+0);
    num = num * r + BSX_K(3.64784832476320460504e+0);
    num = num * r + BSX_K(5.76949722146069140550e+0);
    num = num * r + BSX_K(4.63033784615654529590e+0);
    num = num * r + BSX_K(1.42343711074968357734e+0);
    double den = BSX_K(1.05075007164441684324e-9);
    den = den * r + BSX_K(5.47593808499534494600e-4);
    den = den * r + BSX_K(1.51986665636164571966e-2);
    den = den * r + BSX_K(1.48103976427480074590e-1);
    den = den * r + BSX_K(6.89767334985100004550e-1);
    den = den * r + BSX_K(1.67638483018380384940e+0);
    den = den * r + BSX_K(2.05319162663775882187e+0);
    den = den * r + 1.0;
    val = num / den;
  } else {
    r = r - 5.0;
    double num = BSX_K(2.01033439929228813265e-7);
    num = num * r + BSX_K(2.71155556874348757815e-5);
    num = num * r + BSX_K(1.24266094738807843860e-3);
    num = num * r + BSX_K(2.65321895265761230930e-2);
    num = num * r + BSX_K(2.96560571828504891230e-1);
    num = num * r + BSX_K(1.78482653991729133580e+0);
    num = num * r + BSX_K(5.46378491116411436990e+0);
    num = num * r + BSX_K(6.65790464350110377720e+0);
    double den = BSX_K(2.04426310338993978564e-15);
    den = den * r + BSX_K(1.42151175831644588870e-7);
    den = den * r + BSX_K(1.84631831751005468180e-5);
    den = den * r + BSX_K(7.86869131145613259100e-4);
    den = den * r + BSX_K(1.48753612908506148525e-2);
    den = den * r + BSX_K(1.36929880922735805310e-1);
    den = den * r + BSX_K(5.99832206555887937690e-1);
    den = den * r + 1.0;
    val = num / den;
  }
  return q < 0 ? -val : val;
}
/* np.random.RandomState.randn() / standard_normal(): numpy/random/src/legacy/legacy-distributions.c
 * legacy_gauss, driven by the lane's MT19937 (used by the reference at utils/wrappers.py:278 and
 * environments/deep_sea.py:126). */
BSX_HD double bsx_mt_gauss(bsx_draws* d) {
  BSX_NO_CONTRACT
  if (d->mt_has_gauss) {
    const double t = d->mt_gauss;
    d->mt_has_gauss = 0;
    d->mt_gauss = 0.0;
    return t;
  }
  double x1, x2, r2;
  do {
    x1 = 2.0 * bsx_uniform(d) - 1.0;
    x2 = 2.0 * bsx_uniform(d) - 1.0;
    r2 = x1 * x1 + x2 * x2;
  } while (r2 >= 1.0 || r2 == 0.0);
  const double f = BSX_SQRT(-2.0 * bsx_libm_log(r2) / r2);
  d->mt_gauss = f * x1;
  d->mt_has_gauss = 1;
  return f * x2;
}
BSX_HD double bsx_normal(bsx_draws* d) {
  if (d->mt) return bsx_mt_gauss(d);
  return bsx_normal_from_k53(bsx_k53(d));
}

#endif /* BSX_STREAM_H_ */
