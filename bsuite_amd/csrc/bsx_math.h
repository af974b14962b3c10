// bsx_math.h — f32 sine/cosine for the physics families (cartpole, swing-up, mountain_car).
//
// The reference evaluates np.sin/np.cos in f64 (bsuite/environments/cartpole.py:44-45,173-174,
// mountain_car.py:79); the contract here is |a-b| <= 1e-6*max(1,|b|) per step (BASELINE north_star).
// The library sinf/cosf carry a generic range reduction (Payne-Hanek for huge arguments) and are
// evaluated separately; the angles on this path are small — theta lives in [0, 2*pi) after
// np.remainder, 3*position in [-3.6, 1.8] — so one shared Cody-Waite reduction by pi/2 plus two
// short polynomials gives both values in ~30 VALU instructions instead of ~90, with an absolute
// error below 1.2e-7 for |x| <= 64 (tests/test_physics_math.py checks it on the host against libm).
//
// Plain C99 + BSX_HD so the CPU tests compile the very same code with gcc.
#ifndef BSX_MATH_H_
#define BSX_MATH_H_

#include "../../include/bsx_stream.h"   // BSX_HD

#if defined(__HIPCC__)
#define BSX_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define BSX_RINTF(x) __builtin_rintf(x)
#define BSX_FABSF(x) __builtin_fabsf(x)
#else
#include <math.h>
#define BSX_FMAF(a, b, c) fmaf((a), (b), (c))
#define BSX_RINTF(x) rintf(x)
#define BSX_FABSF(x) fabsf(x)
#endif

/* np.float32(np.int8(b)) / np.float32(255) — a pixel of the MNIST bandit's observation (bsuite/utils/datasets.py:55-56 parses
 * the idx bytes as int8, mnist.py:64 divides by 255 in f32) — correctly rounded, without a table and without a division:
 * q = x * RN(1/255), then one Newton step on the exact remainder (two fused multiply-adds).  Equal to the IEEE division
 * for all 256 bytes (tests/test_physics_math.py, against numpy): a 16-byte chunk of the observation stream as four
 * {v_bfe_i32, v_cvt, 3 VALU} instead of four LDS reads behind a table fill and a workgroup barrier.  Measured in round 5
 * and NOT adopted (the stream is slower with it, profiles/r05/ab_mnist_arith.log): compiled into the tuning build only
 * (BSX_MNIST_ARITH=1); the host side uses it to recognise the reference's table.
 * four_pixels: one aligned dword of the image table; k = 0..3: which byte. */
BSX_HD float bsx_mnist_pixel_value(uint32_t four_pixels, int k) {
  const float x = (float)((int32_t)(four_pixels << (24 - 8 * k)) >> 24);
  const float r = 0x1.010102p-8f;                                  /* RN(1/255) = 0x3b808081 */
  const float q = x * r;
  const float e = BSX_FMAF(-q, 255.0f, x);
  return BSX_FMAF(e, r, q);
}

#define BSX_SINCOS_MAX_ARG 64.0f   /* beyond this callers must use the library routines */

/* sin and cos of x for |x| <= BSX_SINCOS_MAX_ARG.
 * Reduction: k = rint(x * 2/pi), r = x - k*pi/2 with pi/2 = HI + LO; the first fma is exact (k*HI
 * and x are both multiples of 2^-23 in this range and |r| < 1), the second rounds once.
 * Kernels on [-pi/4, pi/4]: the minimax forms of FreeBSD msun's k_sinf.c / k_cosf.c (degree 9 / 8). */
BSX_HD void bsx_sincosf(float x, float* s, float* c) {
  const float k = BSX_RINTF(x * 0.63661977236758134308f);
  float r = BSX_FMAF(-k, 1.57079637050628662109375f, x);
  r = BSX_FMAF(-k, -4.37113900018624283e-8f, r);
  const float z = r * r;
  /* sin(r) = r + r*z*(S1 + z*(S2 + z*(S3 + z*S4))) */
  float ps = BSX_FMAF(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f);
  ps = BSX_FMAF(z, ps, 8.3333293858894631756e-3f);
  ps = BSX_FMAF(z, ps, -1.66666666416265235595e-1f);
  const float sr = BSX_FMAF(r * z, ps, r);
  /* cos(r) = 1 + z*(C0 + z*(C1 + z*(C2 + z*C3))) */
  float pc = BSX_FMAF(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f);
  pc = BSX_FMAF(z, pc, 4.16666233237390631894e-2f);
  pc = BSX_FMAF(z, pc, -4.99999997251031003120e-1f);
  const float cr = BSX_FMAF(z, pc, 1.0f);
  const int q = (int)k;                      /* quadrant: x = r + q*pi/2 */
  const float ss = (q & 1) ? cr : sr;
  const float cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

/* sin(t + d), cos(t + d) from s0 = sin t, c0 = cos t for a small step |d| <= 0.5 (cartpole's
 * dt*theta_dot): sin d and cos d by Taylor series to d^7 / d^8 (truncation < 6e-9 at |d| = 0.5),
 * then the angle-addition formulas.  Cheaper than a second bsx_sincosf and closer to the reference,
 * which evaluates sin/cos of the un-rounded f64 angle. */
BSX_HD void bsx_sincos_advance(float s0, float c0, float d, float* s1, float* c1) {
  const float z = d * d;
  float ps = BSX_FMAF(z, -1.98412698412698413e-4f, 8.33333333333333333e-3f);
  ps = BSX_FMAF(z, ps, -1.66666666666666667e-1f);
  const float sd = BSX_FMAF(d * z, ps, d);
  float pc = BSX_FMAF(z, 2.48015873015873016e-5f, -1.38888888888888889e-3f);
  pc = BSX_FMAF(z, pc, 4.16666666666666667e-2f);
  pc = BSX_FMAF(z, pc, -0.5f);
  const float cdm1 = z * pc;                 /* cos d - 1 */
  *s1 = BSX_FMAF(s0, cdm1, BSX_FMAF(c0, sd, s0));
  *c1 = BSX_FMAF(c0, cdm1, BSX_FMAF(-s0, sd, c0));
}

#endif  /* BSX_MATH_H_ */
