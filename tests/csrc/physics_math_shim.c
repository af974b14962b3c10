/* Test shim: exposes the f32 sine/cosine of bsuite_amd/csrc/bsx_math.h (the header the HIP kernels
 * compile) to ctypes, evaluated on the host by gcc. */
#include <stdint.h>
#include "../../bsuite_amd/csrc/bsx_math.h"

void shim_sincosf(const float* x, int64_t n, float* s, float* c) {
  for (int64_t i = 0; i < n; ++i) bsx_sincosf(x[i], &s[i], &c[i]);
}

/* sin/cos of (t + d) by one bsx_sincosf(t) followed by bsx_sincos_advance(d) */
void shim_sincos_advance(const float* t, const float* d, int64_t n, float* s, float* c) {
  for (int64_t i = 0; i < n; ++i) {
    float s0, c0;
    bsx_sincosf(t[i], &s0, &c0);
    bsx_sincos_advance(s0, c0, d[i], &s[i], &c[i]);
  }
}

/* the MNIST bandit's pixel value for every byte b = 0..255, in each of the four byte positions of a dword */
void shim_mnist_pixels(float* out /* [4][256] */) {
  for (int k = 0; k < 4; ++k)
    for (uint32_t b = 0; b < 256u; ++b) out[k * 256 + b] = bsx_mnist_pixel_value((b << (8 * k)) | (0xA5A5A5A5u & ~(0xFFu << (8 * k))), k);
}
