#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// no-loop: block b writes chunks [b*K*256, (b+1)*K*256), K lane-interleaved stores per thread
template <int K>
__global__ void __launch_bounds__(256) fill_noloop(f4* __restrict__ p, int64_t n16) {
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  int64_t i = (int64_t)blockIdx.x * (K * 256) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) { int64_t j = i + k * 256; if (j < n16) p[j] = z; }
}
// same but 32-bit index math (torch style: int idx)
template <int K>
__global__ void __launch_bounds__(256) fill_noloop32(f4* __restrict__ p, uint32_t n16) {
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  uint32_t i = blockIdx.x * (K * 256) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) { uint32_t j = i + k * 256; if (j < n16) p[j] = z; }
}
// values computed (not constant zero): emulate the one-hot compare cost
template <int K>
__global__ void __launch_bounds__(256) fill_noloop_hot(f4* __restrict__ p, uint32_t n16, uint32_t magic, const int* __restrict__ hot) {
  uint32_t i = blockIdx.x * (K * 256) + threadIdx.x;
  uint32_t l[K]; int h[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { uint32_t j = i + k * 256; l[k] = __umulhi(j << 2, magic); h[k] = (j < n16) ? hot[l[k] & 0xFFFFF] : -1; }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    uint32_t j = i + k * 256;
    if (j < n16) { int d = h[k] - (int)((j << 2) - l[k] * 900u); f4 v = {d == 0 ? 1.f : 0.f, d == 1 ? 1.f : 0.f, d == 2 ? 1.f : 0.f, d == 3 ? 1.f : 0.f}; p[j] = v; }
  }
}
extern "C" int calib4(void* ptr, int64_t nbytes, int variant, const int* hot, void* stream) {
  hipStream_t st = (hipStream_t)stream; int64_t n16 = nbytes / 16; f4* p = (f4*)ptr;
#define G(K) dim3((unsigned)((n16 + (K)*256 - 1) / ((K)*256)))
  switch (variant) {
    case 0: fill_noloop<1><<<G(1), dim3(256), 0, st>>>(p, n16); break;
    case 1: fill_noloop<2><<<G(2), dim3(256), 0, st>>>(p, n16); break;
    case 2: fill_noloop<4><<<G(4), dim3(256), 0, st>>>(p, n16); break;
    case 3: fill_noloop<8><<<G(8), dim3(256), 0, st>>>(p, n16); break;
    case 4: fill_noloop32<2><<<G(2), dim3(256), 0, st>>>(p, (uint32_t)n16); break;
    case 5: fill_noloop32<4><<<G(4), dim3(256), 0, st>>>(p, (uint32_t)n16); break;
    case 6: fill_noloop_hot<2><<<G(2), dim3(256), 0, st>>>(p, (uint32_t)n16, (uint32_t)((0x100000000ull / 900) + 1), hot); break;
    case 7: fill_noloop_hot<4><<<G(4), dim3(256), 0, st>>>(p, (uint32_t)n16, (uint32_t)((0x100000000ull / 900) + 1), hot); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
