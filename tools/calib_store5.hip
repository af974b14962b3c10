// Store cache-policy bits on gfx950: global_store_dwordx4 with sc0 / sc1 / nt combinations (inline asm),
// no-loop fill shape (K lane-interleaved 16-byte stores per thread).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

#define STORE_VARIANT(NAME, BITS)                                                                  \
  template <int K>                                                                                 \
  __global__ void __launch_bounds__(256) NAME(f4* __restrict__ p, int64_t n16) {                    \
    const f4 z = {0.f, 0.f, 0.f, 0.f};                                                              \
    int64_t i = (int64_t)blockIdx.x * (K * 256) + threadIdx.x;                                      \
    _Pragma("unroll") for (int k = 0; k < K; ++k) {                                                 \
      int64_t j = i + k * 256;                                                                      \
      if (j < n16) { f4* q = p + j; asm volatile("global_store_dwordx4 %0, %1, off " BITS :: "v"(q), "v"(z) : "memory"); } \
    }                                                                                               \
  }
STORE_VARIANT(fill_plain, "")
STORE_VARIANT(fill_sc0, "sc0")
STORE_VARIANT(fill_sc1, "sc1")
STORE_VARIANT(fill_sc0sc1, "sc0 sc1")
STORE_VARIANT(fill_nt, "nt")
STORE_VARIANT(fill_sc0nt, "sc0 nt")
STORE_VARIANT(fill_sc1nt, "sc1 nt")
STORE_VARIANT(fill_all, "sc0 sc1 nt")

extern "C" int calib5(void* ptr, int64_t nbytes, int variant, void* stream) {
  hipStream_t st = (hipStream_t)stream; int64_t n16 = nbytes / 16; f4* p = (f4*)ptr;
  const dim3 g((unsigned)((n16 + 4 * 256 - 1) / (4 * 256))), b(256);
  switch (variant) {
    case 0: fill_plain<4><<<g, b, 0, st>>>(p, n16); break;
    case 1: fill_sc0<4><<<g, b, 0, st>>>(p, n16); break;
    case 2: fill_sc1<4><<<g, b, 0, st>>>(p, n16); break;
    case 3: fill_sc0sc1<4><<<g, b, 0, st>>>(p, n16); break;
    case 4: fill_nt<4><<<g, b, 0, st>>>(p, n16); break;
    case 5: fill_sc0nt<4><<<g, b, 0, st>>>(p, n16); break;
    case 6: fill_sc1nt<4><<<g, b, 0, st>>>(p, n16); break;
    case 7: fill_all<4><<<g, b, 0, st>>>(p, n16); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
