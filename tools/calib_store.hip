// Store-bandwidth calibration for MI355X (gfx950): how fast can a pure store stream go?
// deep_sea at N=30 is 99% observation stores, so this number is the practical roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float float4v __attribute__((ext_vector_type(4)));

// variant 0: grid-stride plain 16-B stores
__global__ void __launch_bounds__(256) fill_gs(float4v* __restrict__ p, int64_t n16) {
  float4v z = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = z;
}
// variant 1: grid-stride nontemporal 16-B stores
__global__ void __launch_bounds__(256) fill_gs_nt(float4v* __restrict__ p, int64_t n16) {
  float4v z = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    __builtin_nontemporal_store(z, &p[i]);
}
// variant 2/3: each block owns one contiguous tile of tile16 16-B chunks (deep_sea pattern)
template <bool NT>
__global__ void __launch_bounds__(256) fill_tile(float4v* __restrict__ p, int64_t n16, int64_t tile16) {
  float4v z = {0.f, 0.f, 0.f, 0.f};
  int64_t base = (int64_t)blockIdx.x * tile16;
  int64_t end = base + tile16; if (end > n16) end = n16;
  for (int64_t i = base + threadIdx.x; i < end; i += 256) {
    if (NT) __builtin_nontemporal_store(z, &p[i]); else p[i] = z;
  }
}
// variant 4: tile with 4x unrolled independent stores
__global__ void __launch_bounds__(256) fill_tile_u4(float4v* __restrict__ p, int64_t n16, int64_t tile16) {
  float4v z = {0.f, 0.f, 0.f, 0.f};
  int64_t base = (int64_t)blockIdx.x * tile16;
  int64_t end = base + tile16; if (end > n16) end = n16;
  int64_t i = base + threadIdx.x;
  for (; i + 768 < end; i += 1024) { p[i] = z; p[i + 256] = z; p[i + 512] = z; p[i + 768] = z; }
  for (; i < end; i += 256) p[i] = z;
}
// variant 5: copy (read+write) for reference against the guide's 6.29 TB/s
__global__ void __launch_bounds__(256) copy_gs(const float4v* __restrict__ s, float4v* __restrict__ d, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}

extern "C" int calib_fill(void* ptr, int64_t nbytes, int variant, int64_t grid, int64_t tile_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int64_t n16 = nbytes / 16;
  float4v* p = (float4v*)ptr;
  int64_t tile16 = tile_bytes / 16;
  switch (variant) {
    case 0: fill_gs<<<dim3((unsigned)grid), dim3(256), 0, st>>>(p, n16); break;
    case 1: fill_gs_nt<<<dim3((unsigned)grid), dim3(256), 0, st>>>(p, n16); break;
    case 2: fill_tile<false><<<dim3((unsigned)((n16 + tile16 - 1) / tile16)), dim3(256), 0, st>>>(p, n16, tile16); break;
    case 3: fill_tile<true><<<dim3((unsigned)((n16 + tile16 - 1) / tile16)), dim3(256), 0, st>>>(p, n16, tile16); break;
    case 4: fill_tile_u4<<<dim3((unsigned)((n16 + tile16 - 1) / tile16)), dim3(256), 0, st>>>(p, n16, tile16); break;
    case 5: copy_gs<<<dim3((unsigned)grid), dim3(256), 0, st>>>(p, p + n16 / 2, n16 / 2); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
