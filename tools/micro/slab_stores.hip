// What does the [T, B, ...] output of a fused rollout cost against a linear fill — and which property of the pattern
// is to blame?  N workgroups each own a piece of P bytes in each of T slabs (slab t = the [B, ...] slice of step t) and
// write their piece of slab 0, then of slab 1, ... with 16-byte stores, F FMAs per thread between two slabs (the
// step's arithmetic).  Variants:
//   blocked   workgroup b writes T*P contiguous bytes (the [B/256, T, 256, ...] layout: a linear stream per workgroup)
//   slab      piece b of every slab (the rollout kernels of r03)
//   slab-xcd  the same with an XCD-contiguous block map: workgroups are dealt to the 8 XCDs round-robin by blockIdx, so
//             b' = (b % 8) * (N / 8) + b / 8 hands every XCD (its L2, its address translation) one contiguous eighth
//             of each slab instead of every eighth piece of all of it
// each with 256- and 1024-thread workgroups (P = 6 KiB / 24 KiB for 24-byte rows), T = 16.
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/slab_stores.hip -o tools/ab/slab_stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int BS, int MODE, int F>
__global__ void __launch_bounds__(BS) k(float4* __restrict__ out, int T, int chunks_per_piece, size_t slab_chunks, float seed) {
  float r0 = seed + threadIdx.x, r1 = r0 * 1.5f, r2 = r0 + 2.f, r3 = r0 - 3.f;
  const float m = 1.0000001f, c = 1e-9f * seed;
  uint32_t b = blockIdx.x;
  if (MODE == 2) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int f = 0; f < F / 4; ++f) {
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(m), "v"(c));
    }
    float4* piece = MODE == 0 ? out + ((size_t)b * T + t) * chunks_per_piece : out + (size_t)t * slab_chunks + (size_t)b * chunks_per_piece;
    for (int ch = threadIdx.x; ch < chunks_per_piece; ch += BS) piece[ch] = make_float4(r0, r1, r2, r3);
  }
}

template <int BS, int MODE, int F>
static double run(float4* out, int T, size_t lanes, int row_bytes) {
  const int chunks_per_piece = BS * row_bytes / 16;
  const unsigned n_wg = (unsigned)(lanes / BS);
  const size_t slab_chunks = lanes * row_bytes / 16;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<BS, MODE, F><<<n_wg, BS>>>(out, T, chunks_per_piece, slab_chunks, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) k<BS, MODE, F><<<n_wg, BS>>>(out, T, chunks_per_piece, slab_chunks, 1.0f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 10 * 1e3;
}

template <int F>
static void table(float4* out, int T, size_t lanes, int row_bytes) {
  const double mb = (double)lanes * row_bytes * T / 1e6;
  printf("rows of %d B, %zu lanes, T = %d (%.0f MB per launch), %d FMAs per thread and step\n", row_bytes, lanes, T, mb, F);
  const char* names[3] = {"blocked", "slab", "slab-xcd"};
  double us[2][3];
  us[0][0] = run<256, 0, F>(out, T, lanes, row_bytes); us[0][1] = run<256, 1, F>(out, T, lanes, row_bytes); us[0][2] = run<256, 2, F>(out, T, lanes, row_bytes);
  us[1][0] = run<1024, 0, F>(out, T, lanes, row_bytes); us[1][1] = run<1024, 1, F>(out, T, lanes, row_bytes); us[1][2] = run<1024, 2, F>(out, T, lanes, row_bytes);
  for (int w = 0; w < 2; ++w)
    for (int m = 0; m < 3; ++m)
      printf("  %4d-thread workgroups, %-8s  %8.1f us  %5.2f TB/s  (%5.2f us per step)\n", w ? 1024 : 256, names[m], us[w][m], mb / us[w][m], us[w][m] / T);
}

int main() {
  float4* out;
  const size_t lanes = (size_t)1 << 20;
  hipMalloc(&out, lanes * 32 * 64);                          // 2 GiB: 32-byte rows, T = 64
  hipMemset(out, 0, lanes * 32 * 64);
  table<0>(out, 16, lanes, 24);
  table<128>(out, 16, lanes, 24);
  table<512>(out, 16, lanes, 24);
  table<128>(out, 64, lanes, 24);
  table<128>(out, 16, lanes, 12);
  table<128>(out, 16, lanes >> 3, 24);
  return 0;
}
