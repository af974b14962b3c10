// Do a wave's stores overlap with the chip's arithmetic?  Every thread runs ITERS rounds of F independent-chain FMAs
// followed by one store; time(F, store) against time(F, no store) and time(0, store) says whether arithmetic and
// stores cost their maximum or their sum (the fused physics rollouts cost the sum: profiles/r03/exp_store_ablation*.log).
// Store shapes: 0 none, 1 dense 16 B per lane (1 KiB per wave instruction), 2 8 B per lane at stride 24 (row-per-lane),
// 3 dense 4 B per lane.  2048 workgroups of 256 threads: one dispatch round, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/micro/store_overlap.hip -o tools/ab/store_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 64

template <int F, int SHAPE>
__global__ void __launch_bounds__(256) k(float* __restrict__ out, float seed, size_t stride_floats) {
  float r0 = seed + threadIdx.x, r1 = r0 * 1.5f, r2 = r0 + 2.f, r3 = r0 - 3.f;
  const float m = 1.0000001f, c = 1e-9f * seed;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int f = 0; f < F / 4; ++f) {
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(m), "v"(c));
    }
    float* base = out + (size_t)it * stride_floats;
    if (SHAPE == 1) reinterpret_cast<float4*>(base)[gid] = make_float4(r0, r1, r2, r3);
    if (SHAPE == 2) *reinterpret_cast<float2*>(base + gid * 6) = make_float2(r0, r1);
    if (SHAPE == 3) base[gid] = r0;
  }
  if (SHAPE == 0 && r0 + r1 + r2 + r3 == 12345.678f) out[gid] = r0;
}

template <int F, int SHAPE>
static float run(float* out, size_t stride_floats) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<F, SHAPE><<<2048, 256>>>(out, 1.0f, stride_floats);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) k<F, SHAPE><<<2048, 256>>>(out, 1.0f, stride_floats);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f;
}

template <int F>
static void row(float* out) {
  const size_t n = (size_t)2048 * 256;
  const float t0 = run<F, 0>(out, 0), t1 = run<F, 1>(out, n * 4), t2 = run<F, 2>(out, n * 6), t3 = run<F, 3>(out, n);
  printf("F=%4d FMAs per store: no store %7.1f us | dense 16 B %7.1f us | 8 B at stride 24 %7.1f us | dense 4 B %7.1f us\n", F, t0, t1, t2, t3);
}

int main() {
  float* out;
  const size_t n = (size_t)2048 * 256;
  hipMalloc(&out, n * 6 * 4 * ITERS);                      // 805 MB: the largest shape's footprint
  printf("per launch: %d stores per thread; bytes: dense 16 B %.0f MB, stride-24 %.0f MB (in a %.0f MB range), dense 4 B %.0f MB\n", ITERS,
         n * 16.0 * ITERS / 1e6, n * 8.0 * ITERS / 1e6, n * 24.0 * ITERS / 1e6, n * 4.0 * ITERS / 1e6);
  row<0>(out); row<32>(out); row<64>(out); row<128>(out); row<256>(out); row<512>(out);
  return 0;
}
