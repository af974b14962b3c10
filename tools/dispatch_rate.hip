// dispatch_rate.hip — how fast does the machine place workgroups?  A lane-per-thread kernel that reads 4 B and
// writes 4 B per lane (the shape of the lane-advance kernels), timed for block sizes 64..1024 and 2^18..2^22 lanes.
// If the time beyond the knee follows the number of WORKGROUPS, fatter workgroups are faster for the same lanes.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int BS, int LPT, bool SYNC>
__global__ void __launch_bounds__(BS) touch(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
  __shared__ unsigned int s[2];
  if (SYNC) { if (threadIdx.x < 2) s[threadIdx.x] = 0; __syncthreads(); }
  const int64_t i0 = ((int64_t)blockIdx.x * BS + threadIdx.x) * LPT;
  int32_t v[LPT];
#pragma unroll
  for (int k = 0; k < LPT; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
#pragma unroll
  for (int k = 0; k < LPT; ++k) if (i0 + k < n) out[i0 + k] = v[k] * 3 + 1;
  if (SYNC) {
    if ((threadIdx.x & 63) == 0) atomicAdd(&s[0], (unsigned)v[0] & 1u);
    __syncthreads();
    if (threadIdx.x == 0 && s[0] == 0xFFFFFFFFu) out[0] = 0;
  }
}

template <int BS, int LPT, bool SYNC>
static float run(const int32_t* in, int32_t* out, int64_t n, int reps) {
  const int64_t per = (int64_t)BS * LPT;
  const dim3 grid((unsigned)((n + per - 1) / per)), block(BS);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < 20; ++r) touch<BS, LPT, SYNC><<<grid, block>>>(in, out, n);
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) touch<BS, LPT, SYNC><<<grid, block>>>(in, out, n);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  const int64_t N = 1 << 22;
  int32_t *in, *out;
  hipMalloc(&in, N * 4); hipMalloc(&out, N * 4);
  hipMemset(in, 1, N * 4);
  printf("us per launch; columns: lanes = 2^18 2^19 2^20 2^21 2^22\n");
#define ROW(BS, LPT, SYNC)                                                                    \
  {                                                                                           \
    printf("block %4d  lanes/thread %d  %s :", BS, LPT, SYNC ? "lds+barriers" : "plain       ");  \
    for (int lg = 18; lg <= 22; ++lg) printf(" %7.2f", run<BS, LPT, SYNC>(in, out, (int64_t)1 << lg, 200)); \
    printf("\n");                                                                           \
  }
  ROW(64, 1, false) ROW(128, 1, false) ROW(256, 1, false) ROW(512, 1, false) ROW(1024, 1, false)
  ROW(256, 2, false) ROW(256, 4, false) ROW(1024, 4, false)
  ROW(64, 1, true) ROW(256, 1, true) ROW(512, 1, true) ROW(1024, 1, true) ROW(256, 4, true)
  return 0;
}
