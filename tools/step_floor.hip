// step_floor.hip — what a lane-per-thread step kernel pays for, feature by feature (2^20 lanes, 256-thread
// workgroups): the bandit-shaped traffic alone, then the pieces every family's kernel adds around it.
//   hipcc --offload-arch=gfx950 -O3 tools/step_floor.hip -o tools/step_floor.bin && tools/step_floor.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct args_t {
  const int32_t* action; int32_t* state; float* reward; float* discount; int8_t* type; float* obs; double* info;
  unsigned long long* counters; const uint64_t* step_base; int64_t n; double rewards[32];
  uint64_t pad[64];     // makes the kernarg ~1 KiB like the real argument structs
};

template <int V>
__global__ void __launch_bounds__(256) step_like(const args_t a) {
  __shared__ unsigned int s_cnt[2];
  if (V >= 1) { if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0; __syncthreads(); }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t step = 0;
  if (V >= 4) step = *a.step_base;
  int type = -1;
  if (i < a.n) {
    const int32_t st = a.state[i];
    const int act = a.action[i];
    float r = 0.f, d = 1.f;
    if (st) { a.state[i] = 0; type = 0; }
    else {
      const double rew = a.rewards[(act + (int)step) & 15];
      if (V >= 3) a.info[i] += 1.0 - rew;
      a.state[i] = 1; type = 2; r = (float)rew; d = 0.f;
    }
    a.reward[i] = r; a.discount[i] = d; a.type[i] = (int8_t)type; a.obs[i] = 1.0f;
  }
  if (V >= 2) {
    unsigned long long last = __ballot(type == 2), first = __ballot(type == 0);
    if ((threadIdx.x & 63) == 0) {
      if (last) atomicAdd(&s_cnt[0], (unsigned)__popcll(last));
      if (first) atomicAdd(&s_cnt[1], (unsigned)__popcll(first));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long* shard = a.counters + (size_t)(blockIdx.x & 255) * 16;
      if (s_cnt[0]) atomicAdd(&shard[0], (unsigned long long)s_cnt[0]);
      if (s_cnt[1]) atomicAdd(&shard[1], (unsigned long long)s_cnt[1]);
    }
  }
}

template <int V>
static float run(const args_t& a, int reps) {
  const dim3 grid((unsigned)((a.n + 255) / 256)), block(256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 20; ++r) step_like<V><<<grid, block>>>(a);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) step_like<V><<<grid, block>>>(a);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  args_t a{};
  const int64_t N = 1 << 22;
  void* p;
#define ALLOC(field, bytes) (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); a.field = (decltype(a.field))p;
  ALLOC(action, N * 4) ALLOC(state, N * 4) ALLOC(reward, N * 4) ALLOC(discount, N * 4) ALLOC(type, N) ALLOC(obs, N * 4)
  ALLOC(info, N * 8) ALLOC(counters, 256 * 16 * 8) ALLOC(step_base, 8)
  for (int k = 0; k < 32; ++k) a.rewards[k] = k / 10.0;
  const char* what[] = {"traffic only (2 loads, 5 stores per lane)", "+ LDS counter init + barrier in front",
                        "+ ballots, LDS atomics, end barrier, 2 sharded global atomics per workgroup",
                        "+ f64 info read-modify-write on the LAST half of the lanes", "+ call counter read from device memory"};
  printf("us per launch at 2^20 / 2^21 / 2^22 lanes\n");
  for (int lg = 20; lg <= 22; ++lg) {
    a.n = (int64_t)1 << lg;
    float t[5] = {run<0>(a, 300), run<1>(a, 300), run<2>(a, 300), run<3>(a, 300), run<4>(a, 300)};
    for (int v = 0; v < 5; ++v) printf("2^%d  V%d %7.2f  %s\n", lg, v, t[v], what[v]);
  }
  return 0;
}
