// Which of a step's output streams is expensive?  Every lane of a [T, B] rollout writes, per step, a reward (4 B), a
// discount (4 B), a step_type (1 B) and an observation row (12 B: mountain_car; 24 B: cartpole) into four [T, B, ...]
// arrays and reads an action (4 B) — per-lane stores exactly as the fused rollout kernels issue them, no arithmetic.
// Variants switch streams off one at a time, replace the byte stream by a dword stream, and write the scalars as
// 16-byte chunks staged through the wave's LDS.
// Build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/micro/step_stores.hip -o tools/ab/step_stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct row3 { float a, b, c; } __attribute__((packed, aligned(4)));

// MASK bits: 1 reward, 2 discount, 4 step_type (byte), 8 observation row, 16 action load, 32 step_type as dword,
// 64 reward/discount/step_type staged per wave and written as 16-byte chunks, 128 action loads in runs of eight with one
// wait per run, 256 all of a launch's action loads before its first store, 512 the action rows through the scalar cache
template <int MASK, int ROWF, int F = 0>
__global__ void __launch_bounds__(256) k(const int32_t* __restrict__ act, float* __restrict__ rew, float* __restrict__ dis,
                                          int8_t* __restrict__ typ, float* __restrict__ obs, int T, size_t B) {
  __shared__ __attribute__((aligned(16))) float s_stage[4][64 * 3];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int wl = threadIdx.x & 63, w = threadIdx.x >> 6;
  float v = (float)threadIdx.x;
  int a = 0, pk = 0;
  int nx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float r0 = v, r1 = v * 1.5f, r2 = v + 2.f, r3 = v - 3.f;
  const float m = 1.0000001f, c = 1e-9f;
  for (int t = 0; t < T; ++t) {
    const size_t oi = (size_t)t * B + i;
#pragma unroll
    for (int f = 0; f < F / 4; ++f) {
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(m), "v"(c));
    }
    if (F) v = r0 + r1 + r2 + r3;
    if (MASK & 16) a += act[oi];
    if ((MASK & 128) && (t & 7) == 0) {                // actions in runs of eight, one wait per run (the rollout kernels)
      int s = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (t + j < T) ? act[oi + (size_t)j * B] << (4 * j) : 0;
      pk = s;
      __builtin_amdgcn_s_waitcnt(0x0070);
    }
    if ((MASK & 256) && t == 0) {                      // every action of the launch before the first store (T <= 16)
      int s = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += (j < T) ? act[i + (size_t)j * B] << (2 * j) : 0;
      pk = s;
      __builtin_amdgcn_s_waitcnt(0x0070);
    }
    if ((MASK & 2048) && (t & 7) == 0) {               // runs of eight from ONE row (cache-resident): load instructions without HBM reads
      int s = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += act[i + (size_t)(j & 1) * B] << (4 * j);
      pk = s;
      __builtin_amdgcn_s_waitcnt(0x0070);
    }
    if ((MASK & 1024) && (t & 7) == 0) {               // runs of eight, PREFETCHED one run ahead by loads the compiler does not
      constexpr int SPS = ((MASK & 1) ? 1 : 0) + ((MASK & 2) ? 1 : 0) + ((MASK & 4) ? 1 : 0) + ((MASK & 8) ? (ROWF == 3 ? 1 : 3) : 0);
      if (t == 0) {                                    // see; awaited with a count the run's stores need not satisfy
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("global_load_dword %0, %1, off" : "=v"(nx[j]) : "v"(act + i + (size_t)j * B) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) :: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7]) : "n"(8 * SPS) : "memory");
      }
      int s = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += nx[j] << (4 * j);
      pk = s;
      if (t + 8 < T) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("global_load_dword %0, %1, off" : "=v"(nx[j]) : "v"(act + oi + (size_t)(8 + j) * B) : "memory");
      }
    }
    if ((MASK & 4096) && (t & 7) == 0) {               // ONE pre-packed dword per lane and run of eight (a pre-pass wrote it)
      pk = act[(size_t)(t >> 3) * B + i];
      __builtin_amdgcn_s_waitcnt(0x0070);
    }
    if ((MASK & 8192) && (t & 15) == 0) {              // ONE pre-packed qword per lane and run of sixteen
      const long long q = reinterpret_cast<const long long*>(act)[(size_t)(t >> 4) * B + i];
      pk = (int)q ^ (int)(q >> 32);
      __builtin_amdgcn_s_waitcnt(0x0070);
    }
    if (MASK & (128 | 256 | 1024 | 2048 | 4096 | 8192)) { a += (pk >> (t & 7)) & 3; v += (float)(a & 1); }
    if (MASK & 512) {                                  // the wave's 64 actions of this step through the SCALAR cache: 4 x s_load_dwordx16
      const int32_t* row = act + (size_t)t * B + (size_t)__builtin_amdgcn_readfirstlane((int)(i & ~(size_t)63));
      typedef int v16 __attribute__((ext_vector_type(16)));
      const v16* q = reinterpret_cast<const v16*>(row);
      v16 s0 = q[0], s1 = q[1], s2 = q[2], s3 = q[3];
      a += s0[0] + s1[5] + s2[10] + s3[15] + s0[7] + s1[9] + s2[1] + s3[3];
      v += (float)(a & 1);
    }
    if (MASK & 64) {
      s_stage[w][wl] = v; s_stage[w][64 + wl] = v + 1.f;
      reinterpret_cast<int8_t*>(&s_stage[w][128])[wl] = (int8_t)(t & 3);
      __builtin_amdgcn_wave_barrier();
      const size_t w0 = oi - wl;                       // the wave's first element
      if (wl < 16) reinterpret_cast<float4*>(rew + w0)[wl] = reinterpret_cast<float4*>(&s_stage[w][0])[wl];
      else if (wl < 32) reinterpret_cast<float4*>(dis + w0)[wl - 16] = reinterpret_cast<float4*>(&s_stage[w][64])[wl - 16];
      else if (wl < 36) reinterpret_cast<float4*>(typ + w0)[wl - 32] = reinterpret_cast<float4*>(&s_stage[w][128])[wl - 32];
      __builtin_amdgcn_wave_barrier();
    } else {
      if (MASK & 1) rew[oi] = v;
      if (MASK & 2) dis[oi] = v + 1.f;
      if (MASK & 4) typ[oi] = (int8_t)(t & 3);
      if (MASK & 32) reinterpret_cast<int32_t*>(typ)[oi] = t & 3;
    }
    if (MASK & 8) {
      if (ROWF == 3) { row3 r; r.a = v; r.b = v; r.c = v; *reinterpret_cast<row3*>(obs + oi * 3) = r; }
      else { float2* d = reinterpret_cast<float2*>(obs + oi * 6); d[0] = make_float2(v, v); d[1] = make_float2(v, v); d[2] = make_float2(v, v); }
    }
    v += 1.f;
  }
  if (a == 123456789) rew[i] = 0.f;
}

static int g_rot = 1;
static int32_t* g_act; static float *g_rew, *g_dis, *g_obs; static int8_t* g_typ;

template <int MASK, int ROWF, int F = 0>
static void run(const char* what, int T, size_t B) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MASK, ROWF, F><<<(unsigned)(B / 256), 256>>>(g_act, g_rew, g_dis, g_typ, g_obs, T, B);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  // g_rot > 1: every launch reads another copy of the action array (g_rot * 67 MB in rotation: no copy stays in the
  // 256 MiB Infinity Cache from one launch to the next, the loads come from HBM)
  for (int r = 0; r < 10; ++r) k<MASK, ROWF, F><<<(unsigned)(B / 256), 256>>>(g_act + (size_t)(r % g_rot) * B * T, g_rew, g_dis, g_typ, g_obs, T, B);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / 10 * 1e3;
  double bytes = 0;
  if (MASK & 1) bytes += 4; if (MASK & 2) bytes += 4; if (MASK & 4) bytes += 1; if (MASK & 32) bytes += 4;
  if (MASK & 64) bytes += 9; if (MASK & 8) bytes += 4 * ROWF; if (MASK & (16 | 128 | 256 | 512 | 1024 | 2048)) bytes += 4;
  if (MASK & (4096 | 8192)) bytes += 0.5;
  printf("  %-58s %7.2f us per step  %5.2f TB/s (%2.0f B per lane-step)\n", what, us / T, bytes * B * T / us / 1e6, bytes);
}

template <int ROWF>
static void table(int T, size_t B) {
  printf("rows of %d floats, %zu lanes, T = %d\n", ROWF, B, T);
  run<1 | 2 | 4 | 8 | 16, ROWF>("all five streams (the rollout kernels)", T, B);
  run<1 | 2 | 4 | 8, ROWF>("without the action loads", T, B);
  run<1 | 2 | 4 | 8 | 128, ROWF>("action loads in runs of 8 (one wait per run)", T, B);
  run<1 | 2 | 4 | 8 | 256, ROWF>("all action loads before the first store", T, B);
  run<128, ROWF>("action loads only (runs of 8)", T, B);
  run<1 | 2 | 8 | 16, ROWF>("without step_type", T, B);
  run<1 | 2 | 32 | 8 | 16, ROWF>("step_type as a dword stream", T, B);
  run<4 | 8 | 16, ROWF>("without reward / discount", T, B);
  run<1 | 2 | 4 | 16, ROWF>("without the observation rows", T, B);
  run<8, ROWF>("observation rows only", T, B);
  run<1, ROWF>("reward only", T, B);
  run<4, ROWF>("step_type only", T, B);
  run<64 | 8 | 16, ROWF>("scalars staged per wave, 16-byte chunks", T, B);
  run<64, ROWF>("staged scalars only", T, B);
  run<0, ROWF, 128>("128 FMAs per step, no memory traffic", T, B);
  run<1 | 2 | 4 | 8, ROWF, 128>("128 FMAs per step + the four store streams", T, B);
  run<0, ROWF, 256>("256 FMAs per step, no memory traffic", T, B);
  run<1 | 2 | 4 | 8, ROWF, 256>("256 FMAs per step + the four store streams", T, B);
  run<0, ROWF, 512>("512 FMAs per step, no memory traffic", T, B);
  run<1 | 2 | 4 | 8, ROWF, 512>("512 FMAs per step + the four store streams", T, B);
  run<8, ROWF, 256>("256 FMAs per step + observation rows only", T, B);
  run<1 | 2 | 4, ROWF, 256>("256 FMAs per step + the three scalar streams", T, B);
}

int main() {
  const size_t B = (size_t)1 << 20;
  const int T = 16;
  hipMalloc(&g_act, B * T * 4 * 8); hipMalloc(&g_rew, B * T * 4); hipMalloc(&g_dis, B * T * 4); hipMalloc(&g_typ, B * T * 4);
  hipMalloc(&g_obs, B * T * 24);
  hipMemset(g_act, 0, B * T * 4 * 8);
  table<3>(T, B);
  table<6>(T, B);
  g_rot = 8;
  printf("the same with the actions read from HBM (8 copies in rotation)\n");
  run<1 | 2 | 4 | 8 | 16, 3>("rows of 3: an action load per step", T, B);
  run<1 | 2 | 4 | 8 | 16, 6>("rows of 6: an action load per step", T, B);
  run<1 | 2 | 4 | 8 | 128, 3>("rows of 3: action loads in runs of 8", T, B);
  run<1 | 2 | 4 | 8 | 256, 3>("rows of 3: all action loads before the first store", T, B);
  run<128, 3>("rows of 3: action loads only (runs of 8)", T, B);
  run<1 | 2 | 4 | 8 | 128, 6>("rows of 6: action loads in runs of 8", T, B);
  run<1 | 2 | 4 | 8 | 128, 3, 128>("rows of 3: 128 FMAs + action loads in runs of 8", T, B);
  run<1 | 2 | 4 | 8 | 1024, 3>("rows of 3: runs of 8 prefetched a run ahead, exact vmcnt", T, B);
  run<1 | 2 | 4 | 8 | 1024, 6>("rows of 6: runs of 8 prefetched a run ahead, exact vmcnt", T, B);
  run<1 | 2 | 4 | 8 | 2048, 3>("rows of 3: runs of 8 from one cache-resident row", T, B);
  run<1 | 2 | 4 | 8 | 2048, 6>("rows of 6: runs of 8 from one cache-resident row", T, B);
  run<1 | 2 | 4 | 8 | 4096, 3>("rows of 3: one pre-packed dword per run of 8", T, B);
  run<1 | 2 | 4 | 8 | 4096, 6>("rows of 6: one pre-packed dword per run of 8", T, B);
  run<1 | 2 | 4 | 8 | 8192, 3>("rows of 3: one pre-packed qword per run of 16", T, B);
  run<1 | 2 | 4 | 8 | 8192, 6>("rows of 6: one pre-packed qword per run of 16", T, B);
  run<1 | 2 | 4 | 8 | 512, 3>("rows of 3: the same rows through the scalar cache (s_load)", T, B);
  run<1 | 2 | 4 | 8 | 512, 6>("rows of 6: the same rows through the scalar cache (s_load)", T, B);
  run<512, 3>("rows of 3: scalar loads only", T, B);
  run<1 | 2 | 4 | 8, 3>("rows of 3: no loads", T, B);
  run<1 | 2 | 4 | 8, 6>("rows of 6: no loads", T, B);
  return 0;
}
