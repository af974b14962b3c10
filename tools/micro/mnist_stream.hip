// What holds the mnist observation stream (mnist_fam.h) below deep_sea's store rate for the same 784-float row?
// (VERDICT r05 next #1.)  The bare patterns, every one over the same [lanes x cells] f32 array, same box, same process,
// interleaved repetitions.  A kernel = K x 4 KiB runs per workgroup, 16-byte stores, wave-contiguous (the product shape):
//   k_obs    guarded code (`if (live) load`, `if (show) gather`, `if (!live) continue`: exec-mask branches, the compiler's
//            s_waitcnt insertion waits vmcnt(0) at every join — in the one-hot kernel that is ONE store in flight per wave):
//            fill (no loads), hot (deep_sea's chain), cur (the r05 mnist body), nogather / nolut (cur minus one element),
//            lutA (table by arithmetic + workgroup barrier), lutW (per-wave table), arith (no table)
//   k_pipe   STRAIGHT-LINE code: unconditional loads with selected addresses, selected values; R rounds of K stores per wave
//            with the loads of later rounds issued ahead of this round's stores (R = 1: "flat"); LUTMODE: 0 mnist with the
//            per-wave arithmetic table (the r06 product body), 1 per-pixel arithmetic, 2.. one-hot chains with extras
//   k_xcd    which XCD writes which addresses (granule 2^lgG bytes, independent of K)
//   k_paced  all loads first, then the stores with at most N older ones unacknowledged (s_waitcnt vmcnt(N)) or spaced by
//            s_sleep / s_nop
// Findings (profiles/r06/mnist_stream_microbench*.log): the mnist loss was the guarded CODE SHAPE (cur 5.5 -> flat 7.0 TB/s);
// K = 4 is a sharp optimum for an L2-resident image table, 5-6 for a 47 MB one; the one-hot chain is best exactly as the
// product has it (guarded = ack-serialised stores: 6.7; flat 6.1; any pacing by sleeps 5.8-6.1); rounds per wave, XCD granules
// and table-free arithmetic do not help.
// Lane patterns: half (each lane shows an image with probability 1/2: the bench's staggered phases), all, none (ALL_PATTERNS=1).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/mnist_stream.hip -o tools/ab/mnist_stream
// Run:   tools/ab/mnist_stream [lanes=2^20] [cells=784] [images=96]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define BS 256
#define SHOW_BIT (1 << 29)

struct div64 { uint64_t m; uint32_t s; };
struct args {
  float* obs; const int32_t* state; const int8_t* images; const float* lut;
  int64_t n_lanes; uint32_t cells, cells_magic; div64 dv;
};

__device__ __forceinline__ float pixel_value(uint32_t four, int k) {
  const float x = (float)((int32_t)(four << (24 - 8 * k)) >> 24);
  const float r = 0x1.010102p-8f;
  const float q = x * r;
  const float e = __builtin_fmaf(-q, 255.0f, x);
  return __builtin_fmaf(e, r, q);
}

enum { M_FILL, M_HOT, M_CUR, M_NOGATHER, M_NOLUT, M_LUTA, M_LUTW, M_ARITH };

template <int K, int MODE>
__global__ void __launch_bounds__(BS) k_obs(const args a) {
  __shared__ float s_lut_all[1024];
  float* s_lut = s_lut_all;
  const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
  if (MODE == M_LUTW) s_lut += wave * 256;
  const uint32_t cells = a.cells;
  const uint64_t total = (uint64_t)a.n_lanes * cells;
  const uint64_t F0 = (uint64_t)blockIdx.x * (uint64_t)(K * 4 * BS);
  const uint64_t lane_b = __umul64hi(F0, a.dv.m) >> a.dv.s;
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);
  float4* __restrict__ o4 = reinterpret_cast<float4*>(a.obs + F0);
  const int32_t* __restrict__ st = a.state + lane_b;
  uint32_t px[K], r0[K];
  int32_t s[K];
  bool live[K], show[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t c = (wave * K + u) * 64u + wl;
    const uint32_t f = r_b + (c << 2);
    const uint32_t dl = __umulhi(f, a.cells_magic);
    r0[u] = f - dl * cells;
    live[u] = F0 + ((uint64_t)c << 2) + 3 < total;
    show[u] = false; px[u] = 0; s[u] = 0;
    if (MODE == M_FILL) continue;
    if (live[u]) {
      s[u] = st[dl];
      show[u] = (s[u] & SHOW_BIT) != 0;
      if (MODE == M_HOT) continue;
      if (MODE == M_NOGATHER) px[u] = (uint32_t)s[u] * 0x9E3779B1u + r0[u];
      else if (show[u]) px[u] = *reinterpret_cast<const uint32_t*>(a.images + (uint64_t)(s[u] & 0x00FFFFFF) * cells + r0[u]);
    }
  }
  if (MODE == M_CUR || MODE == M_NOGATHER) { s_lut[threadIdx.x] = a.lut[threadIdx.x]; __syncthreads(); }
  if (MODE == M_LUTA) { s_lut[threadIdx.x] = pixel_value(threadIdx.x, 0); __syncthreads(); }
  if (MODE == M_LUTW) {
    float4 l;
    l.x = pixel_value(4 * wl, 0); l.y = pixel_value(4 * wl + 1, 0); l.z = pixel_value(4 * wl + 2, 0); l.w = pixel_value(4 * wl + 3, 0);
    reinterpret_cast<float4*>(s_lut)[wl] = l;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (!live[u]) continue;
    const uint32_t c = (wave * K + u) * 64u + wl;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (MODE == M_HOT) {
      const int h = (s[u] & 0x3FF) - (int)r0[u];      // one hot cell per row
      v.x = h == 0 ? 1.f : 0.f; v.y = h == 1 ? 1.f : 0.f; v.z = h == 2 ? 1.f : 0.f; v.w = h == 3 ? 1.f : 0.f;
    } else if (MODE != M_FILL && show[u]) {
      const uint32_t p = px[u];
      if (MODE == M_NOLUT) { v.x = __uint_as_float(p & 0x3FFFFFFF); v.y = v.x; v.z = v.x; v.w = v.x; }
      else if (MODE == M_ARITH) { v.x = pixel_value(p, 0); v.y = pixel_value(p, 1); v.z = pixel_value(p, 2); v.w = pixel_value(p, 3); }
      else { v.x = s_lut[p & 0xFF]; v.y = s_lut[(p >> 8) & 0xFF]; v.z = s_lut[(p >> 16) & 0xFF]; v.w = s_lut[p >> 24]; }
    }
    o4[c] = v;
  }
}

// R rounds of K stores per wave; a wave owns R*K consecutive KiB.  LUTMODE 0: per-wave arithmetic LUT; 1: per-pixel arithmetic.
// BRANCH-FREE: a divergent `if (show) load` is an exec-mask branch, and the compiler's s_waitcnt insertion gives up at
// every join (vmcnt(0): the wave then waits for its own stores).  Every load is unconditional with a selected address
// (lanes that show nothing read the first image's row: always in L2), every LDS lookup too; the values are selected.
// Workgroups that contain the array's end take the guarded slow path (uniform branch).
template <int K, int R, int LUTMODE, int DEPTH, bool full>
__device__ __forceinline__ void pipe_body(const args& a, float* s_lut_all) {
  const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
  float* s_lut = s_lut_all + wave * 256;
  const uint32_t cells = a.cells;
  const uint64_t total = (uint64_t)a.n_lanes * cells;
  const uint64_t F0 = (uint64_t)blockIdx.x * (uint64_t)(R * K * 4 * BS);
  const uint64_t lane_b = __umul64hi(F0, a.dv.m) >> a.dv.s;
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);
  float4* __restrict__ o4 = reinterpret_cast<float4*>(a.obs + F0);
  const int32_t* __restrict__ st = a.state + lane_b;
  const int8_t* __restrict__ images = a.images;
  uint32_t px[R][K], r0[R][K], dlk[R][K];
  int32_t s[R][K];
#define CH(r, u) ((wave * (R * K) + (r) * K + (u)) * 64u + wl)
#define STAGE_A(r)                                                                 \
  _Pragma("unroll") for (int u = 0; u < K; ++u) {                                  \
    const uint32_t c = CH(r, u);                                                   \
    const uint32_t f = r_b + (c << 2);                                             \
    uint32_t dl = __umulhi(f, a.cells_magic);                                      \
    r0[r][u] = f - dl * cells;                                                     \
    if (!full) dl = (F0 + ((uint64_t)c << 2) + 3 < total) ? dl : 0u;               \
    dlk[r][u] = dl;                                                                \
    s[r][u] = st[dl];                                                              \
  }
#define STAGE_G(r)                                                                 \
  if (LUTMODE == 6) _Pragma("unroll") for (int u = 0; u < K; ++u) {                \
    uint32_t z = (uint32_t)s[r][u]; asm volatile("v_and_b32 %0, 0, %0" : "+v"(z)); \
    px[r][u] = (uint32_t)st[dlk[r][u] + z];                                        \
  }                                                                                \
  if (LUTMODE != 2 && LUTMODE != 4 && LUTMODE != 5 && LUTMODE != 6 && LUTMODE != 7) _Pragma("unroll") for (int u = 0; u < K; ++u) { \
    const uint32_t row = (s[r][u] & SHOW_BIT) ? (uint32_t)(s[r][u] & 0x00FFFFFF) * cells : 0u; \
    px[r][u] = *reinterpret_cast<const uint32_t*>(images + (row + r0[r][u]));      \
  }
#define STAGE_S(r)                                                                 \
  _Pragma("unroll") for (int u = 0; u < K; ++u) {                                  \
    const uint32_t p = px[r][u];                                                   \
    float4 v;                                                                      \
    if (LUTMODE >= 2) {                                                            \
      int sv = s[r][u];                                                            \
      if (LUTMODE == 3) { uint32_t q = p; asm volatile("" : "+v"(q)); sv += (q == 0x9E3779B1u) ? 1 : 0; } \
      if (LUTMODE == 4) sv = reinterpret_cast<int*>(s_lut)[(u * 64 + wl) & 255];   \
      if (LUTMODE == 6) sv = (int)p;                                               \
      const int h = (sv & 0x3FF) - (int)r0[r][u];                                  \
      if (LUTMODE >= 6) { v.x = s_lut[h == 0]; v.y = s_lut[h == 1]; v.z = s_lut[h == 2]; v.w = s_lut[h == 3]; } \
      else { v.x = h == 0 ? 1.f : 0.f; v.y = h == 1 ? 1.f : 0.f; v.z = h == 2 ? 1.f : 0.f; v.w = h == 3 ? 1.f : 0.f; } \
    } else if (LUTMODE == 1) { v.x = pixel_value(p, 0); v.y = pixel_value(p, 1); v.z = pixel_value(p, 2); v.w = pixel_value(p, 3); } \
    else { v.x = s_lut[p & 0xFF]; v.y = s_lut[(p >> 8) & 0xFF]; v.z = s_lut[(p >> 16) & 0xFF]; v.w = s_lut[p >> 24]; } \
    const bool sh = LUTMODE >= 2 || (s[r][u] & SHOW_BIT) != 0;                     \
    v.x = sh ? v.x : 0.f; v.y = sh ? v.y : 0.f; v.z = sh ? v.z : 0.f; v.w = sh ? v.w : 0.f; \
    if (full || F0 + ((uint64_t)CH(r, u) << 2) + 3 < total) o4[CH(r, u)] = v;     \
  }
  STAGE_A(0);
  if (R > 1) { STAGE_A(1); }
  if (DEPTH > 1 && R > 2) { STAGE_A(2); }
  if (LUTMODE == 0) {
    float4 l;
    l.x = pixel_value(4 * wl, 0); l.y = pixel_value(4 * wl + 1, 0); l.z = pixel_value(4 * wl + 2, 0); l.w = pixel_value(4 * wl + 3, 0);
    reinterpret_cast<float4*>(s_lut)[wl] = l;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (LUTMODE == 4) {
    _Pragma("unroll") for (int u = 0; u < K; ++u) reinterpret_cast<int*>(s_lut)[(u * 64 + wl) & 255] = s[0][u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (LUTMODE >= 6) {
    if (wl < 2) s_lut[wl] = (float)wl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (LUTMODE == 5) {
    _Pragma("unroll") for (int u = 0; u < K; ++u) asm volatile("" : "+v"(s[0][u]));
    for (int q = 0; q < DEPTH; ++q) __builtin_amdgcn_s_sleep(8);
  }
  if (LUTMODE == 2 || LUTMODE == 4 || LUTMODE == 5 || LUTMODE == 7) { _Pragma("unroll") for (int rr = 0; rr < R; ++rr) _Pragma("unroll") for (int u = 0; u < K; ++u) px[rr][u] = 0; }
  STAGE_G(0);
  if (DEPTH > 1 && R > 1) { STAGE_G(1); }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r + DEPTH + 1 < R) { STAGE_A(r + DEPTH + 1); }
    if (r + DEPTH < R) { STAGE_G(r + DEPTH); }
    asm volatile("" ::: "memory");
    STAGE_S(r);
    asm volatile("" ::: "memory");
    // nothing that consumes the next round's pixels may be scheduled above this round's stores (it would drag the
    // wait for those loads — and with it, in-order, for older stores — in front of them)
    if (r + 1 < R) {
      _Pragma("unroll") for (int u = 0; u < K; ++u) {
        if (LUTMODE == 2 || LUTMODE >= 4) asm volatile("" : "+v"(s[r + 1 < R ? r + 1 : r][u]));
        else asm volatile("" : "+v"(px[r + 1 < R ? r + 1 : r][u]));
      }
    }
  }
#undef CH
#undef STAGE_A
#undef STAGE_G
#undef STAGE_S
}

template <int K, int R, int LUTMODE, int DEPTH>
__global__ void __launch_bounds__(BS) k_pipe(const args a) {
  __shared__ float s_lut_all[1024];
  const uint64_t total = (uint64_t)a.n_lanes * a.cells;
  if ((uint64_t)(blockIdx.x + 1) * (uint64_t)(R * K * 4 * BS) <= total) pipe_body<K, R, LUTMODE, DEPTH, true>(a, s_lut_all);
  else pipe_body<K, R, LUTMODE, DEPTH, false>(a, s_lut_all);
}


// Which XCD writes which addresses?  Workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b % 8), so with the
// natural order (block b writes bytes [b*P, (b+1)*P), P = K * 4 KiB) XCD x owns every 8th P-sized piece: the piece size IS the
// granule of the XCD <-> address affinity.  This kernel separates the two: XCD x owns the granules q with q % 8 == x of
// 2^lgG bytes each, and its j-th workgroup writes the j-th P bytes of that owned space (G = P: the natural order).
// MODE 0: one-hot chain, guarded loads/stores (the product's code shape); 1: flat mnist chain with the per-wave LUT.
template <int K, int MODE>
__global__ void __launch_bounds__(BS) k_xcd(const args a, const uint32_t lgG) {
  __shared__ float s_lut_all[1024];
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wl = threadIdx.x & 63u;
  float* s_lut = s_lut_all + wave * 256;
  const uint32_t cells = a.cells;
  const uint64_t total = (uint64_t)a.n_lanes * cells;
  const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3;
  uint32_t r0[K], px[K];
  int32_t s[K];
  float4* dst[K];
  bool live[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint64_t o = (uint64_t)j * (uint64_t)(K * 4096) + (uint64_t)((wave * K + u) * 1024u);
    const uint64_t q = o >> lgG;
    const uint64_t addr = (((q << 3) + x) << lgG) + (o & ((1ull << lgG) - 1ull));       // bytes; uniform per wave
    const uint64_t F = addr >> 2;
    const uint64_t lane_run = __umul64hi(F, a.dv.m) >> a.dv.s;
    const uint32_t r_run = (uint32_t)(F - lane_run * cells);
    const uint32_t f = r_run + (wl << 2);
    const uint32_t dl = __umulhi(f, a.cells_magic);
    r0[u] = f - dl * cells;
    dst[u] = reinterpret_cast<float4*>(a.obs + F) + wl;
    live[u] = F + (wl << 2) + 3 < total;
    if (MODE == 0) { s[u] = 0; if (live[u]) s[u] = a.state[lane_run + dl]; }
    else s[u] = a.state[lane_run + dl];
  }
  if (MODE == 1) {
    float4 l;
    l.x = pixel_value(4 * wl, 0); l.y = pixel_value(4 * wl + 1, 0); l.z = pixel_value(4 * wl + 2, 0); l.w = pixel_value(4 * wl + 3, 0);
    reinterpret_cast<float4*>(s_lut)[wl] = l;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const uint32_t row = (s[u] & SHOW_BIT) ? (uint32_t)(s[u] & 0x00FFFFFF) * cells : 0u;
      px[u] = *reinterpret_cast<const uint32_t*>(a.images + (row + r0[u]));
    }
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    float4 v;
    if (MODE == 0) {
      if (!live[u]) continue;
      const int h = (s[u] & 0x3FF) - (int)r0[u];
      v.x = h == 0 ? 1.f : 0.f; v.y = h == 1 ? 1.f : 0.f; v.z = h == 2 ? 1.f : 0.f; v.w = h == 3 ? 1.f : 0.f;
    } else {
      const uint32_t p = px[u];
      const bool sh = (s[u] & SHOW_BIT) != 0;
      v.x = s_lut[p & 0xFF]; v.y = s_lut[(p >> 8) & 0xFF]; v.z = s_lut[(p >> 16) & 0xFF]; v.w = s_lut[p >> 24];
      v.x = sh ? v.x : 0.f; v.y = sh ? v.y : 0.f; v.z = sh ? v.z : 0.f; v.w = sh ? v.w : 0.f;
    }
    *dst[u] = v;
  }
}


// How many of a wave's stores should be in flight?  The guarded kernels above wait vmcnt(0) in front of EVERY store (the
// compiler's s_waitcnt insertion at the joins of their exec-mask branches): a wave has ONE store in flight, and they are the
// faster ones.  Straight-line body, all K state loads first, then store u waits until at most N older stores are unacknowledged.
// MODE 0: one-hot chain; 1: mnist chain (per-wave arithmetic table)
template <int K, int N, int MODE>
__global__ void __launch_bounds__(BS) k_paced(const args a) {
  __shared__ float s_lut_all[1024];
  const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
  float* s_lut = s_lut_all + wave * 256;
  const uint32_t cells = a.cells;
  const uint64_t F0 = (uint64_t)blockIdx.x * (uint64_t)(K * 4 * BS);
  const uint64_t lane_b = __umul64hi(F0, a.dv.m) >> a.dv.s;
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);
  float4* __restrict__ o4 = reinterpret_cast<float4*>(a.obs + F0);
  const int32_t* __restrict__ st = a.state + lane_b;
  int32_t s[K];
  uint32_t r0[K], px[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t c = (wave * K + u) * 64u + wl;
    const uint32_t f = r_b + (c << 2);
    const uint32_t dl = __umulhi(f, a.cells_magic);
    r0[u] = f - dl * cells;
    s[u] = st[dl];
  }
  if (MODE == 1) {
    float4 l;
    l.x = pixel_value(4 * wl, 0); l.y = pixel_value(4 * wl + 1, 0); l.z = pixel_value(4 * wl + 2, 0); l.w = pixel_value(4 * wl + 3, 0);
    reinterpret_cast<float4*>(s_lut)[wl] = l;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const uint32_t row = (s[u] & SHOW_BIT) ? (uint32_t)(s[u] & 0x00FFFFFF) * cells : 0u;
      px[u] = *reinterpret_cast<const uint32_t*>(a.images + (row + r0[u]));
    }
  }
  float4 v[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (MODE == 0) {
      const int h = (s[u] & 0x3FF) - (int)r0[u];
      v[u].x = h == 0 ? 1.f : 0.f; v[u].y = h == 1 ? 1.f : 0.f; v[u].z = h == 2 ? 1.f : 0.f; v[u].w = h == 3 ? 1.f : 0.f;
    } else {
      const uint32_t p = px[u];
      const bool sh = (s[u] & SHOW_BIT) != 0;
      v[u].x = s_lut[p & 0xFF]; v[u].y = s_lut[(p >> 8) & 0xFF]; v[u].z = s_lut[(p >> 16) & 0xFF]; v[u].w = s_lut[p >> 24];
      v[u].x = sh ? v[u].x : 0.f; v[u].y = sh ? v[u].y : 0.f; v[u].z = sh ? v[u].z : 0.f; v[u].w = sh ? v[u].w : 0.f;
    }
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if (N >= 100 && N < 200 && u > 0) __builtin_amdgcn_s_sleep(N - 100);
    if (N >= 200 && u > 0) { _Pragma("unroll") for (int q = 0; q < N - 200; ++q) asm volatile("s_nop 7" ::: "memory"); }
    o4[(wave * K + u) * 64u + wl] = v[u];
    asm volatile("" ::: "memory");
  }
}


// The real dataset's class (image table beyond L2): (NT) the same straight-line body with NON-TEMPORAL stores — do the 3.3 GB
// of stores per launch evict the table from L2 / the Infinity Cache? — and (DELTA > 0) with every workgroup also touching the
// image rows that workgroup b + DELTA (a multiple of 8: the same XCD, hence the same L2) will gather, issued AFTER its own
// gathers (one in-order vmcnt: its own pixels return first and nothing ever waits for the prefetch).
typedef float f4v __attribute__((ext_vector_type(4)));
template <int K, bool NT, int DELTA>
__global__ void __launch_bounds__(BS) k_pref(const args a) {
  __shared__ float s_lut_all[1024];
  const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
  float* s_lut = s_lut_all + wave * 256;
  const uint32_t cells = a.cells;
  const uint64_t total = (uint64_t)a.n_lanes * cells;
  const uint64_t F0 = (uint64_t)blockIdx.x * (uint64_t)(K * 4 * BS);
  const uint64_t lane_b = __umul64hi(F0, a.dv.m) >> a.dv.s;
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);
  f4v* __restrict__ o4 = reinterpret_cast<f4v*>(a.obs + F0);
  const int32_t* __restrict__ st = a.state + lane_b;
  // the block DELTA ahead (clamped to the array)
  uint64_t G0 = F0 + (uint64_t)DELTA * (uint64_t)(K * 4 * BS);
  if (G0 + (uint64_t)(K * 4 * BS) > total) G0 = F0;
  const uint64_t lane_g = __umul64hi(G0, a.dv.m) >> a.dv.s;
  const uint32_t r_g = (uint32_t)(G0 - lane_g * cells);
  const int32_t* __restrict__ stg = a.state + lane_g;
  int32_t s[K], sg[K];
  uint32_t r0[K], rg[K], px[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t c = (wave * K + u) * 64u + wl;
    const uint32_t f = r_b + (c << 2);
    const uint32_t dl = __umulhi(f, a.cells_magic);
    r0[u] = f - dl * cells;
    s[u] = st[dl];
    if (DELTA > 0) {
      const uint32_t g = r_g + (c << 2);
      const uint32_t dg = __umulhi(g, a.cells_magic);
      rg[u] = g - dg * cells;
      sg[u] = stg[dg];
    }
  }
  {
    float4 l;
    l.x = pixel_value(4 * wl, 0); l.y = pixel_value(4 * wl + 1, 0); l.z = pixel_value(4 * wl + 2, 0); l.w = pixel_value(4 * wl + 3, 0);
    reinterpret_cast<float4*>(s_lut)[wl] = l;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t row = (s[u] & SHOW_BIT) ? (uint32_t)(s[u] & 0x00FFFFFF) * cells : 0u;
    px[u] = *reinterpret_cast<const uint32_t*>(a.images + (row + r0[u]));
  }
  uint32_t dummy[K];      // destinations of the prefetch loads: written whenever the data arrives, so they stay reserved
  if (DELTA > 0) {        // (an asm sink after the stores) — the compiler does not know the asm is a load and would reuse them
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const uint32_t row = (sg[u] & SHOW_BIT) ? (uint32_t)(sg[u] & 0x00FFFFFF) * cells : 0u;
      asm volatile("global_load_dword %0, %1, off" : "=&v"(dummy[u]) : "v"(a.images + (row + rg[u])) : "memory");   // never waited for
    }
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t p = px[u];
    const bool sh = (s[u] & SHOW_BIT) != 0;
    f4v v;
    v.x = s_lut[p & 0xFF]; v.y = s_lut[(p >> 8) & 0xFF]; v.z = s_lut[(p >> 16) & 0xFF]; v.w = s_lut[p >> 24];
    v.x = sh ? v.x : 0.f; v.y = sh ? v.y : 0.f; v.z = sh ? v.z : 0.f; v.w = sh ? v.w : 0.f;
    if (NT) __builtin_nontemporal_store(v, &o4[(wave * K + u) * 64u + wl]);
    else o4[(wave * K + u) * 64u + wl] = v;
  }
  if (DELTA > 0) {
#pragma unroll
    for (int u = 0; u < K; ++u) asm volatile("" :: "v"(dummy[u]));
  }
}

static div64 make_div64(uint32_t d) {
  div64 r; uint32_t lg = 0;
  while ((2u << lg) <= d) ++lg;
  r.s = lg - 1;
  r.m = (uint64_t)((((unsigned __int128)1) << (64 + r.s)) / d) + 1;
  return r;
}

typedef void (*launch_fn)(const args&, uint64_t total_floats);
template <int K, int MODE> static void L(const args& a, uint64_t total) {
  const uint64_t per = (uint64_t)K * 4 * BS;
  k_obs<K, MODE><<<dim3((unsigned)((total + per - 1) / per)), dim3(BS)>>>(a);
}
template <int K, int R, int LM, int DEPTH = 1> static void P(const args& a, uint64_t total) {
  const uint64_t per = (uint64_t)R * K * 4 * BS;
  k_pipe<K, R, LM, DEPTH><<<dim3((unsigned)((total + per - 1) / per)), dim3(BS)>>>(a);
}

template <int K, int MODE, int LGG> static void X(const args& a, uint64_t total) {
  const uint64_t per = (uint64_t)K * 4 * BS;
  k_xcd<K, MODE><<<dim3((unsigned)(total / per)), dim3(BS)>>>(a, LGG);      // total % (8 granules) == 0 in this bench
}

template <int K, int N, int MODE> static void W(const args& a, uint64_t total) {
  const uint64_t per = (uint64_t)K * 4 * BS;
  k_paced<K, N, MODE><<<dim3((unsigned)(total / per)), dim3(BS)>>>(a);      // (the tail of a non-multiple is left unwritten: timing only)
}

template <int K, bool NT, int DELTA> static void F(const args& a, uint64_t total) {
  const uint64_t per = (uint64_t)K * 4 * BS;
  k_pref<K, NT, DELTA><<<dim3((unsigned)(total / per)), dim3(BS)>>>(a);     // (total % per == 0 at the bench sizes)
}

struct variant { const char* name; launch_fn fn; bool exact; };

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : (1 << 20);
  const uint32_t cells = argc > 2 ? (uint32_t)atoi(argv[2]) : 784;
  const int n_img = argc > 3 ? atoi(argv[3]) : 96;
  const int reps = 12, rounds = 2;
  const uint64_t total = (uint64_t)B * cells;
  args a;
  int32_t* d_state[3]; int8_t* d_img; float* d_lut; float* d_obs;
  hipMalloc(&d_obs, total * 4 + 65536); hipMalloc(&d_img, (size_t)n_img * cells); hipMalloc(&d_lut, 1024);
  std::vector<int8_t> img((size_t)n_img * cells);
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (auto& p : img) p = (int8_t)(rnd() >> 40);
  hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice);
  float lut[256];
  for (int k = 0; k < 256; ++k) lut[k] = (float)(int8_t)k / 255.0f;
  hipMemcpy(d_lut, lut, 1024, hipMemcpyHostToDevice);
  std::vector<int32_t> st((size_t)B);
  const char* pat_name[3] = {"half", "all", "none"};
  for (int p = 0; p < 3; ++p) {
    for (int64_t i = 0; i < B; ++i) {
      const uint32_t idx = (uint32_t)(rnd() >> 33) % (uint32_t)n_img;
      const bool show = p == 0 ? ((rnd() >> 50) & 1) : p == 1;
      st[(size_t)i] = (int32_t)idx | (int32_t)((idx % 10) << 24) | (show ? SHOW_BIT : 0);
    }
    hipMalloc(&d_state[p], (size_t)B * 4);
    hipMemcpy(d_state[p], st.data(), (size_t)B * 4, hipMemcpyHostToDevice);
  }
  a.obs = d_obs; a.images = d_img; a.lut = d_lut; a.n_lanes = B; a.cells = cells;
  a.cells_magic = (uint32_t)((0x100000000ull / cells) + 1ull); a.dv = make_div64(cells);
  const variant vs[] = {
      {"hot K4 (deep_sea chain)", L<4, M_HOT>, false},
      {"cur K4 (r05 product)", L<4, M_CUR>, true},
      {"flat lutW K4", P<4, 1, 0>, true}, {"flat lutW K6", P<6, 1, 0>, true},
      {"pref K4 plain", F<4, false, 0>, false}, {"pref K6 plain", F<6, false, 0>, false},
      {"pref K4 nt-stores", F<4, true, 0>, false}, {"pref K6 nt-stores", F<6, true, 0>, false},
      {"pref K4 ahead 64", F<4, false, 64>, false}, {"pref K4 ahead 256", F<4, false, 256>, false},
      {"pref K4 ahead 1024", F<4, false, 1024>, false}, {"pref K4 ahead 2048", F<4, false, 2048>, false}, {"pref K4 ahead 4096", F<4, false, 4096>, false},
      {"pref K6 ahead 256", F<6, false, 256>, false}, {"pref K6 ahead 1024", F<6, false, 1024>, false}, {"pref K6 ahead 2048", F<6, false, 2048>, false},
      {"pref K4 nt ahead 1024", F<4, true, 1024>, false}, {"pref K4 nt ahead 2048", F<4, true, 2048>, false},
      {"flat lutW K4 (again)", P<4, 1, 0>, true},
  };
  const int nv = sizeof(vs) / sizeof(vs[0]);
  // correctness of the exact variants against the host (half pattern), first 4096 + last 4096 lanes
  {
    a.state = d_state[0];
    hipMemcpy(st.data(), d_state[0], (size_t)B * 4, hipMemcpyDeviceToHost);
    const int64_t chk = B < 4096 ? B : 4096;
    std::vector<float> got((size_t)chk * cells);
    for (int v = 0; v < nv; ++v) {
      if (!vs[v].exact) continue;
      hipMemset(d_obs, 0xFF, total * 4);
      vs[v].fn(a, total);
      hipDeviceSynchronize();
      long bad = 0;
      for (int part = 0; part < 2; ++part) {
        const int64_t l0 = part == 0 ? 0 : B - chk;
        hipMemcpy(got.data(), d_obs + (uint64_t)l0 * cells, got.size() * 4, hipMemcpyDeviceToHost);
        for (int64_t i = 0; i < chk; ++i)
          for (uint32_t c = 0; c < cells; ++c) {
            const int32_t s = st[(size_t)(l0 + i)];
            const float want = (s & SHOW_BIT) ? lut[(uint8_t)img[(size_t)(s & 0xFFFFFF) * cells + c]] : 0.f;
            if (memcmp(&want, &got[(size_t)i * cells + c], 4) != 0) ++bad;
          }
      }
      if (bad) printf("MISMATCH %-26s %ld floats differ\n", vs[v].name, bad);
    }
    printf("exact variants checked against the host\n");
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("lanes %lld x %u floats = %.3f GB per launch, %d images; us per launch (best of %d rounds of %d launches) and TB/s\n",
         (long long)B, cells, total * 4 / 1e9, n_img, rounds, reps);
  for (int p = 0; p < (getenv("ALL_PATTERNS") ? 3 : 1); ++p) {
    a.state = d_state[p];
    std::vector<float> best((size_t)nv, 1e30f), sum((size_t)nv, 0.f);
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < nv; ++v) {
        vs[v].fn(a, total); vs[v].fn(a, total);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) vs[v].fn(a, total);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const float us = ms * 1e3f / reps;
        if (us < best[v]) best[v] = us;
        sum[v] += us;
      }
    for (int v = 0; v < nv; ++v)
      printf("%-5s %-26s best %8.1f us %6.3f TB/s | mean %8.1f us %6.3f TB/s\n", pat_name[p], vs[v].name, best[v],
             total * 4 / best[v] / 1e6, sum[v] / rounds, total * 4 / (sum[v] / rounds) / 1e6);
  }
  return 0;
}
