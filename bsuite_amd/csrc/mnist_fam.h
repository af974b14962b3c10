// mnist_fam.h — the device side of the batched MNIST bandit (bsuite/environments/mnist.py:61-75): the lane
// advance body and the observation stream body.  Shared by mnist.hip and pair_mixed.hip (the sweep's
// mixed two-kernel group).
#ifndef BSX_MNIST_FAM_H_
#define BSX_MNIST_FAM_H_

#include "bsx_device.h"
#include "bsx_math.h"

struct mnist_args {
  bsx_ctl ctl;
  const int32_t* action;
  int32_t* state;
  bsx_timestep_t out;
  double* info;
  const int8_t* images;
  const uint8_t* labels;
  int32_t num_data;
  int32_t num_pixels;
};

#define MN_RESET_BIT (1 << 28)
#define MN_SHOW_BIT (1 << 29)

// MT = 0: no segment of the launch is in MT19937-exact mode (the whole-sweep group): those draws are compiled out
template <int MT = -1>
__device__ __forceinline__ void mnist_advance_body(const mnist_args& a, uint32_t block_id, unsigned int* s_cnt) {
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)block_id * BSX_BLOCK + threadIdx.x;
  int type = -1;
  if (i < a.ctl.n_lanes) {
    const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;
    const uint64_t step = bsx_step_of(a.ctl);
    const int32_t st = a.ctl.state_in != nullptr ? a.ctl.state_in[i] : a.state[i];
    double reward = 0.0;
    int32_t nst;
    if (a.ctl.force_reset || (st & MN_RESET_BIT)) {             // mnist.py:61-67
      bsx_draws d;
      bsx_draws_begin<MT>(&d, a.ctl, i, lane, step);
      const uint32_t idx = bsx_randint(&d, (uint32_t)a.num_data);
      bsx_draws_end<MT>(&d, a.ctl, i);
      nst = (int32_t)idx | ((int32_t)a.labels[idx] << 24) | MN_SHOW_BIT;
      type = BSX_FIRST;
    } else {                                                    // mnist.py:69-75
      const int label = (st >> 24) & 0xF;
      reward = (bsx_action(a.ctl, a.action, i, step) == label) ? 1.0 : -1.0;
      a.info[i] += 1.0 - reward;
      nst = (st & 0x0FFFFFFF) | MN_RESET_BIT;                   // SHOW bit cleared: obs = zeros
      type = BSX_LAST;
    }
    a.state[i] = nst;
    bsx_emit_at<-1, -1, true, MT>(a.ctl, a.out, i, i, lane, step, type, reward);
  }
  bsx_count_types(a.ctl, type, s_cnt);
  bsx_final_barrier();
  bsx_flush_counts(a.ctl, s_cnt, block_id);
}

struct mnist_observe_args {
  float* obs;
  const int32_t* state;
  const int8_t* images;
  int64_t n_lanes;
  uint32_t cells;
  uint32_t cells_magic;
  bsx_div64 dv;
  int32_t arith;          // the table is exactly np.float32(int8) / 255 (mnist_make checks all 256 entries): computed, never read
  int32_t nt;             // the image table is larger than the chip's L2: non-temporal observation stores (mnist.hip)
  float lut[256];         // read only when !arith (a caller-defined pixel table)
};

// Block b writes floats [b*K*1024, (b+1)*K*1024) of the [B x num_pixels] observation array (num_pixels % 4 == 0, so a
// 16-byte chunk never straddles two lanes); each wave owns K consecutive KiB (the deep_sea stream order).
//
// Round 6 (profiles/r06/mnist_stream_microbench*.log, tools/micro/mnist_stream.hip — every element of the chain switched on and
// off over the same [2^20 x 784] array in one process): the r05 body {guarded state load -> guarded image gather -> 256-entry
// table copied from the kernel arguments + workgroup barrier -> LDS lookups -> guarded store} streamed 5.5-5.8 TB/s where
// deep_sea's one-hot chain over the same rows streams 6.7-6.8; neither the row geometry (deep_sea size 28 through the
// library: 6.56) nor the gather, the table or the barrier alone explains it (each removed alone: 5.0-5.9).  What does is
// the CODE SHAPE: every `if (live) load` / `if (show) gather` / `if (!live) continue` is an exec-mask branch, and the
// compiler's s_waitcnt insertion falls back to vmcnt(0) at every join — the four chunks of a thread run as four
// serialised {load, wait, gather, wait, lookup, store} chains.  This body is STRAIGHT-LINE: every load is unconditional with a
// selected address (a lane that shows nothing reads the first image's row — one L2-resident line — and discards it), every
// lookup too, the values are selected (v_cndmask); the table is per WAVE (no workgroup barrier) and filled by arithmetic
// (bsx_mnist_pixel_value: exact, tests/test_physics_math.py), so nothing but the state word and the pixels is waited for:
// 7.0 TB/s in the bare pattern with half, all or none of the lanes showing.  Workgroups that contain the array's end take
// the guarded form of the same body (uniform branch).  What did NOT help (same log): more or fewer KiB per wave (K = 3: 5.95,
// K = 5: 5.89, K = 8: 5.9), several rounds per wave with the next round's loads issued ahead of this round's stores
// (5.6-5.8), table-free pixel arithmetic (equal when every lane shows, 6.2-6.5 otherwise), a fixed XCD <-> address granule.
#define MNIST_LUT_FLOATS (256 * (BSX_BLOCK / BSX_WAVE))
// NT: non-temporal stores — for an image table beyond the chip's L2 (the real dataset): 3.3 GB of ordinary stores per call
// evict the table between two gathers of the same row; with nt stores it stays cached (47 MB table: 5.0-5.3 -> 5.8 TB/s,
// 34 MB: 5.6-5.8 -> 6.1); on a table that fits L2 they cost 10 % (6.95 -> 6.2: profiles/r06/mnist_stream_microbench_9*, _10*.log)
template <int K, bool ARITH, bool FULL, bool NT = false>
__device__ __forceinline__ void mnist_observe_chunks(const mnist_observe_args& a, uint32_t block_id, float* s_lut) {
  const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
  s_lut += wave * 256;                                                   // this wave's copy
  const uint32_t cells = a.cells;
  const uint64_t total = (uint64_t)a.n_lanes * cells;
  const uint64_t F0 = (uint64_t)block_id * (uint64_t)(K * 4 * BSX_BLOCK);
  const uint64_t lane_b = __umul64hi(F0, a.dv.m) >> a.dv.s;
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);
  bsx_f4* __restrict__ o4 = reinterpret_cast<bsx_f4*>(a.obs + F0);
  const int32_t* __restrict__ st = a.state + lane_b;
  const int8_t* __restrict__ images = a.images;
  int32_t s[K];
  uint32_t r0[K], px[K];
  bool live[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t c = (wave * K + u) * 64u + wl;
    const uint32_t f = r_b + (c << 2);
    uint32_t dl = __umulhi(f, a.cells_magic);
    r0[u] = f - dl * cells;
    live[u] = FULL || F0 + ((uint64_t)c << 2) + 3 < total;
    if (!FULL) { dl = live[u] ? dl : 0u; r0[u] = live[u] ? r0[u] : 0u; }  // past the end: the block's first lane (in range)
    s[u] = st[dl];
  }
  {                                                                      // mnist.py:64 astype(f32) / 255, all 256 bytes
    bsx_f4 l;
    if (ARITH) {
      l.x = bsx_mnist_pixel_value(4u * wl, 0); l.y = bsx_mnist_pixel_value(4u * wl + 1u, 0);
      l.z = bsx_mnist_pixel_value(4u * wl + 2u, 0); l.w = bsx_mnist_pixel_value(4u * wl + 3u, 0);
    } else {
      l.x = a.lut[4u * wl]; l.y = a.lut[4u * wl + 1u]; l.z = a.lut[4u * wl + 2u]; l.w = a.lut[4u * wl + 3u];
    }
    reinterpret_cast<bsx_f4*>(s_lut)[wl] = l;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {   // four int8 pixels of image idx: one aligned dword of the table (row 0 when nothing shows)
    const uint64_t row = (s[u] & MN_SHOW_BIT) ? (uint64_t)(uint32_t)(s[u] & 0x00FFFFFF) * cells : 0ull;
    px[u] = *reinterpret_cast<const uint32_t*>(images + (row + r0[u]));
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t p = px[u];
    const bool show = (s[u] & MN_SHOW_BIT) != 0;                          // mnist.py:73 zeros after the guess
    bsx_f4 v;
    v.x = s_lut[p & 0xFF]; v.y = s_lut[(p >> 8) & 0xFF]; v.z = s_lut[(p >> 16) & 0xFF]; v.w = s_lut[p >> 24];
    v.x = show ? v.x : 0.f; v.y = show ? v.y : 0.f; v.z = show ? v.z : 0.f; v.w = show ? v.w : 0.f;
    if (FULL || live[u]) {
      if (NT) __builtin_nontemporal_store(v, &o4[(wave * K + u) * 64u + wl]);
      else o4[(wave * K + u) * 64u + wl] = v;
    }
  }
}

template <int K, bool NT>
__device__ __forceinline__ void mnist_observe_body_nt(const mnist_observe_args& a, uint32_t block_id, float* s_lut) {
  const uint64_t total = (uint64_t)a.n_lanes * a.cells;
  const bool full = ((uint64_t)block_id + 1ull) * (uint64_t)(K * 4 * BSX_BLOCK) <= total;      // uniform
  if (a.arith) {
    if (full) mnist_observe_chunks<K, true, true, NT>(a, block_id, s_lut);
    else mnist_observe_chunks<K, true, false, NT>(a, block_id, s_lut);
  } else {
    if (full) mnist_observe_chunks<K, false, true, NT>(a, block_id, s_lut);
    else mnist_observe_chunks<K, false, false, NT>(a, block_id, s_lut);
  }
}

template <int K>
__device__ __forceinline__ void mnist_observe_body(const mnist_observe_args& a, uint32_t block_id, float* s_lut) {
#if defined(BSX_AB_MNIST_NT)        // measurement builds only
  mnist_observe_body_nt<K, true>(a, block_id, s_lut);
  return;
#endif
  if (a.nt) mnist_observe_body_nt<K, true>(a, block_id, s_lut);                                // uniform
  else mnist_observe_body_nt<K, false>(a, block_id, s_lut);
}

#endif  // BSX_MNIST_FAM_H_
