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
  int32_t arith;          // the LUT is exactly np.float32(int8) / 255 (mnist_make checks all 256 entries): compute it
  int32_t _pad;
  float lut[256];
};

// Block b writes floats [b*K*1024, (b+1)*K*1024) of the [B x num_pixels] observation array
// (num_pixels % 4 == 0, so a 16-byte chunk never straddles two lanes).  VAR (A/B knob
// BSX_MNIST_VARIANT): bit 0 = issue the state loads + image gathers BEFORE the LUT fill and its
// barrier; bit 1 = each wave owns K consecutive KiB (the deep_sea stream order) instead of the
// block-interleaved order; bit 2 = no workgroup barrier at all: every WAVE keeps its own copy of the LUT
// (s_lut: MNIST_LUT_FLOATS floats) and fills it only when one of its chunks shows an image — on the calls after the
// guess, when every lane's observation is zeros (mnist.py:73), a workgroup's stores wait for nothing but the state
// loads (r03: the mnist half of the sweep's stream ran at 5.0-5.2 TB/s even then, WAIT_ANY 68 % of the wave cycles:
// the LUT load + barrier in front of every workgroup's stores, profiles/r03/stream_mnist_pmc_sq.json).
#define MNIST_LUT_FLOATS (256 * (BSX_BLOCK / BSX_WAVE))
// bit 3 = no LUT at all: the pixel values are computed (mnist_pixel_value) — no LDS, no barrier.
template <int K, int VAR>
__device__ __forceinline__ void mnist_observe_body(const mnist_observe_args& a, uint32_t block_id, float* s_lut) {
  if (VAR & 4) s_lut += (threadIdx.x >> 6) * 256;          // this wave's copy
  if (!(VAR & 1) && !(VAR & 4) && !(VAR & 8)) {
    s_lut[threadIdx.x] = a.lut[threadIdx.x];
    __syncthreads();
  }
  const uint32_t cells = a.cells;
  const uint64_t total = (uint64_t)a.n_lanes * cells;
  const uint64_t F0 = (uint64_t)block_id * (uint64_t)(K * 4 * BSX_BLOCK);
  const uint64_t lane_b = __umul64hi(F0, a.dv.m) >> a.dv.s;
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);
  bsx_f4* __restrict__ o4 = reinterpret_cast<bsx_f4*>(a.obs + F0);
  const int32_t* __restrict__ st = a.state + lane_b;
  const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
  uint32_t px[K];
  bool live[K], show[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t c = (VAR & 2) ? (wave * K + u) * 64u + wl : threadIdx.x + u * BSX_BLOCK;
    const uint32_t f = r_b + (c << 2);
    const uint32_t dl = __umulhi(f, a.cells_magic);
    const uint32_t r0 = f - dl * cells;
    live[u] = F0 + ((uint64_t)c << 2) + 3 < total;
    show[u] = false;
    px[u] = 0;
    if (live[u]) {
      const int32_t s = st[dl];
      show[u] = (s & MN_SHOW_BIT) != 0;
      if (show[u])    // four int8 pixels of image idx: one aligned dword of the table
        px[u] = *reinterpret_cast<const uint32_t*>(a.images + (uint64_t)(s & 0x00FFFFFF) * cells + r0);
    }
  }
  if (VAR & 8) {
  } else if (VAR & 4) {
    bool any_show = false;
#pragma unroll
    for (int u = 0; u < K; ++u) any_show |= show[u];
    if (__ballot(any_show) != 0ull) {                          // wave-uniform
#pragma unroll
      for (int k = 0; k < 4; ++k) s_lut[wl + 64u * k] = a.lut[wl + 64u * k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  } else if (VAR & 1) {
    s_lut[threadIdx.x] = a.lut[threadIdx.x];
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (!live[u]) continue;
    const uint32_t c = (VAR & 2) ? (wave * K + u) * 64u + wl : threadIdx.x + u * BSX_BLOCK;
    bsx_f4 v = {0.f, 0.f, 0.f, 0.f};                            // mnist.py:73 zeros after the guess
    if (show[u]) {                                              // mnist.py:64 astype(f32) / 255
      const uint32_t p = px[u];
      if (VAR & 8) { v.x = bsx_mnist_pixel_value(p, 0); v.y = bsx_mnist_pixel_value(p, 1); v.z = bsx_mnist_pixel_value(p, 2); v.w = bsx_mnist_pixel_value(p, 3); }
      else { v.x = s_lut[p & 0xFF]; v.y = s_lut[(p >> 8) & 0xFF]; v.z = s_lut[(p >> 16) & 0xFF]; v.w = s_lut[p >> 24]; }
    }
    o4[c] = v;
  }
}

#endif  // BSX_MNIST_FAM_H_
