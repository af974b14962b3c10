// bsx_device.h — device-side building blocks shared by every environment-family kernel.
//
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, 256-thread workgroups (4 waves = one per
// SIMD), LDS for per-block constants and hot-cell indices, 16-byte cooperative stores for the
// observation stream.  No MFMA anywhere: the path is integer indexing + scalar f32/f64.
#ifndef BSX_DEVICE_H_
#define BSX_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bsuite_amd.h"
#include "../../include/bsx_stream.h"

#define BSX_BLOCK 256
#define BSX_WAVE 64

typedef float bsx_f4 __attribute__((ext_vector_type(4)));

// Per-call values every kernel needs, flattened out of bsx_call_t on the host.
struct bsx_ctl {
  int64_t n_lanes;
  uint64_t seed;
  uint64_t lane_offset;
  uint64_t step_index;
  const uint64_t* step_base;
  uint64_t* counters;
  double wrap_param;
  uint64_t wrap_seed;
  int32_t wrap_kind;
  int32_t force_reset;
};

__device__ __forceinline__ uint64_t bsx_step_of(const bsx_ctl& c) {
  return c.step_index + (c.step_base ? *c.step_base : 0ull);
}

// Reward epilogue of utils/wrappers.py:275-283 (RewardNoise) and :338-346 (RewardScale): non-FIRST
// lanes only, evaluated in f64 like the reference, result cast to f32 once.
__device__ __forceinline__ double bsx_wrap_reward(const bsx_ctl& c, uint64_t lane, uint64_t step,
                                                  double reward) {
  BSX_NO_CONTRACT
  if (c.wrap_kind == BSX_WRAP_SCALE) return reward * c.wrap_param;
  if (c.wrap_kind == BSX_WRAP_NOISE) {
    bsx_draws w;
    bsx_draws_init(&w, c.wrap_seed, lane, step, BSX_STREAM_WRAP);
    return reward + c.wrap_param * bsx_normal(&w);
  }
  return reward;
}

// Writes the scalar TimeStep fields of one lane (coalesced: lane i -> element i of each column).
__device__ __forceinline__ void bsx_emit(const bsx_ctl& c, const bsx_timestep_t& out, int64_t i,
                                         uint64_t lane, uint64_t step, int type, double reward) {
  float r = 0.0f, d = 1.0f;   // FIRST: dm_env.restart has reward/discount None -> 0 / 1 in a batch
  if (type != BSX_FIRST) {
    r = (float)bsx_wrap_reward(c, lane, step, reward);
    d = (type == BSX_LAST) ? 0.0f : 1.0f;
  }
  out.reward[i] = r;
  out.discount[i] = d;
  out.step_type[i] = (int8_t)type;
}

// Termination / restart masks by wavefront ballot: one popcount + one atomic per wave, never per
// lane.  Inactive lanes (beyond n_lanes) must pass type = -1.
__device__ __forceinline__ void bsx_count_types(const bsx_ctl& c, int type) {
  if (c.counters == nullptr) return;
  unsigned long long last = __ballot(type == BSX_LAST);
  unsigned long long first = __ballot(type == BSX_FIRST);
  if ((threadIdx.x & (BSX_WAVE - 1)) == 0) {
    if (last) atomicAdd((unsigned long long*)&c.counters[0], (unsigned long long)__popcll(last));
    if (first) atomicAdd((unsigned long long*)&c.counters[1], (unsigned long long)__popcll(first));
  }
}

// Cooperative one-/two-hot observation tile writer (deep_sea, catch).
//
// The block owns `lanes_here` consecutive lanes, i.e. one contiguous run of lanes_here*cells floats
// starting at `tile` (16-byte aligned because lanes-per-block is a multiple of 4).  Consecutive
// threads own consecutive 16-byte chunks, so every wave store instruction covers 1 KiB of
// contiguous HBM.  hot_a/hot_b (LDS) hold each lane's flat hot-cell index or -1.
template <bool TWO_HOT>
__device__ __forceinline__ void bsx_write_hot_tile(float* __restrict__ tile, int lanes_here,
                                                   uint32_t cells, uint32_t cells_magic,
                                                   const int* hot_a, const int* hot_b) {
  const uint32_t total = (uint32_t)lanes_here * cells;          // floats in this block's tile
  const uint32_t n_chunks = total >> 2;
  bsx_f4* __restrict__ t4 = reinterpret_cast<bsx_f4*>(tile);
  if ((cells & 3u) == 0) {
    // a chunk never straddles two lanes
    for (uint32_t ch = threadIdx.x; ch < n_chunks; ch += BSX_BLOCK) {
      uint32_t f0 = ch << 2;
      uint32_t l = __umulhi(f0, cells_magic);
      int r0 = (int)(f0 - l * cells);
      int da = hot_a[l] - r0;
      bsx_f4 v;
      if (TWO_HOT) {
        int db = hot_b[l] - r0;
        v.x = (da == 0 || db == 0) ? 1.0f : 0.0f;
        v.y = (da == 1 || db == 1) ? 1.0f : 0.0f;
        v.z = (da == 2 || db == 2) ? 1.0f : 0.0f;
        v.w = (da == 3 || db == 3) ? 1.0f : 0.0f;
      } else {
        v.x = (da == 0) ? 1.0f : 0.0f;
        v.y = (da == 1) ? 1.0f : 0.0f;
        v.z = (da == 2) ? 1.0f : 0.0f;
        v.w = (da == 3) ? 1.0f : 0.0f;
      }
      t4[ch] = v;
    }
  } else {
    for (uint32_t ch = threadIdx.x; ch < n_chunks; ch += BSX_BLOCK) {
      uint32_t f0 = ch << 2;
      float e[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t f = f0 + j;
        uint32_t l = __umulhi(f, cells_magic);
        int r = (int)(f - l * cells);
        bool on = hot_a[l] == r;
        if (TWO_HOT) on = on || hot_b[l] == r;
        e[j] = on ? 1.0f : 0.0f;
      }
      bsx_f4 v = {e[0], e[1], e[2], e[3]};
      t4[ch] = v;
    }
    // ragged tail (< 4 floats) of an odd-sized tile
    uint32_t f = (n_chunks << 2) + threadIdx.x;
    if (f < total) {
      uint32_t l = __umulhi(f, cells_magic);
      int r = (int)(f - l * cells);
      bool on = hot_a[l] == r;
      if (TWO_HOT) on = on || hot_b[l] == r;
      tile[f] = on ? 1.0f : 0.0f;
    }
  }
}

#endif  // BSX_DEVICE_H_
