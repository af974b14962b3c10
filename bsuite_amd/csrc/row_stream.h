// row_stream.h — the wide observation rows of memory_chain / umbrella_chain as lane advance + barrier-free store stream
// (ABI v12, bsx_call_t.row_scratch): the architecture that won for deep_sea / catch, for the two families whose row
// (up to 256 floats) is a few HEAD floats followed by elements that one or two BITS encode.
//
//   bsuite/environments/memory_chain.py:60-70    obs = [1 - t/L, query (t == L-1), +-1 by context bit (t == 0) ...]
//   bsuite/environments/umbrella_chain.py:60-66  obs = [need, has, 1 - t/L, Bernoulli(0.5) x n_distractor]
//
// The lane advance (small_obs_body, ROWS) is the one-launch kernel's advance without its workgroup barriers: every WAVE
// ORs its 64 lanes' bits into flat bit planes in wave-private LDS (in-order LDS: no s_barrier) and stores them — 2 * numel
// whole words per plane — into the call's scratch, next to one f32 column per genuine float of the row (bsx_rows.h has
// the layout).  This file is the second launch: a pure store stream over the [B x numel] observation array — workgroup b
// writes floats [b*K*1024, (b+1)*K*1024) as K 16-byte chunks per thread, no loop, no LDS, no barrier (the shape of
// bsx_hot_stream_body) — in which a chunk is ONE aligned nibble of each plane, a few bit tests away from a float4, plus the
// float heads that fall into it (the only place the row structure shows: one magic division per chunk).
// What it replaces (small_obs_body, PACKED): a 256-lane tile as bit planes in LDS, three workgroup barriers per step and
// the stores of a workgroup all issued behind the last of them.
#ifndef BSX_ROW_STREAM_H_
#define BSX_ROW_STREAM_H_

#include "bsx_device.h"
#include "bsx_rows.h"

// Bit planes, float heads and the element decoder of the two families (also used by the LDS bit-plane path, small_obs.h,
// whose HEAD floats are the leading elements of a row: memory_chain [time, query], umbrella_chain [need, has, time]).
struct memory_rows {
  static constexpr int KIND = BSX_ROWS_MEMORY, HEAD = 2, PLANES = 2, NF = 2;
  __device__ static __forceinline__ float decode(uint32_t nonzero, uint32_t bit) { return __uint_as_float(bsx_rows_decode(KIND, nonzero, bit)); }
};
struct umbrella_rows {
  static constexpr int KIND = BSX_ROWS_UMBRELLA, HEAD = 3, PLANES = 1, NF = 1;
  __device__ static __forceinline__ float decode(uint32_t bit, uint32_t) { return __uint_as_float(bsx_rows_decode(KIND, bit, 0u)); }
};

struct bsx_row_seg {                   // one segment's arguments of the wide-row observation stream
  float* obs;
  const uint32_t* planes;              // the scratch: PLANES planes of `plane_words`, then NF float-head columns [n_lanes]
  int64_t n_lanes;
  uint64_t plane_words;
  uint32_t numel;
  uint32_t numel_magic;                // bsx_div_magic(numel)
  bsx_div64 dv;                        // bsx_make_div64(numel)
};

template <class R, int K>
__device__ __forceinline__ void bsx_row_stream_body(const bsx_row_seg& g, uint32_t block_id) {
  static_assert(R::PLANES == (R::KIND == BSX_ROWS_MEMORY ? 2 : 1) && R::NF == (R::KIND == BSX_ROWS_MEMORY ? 2 : 1), "bsx_rows.h");
  const uint32_t numel = g.numel;                                       // numel >= 9: a chunk touches at most two rows
  const uint64_t total = (uint64_t)g.n_lanes * numel;
  const uint64_t F0 = (uint64_t)block_id * (uint64_t)(K * 4 * BSX_BLOCK);
  const uint64_t lane_b = __umul64hi(F0, g.dv.m) >> g.dv.s;             // uniform: the lane of the workgroup's first float
  const uint32_t r_b = (uint32_t)(F0 - lane_b * numel);
  bsx_f4* __restrict__ o4 = reinterpret_cast<bsx_f4*>(g.obs + F0);
  const uint32_t* __restrict__ pl = g.planes + (F0 >> 5);               // the workgroup's first plane word (F0 is a multiple of 1024)
  const uint32_t* __restrict__ hd = g.planes + (uint64_t)R::PLANES * g.plane_words + lane_b;   // float head 0 of lane lane_b

  uint32_t w0[K], w1[K], j0[K], j1[K], h0[K], h1[K];
  bool live[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    // each wave owns K consecutive KiB (bsx_hot_stream_body); every chunk's loads are issued before the first store
    const uint32_t c = (threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63);
    const uint32_t f = r_b + (c << 2);
    const uint32_t dl = __umulhi(f, g.numel_magic);
    const uint32_t t = f - dl * numel;
    live[u] = F0 + ((uint64_t)c << 2) + 3 < total;                      // (a live chunk lies inside the array: the row it runs over into exists)
    w0[u] = 0u; w1[u] = 0u; j0[u] = 4u; j1[u] = 4u; h0[u] = 0u; h1[u] = 0u;
    if (live[u]) {
      w0[u] = pl[c >> 3];
      if (R::PLANES > 1) w1[u] = pl[g.plane_words + (c >> 3)];
      uint32_t nx;
      j0[u] = bsx_rows_head_slot(bsx_rows_fpos(R::KIND, 0), t, numel, &nx);
      if (j0[u] < 4u) h0[u] = hd[dl + nx];
      if (R::NF > 1) {
        j1[u] = bsx_rows_head_slot(bsx_rows_fpos(R::KIND, 1), t, numel, &nx);
        if (j1[u] < 4u) h1[u] = hd[(uint64_t)g.n_lanes + dl + nx];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (!live[u]) continue;
    const uint32_t c = (threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63);
    const uint32_t sh = (c & 7u) << 2;
    uint32_t v0, v1, v2, v3;
    bsx_rows_chunk(R::KIND, w0[u] >> sh, w1[u] >> sh, j0[u], h0[u], j1[u], h1[u], &v0, &v1, &v2, &v3);
    bsx_f4 q4;
    q4.x = __uint_as_float(v0); q4.y = __uint_as_float(v1); q4.z = __uint_as_float(v2); q4.w = __uint_as_float(v3);
    o4[c] = q4;
  }
  // ragged tail (< 4 floats) of an odd-sized array: the workgroup that contains the array's end
  const uint64_t tail0 = total & ~3ull;
  if (tail0 != total && tail0 >= F0 && tail0 < F0 + (uint64_t)(K * 4 * BSX_BLOCK) && threadIdx.x < 3) {
    const uint64_t F = tail0 + threadIdx.x;
    if (F < total) {
      const uint32_t f = r_b + (uint32_t)(F - F0);
      const uint32_t d = __umulhi(f, g.numel_magic);
      g.obs[F] = __uint_as_float(bsx_rows_element(R::KIND, g.planes, g.plane_words, g.planes + (uint64_t)R::PLANES * g.plane_words,
                                                  g.n_lanes, F, lane_b + d, f - d * numel));
    }
  }
}

template <class R, int K>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_row_stream_kernel(const bsx_row_seg g) {
  bsx_row_stream_body<R, K>(g, blockIdx.x);
}

// chunks per thread of the wide-row store stream (a workgroup writes K x 4 KiB)
#ifndef BSX_ROW_STREAM_K
#define BSX_ROW_STREAM_K 2
#endif

#endif  // BSX_ROW_STREAM_H_
