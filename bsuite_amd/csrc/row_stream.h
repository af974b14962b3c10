// row_stream.h — the wide observation rows of memory_chain / umbrella_chain as lane advance + barrier-free store stream
// (ABI v12, bsx_call_t.row_scratch): the architecture that won for deep_sea / catch, for the two families whose row
// (up to 256 floats) is a few HEAD floats followed by elements that one or two BITS encode.
//
//   bsuite/environments/memory_chain.py:60-70    obs = [1 - t/L, query (t == L-1), +-1 by context bit (t == 0) ...]
//   bsuite/environments/umbrella_chain.py:60-66  obs = [need, has, 1 - t/L, Bernoulli(0.5) x n_distractor]
//
// The lane's own thread (small_obs_body, ROWS) leaves the row in PACKED form in the caller's scratch —
//     row[0 .. HEAD)                      the HEAD floats, as their f32 bit patterns
//     row[HEAD + p*W + k], k < W          word k of bit plane p (bit b of a plane belongs to element HEAD + b)
// `row_words` = HEAD + PLANES*W rounded up to a multiple of 4 uint32 (rows start on 16-byte boundaries), at most 48
// bytes per lane and L2-resident — and a pure store stream over the [B x numel] observation array decodes it: workgroup
// b writes floats [b*K*1024, (b+1)*K*1024) as K 16-byte chunks per thread, no loop, no LDS, no barrier (the shape of
// bsx_hot_stream_body).  What it replaces (small_obs_body, PACKED): a 256-lane tile as bit planes in LDS, every lane
// ORing its bits in with LDS atomics, three workgroup barriers per step and the stores of a workgroup all issued behind
// the last of them: umbrella_length 30.4 us for 118 MB at 2^20 lanes, and the workgroups that lived longest (up to
// 20 us) in the sweep's latency-bound phase 0 (profiles/r04/bench_all_workloads_eager.log, sweep_phase0_life.json).
#ifndef BSX_ROW_STREAM_H_
#define BSX_ROW_STREAM_H_

#include "bsx_device.h"
#include "bsx_rows.h"

// HEAD floats, bit planes and the element decoder of the two families (also used by the LDS bit-plane path, small_obs.h).
struct memory_rows {
  static constexpr int KIND = BSX_ROWS_MEMORY, HEAD = 2, PLANES = 2;
  __device__ static __forceinline__ float decode(uint32_t nonzero, uint32_t bit) { return __uint_as_float(bsx_rows_decode(KIND, nonzero, bit)); }
};
struct umbrella_rows {
  static constexpr int KIND = BSX_ROWS_UMBRELLA, HEAD = 3, PLANES = 1;
  __device__ static __forceinline__ float decode(uint32_t bit, uint32_t) { return __uint_as_float(bsx_rows_decode(KIND, bit, 0u)); }
};

// A lane's handle on its packed row: what Env::step hands its bits to (the LDS path's bsx_bit_sink has the same put()).
struct bsx_row_sink {
  uint32_t* __restrict__ planes;       // row + HEAD
  uint32_t w_words;                    // W
  // nothing zero-fills a row: every word of every plane is put on every call
  static constexpr bool ALWAYS = true;
  __device__ __forceinline__ void put(int p, int k, uint32_t w, int n) const { bsx_row_put(planes, w_words, p, k, w, n); }
};

struct bsx_row_seg {                   // one segment's arguments of the wide-row observation stream
  float* obs;
  const uint32_t* rows;                // [n_lanes, row_words]
  int64_t n_lanes;
  uint32_t numel;
  uint32_t numel_magic;                // bsx_div_magic(numel)
  bsx_div64 dv;                        // bsx_make_div64(numel)
  uint32_t row_words;
  uint32_t w_words;
};

template <class R, int K>
__device__ __forceinline__ void bsx_row_stream_body(const bsx_row_seg& g, uint32_t block_id) {
  static_assert(R::HEAD == (R::KIND == BSX_ROWS_MEMORY ? 2 : 3) && R::PLANES == (R::KIND == BSX_ROWS_MEMORY ? 2 : 1), "bsx_rows.h");
  const uint32_t numel = g.numel, W = g.w_words, RW = g.row_words;      // numel >= 9: a chunk touches at most two rows
  const uint64_t total = (uint64_t)g.n_lanes * numel;
  const uint64_t F0 = (uint64_t)block_id * (uint64_t)(K * 4 * BSX_BLOCK);
  const uint64_t lane_b = __umul64hi(F0, g.dv.m) >> g.dv.s;             // uniform
  const uint32_t r_b = (uint32_t)(F0 - lane_b * numel);
  bsx_f4* __restrict__ o4 = reinterpret_cast<bsx_f4*>(g.obs + F0);
  const uint32_t* __restrict__ rows = g.rows + lane_b * RW;

  bsx_row_chunk_in q[K];
  bool live[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    // each wave owns K consecutive KiB (bsx_hot_stream_body); every chunk's loads are issued before the first store
    const uint32_t c = (threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63);
    const uint32_t f = r_b + (c << 2);
    const uint32_t dl = __umulhi(f, g.numel_magic);
    // (a live chunk lies inside the array, so the row it runs over into exists)
    live[u] = F0 + ((uint64_t)c << 2) + 3 < total;
    if (live[u]) q[u] = bsx_row_chunk_load(rows + (uint64_t)dl * RW, RW, W, numel, f - dl * numel, R::KIND);
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (!live[u]) continue;
    const uint32_t c = (threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63);
    uint32_t v0, v1, v2, v3;
    bsx_row_chunk_decode(&q[u], numel, R::KIND, &v0, &v1, &v2, &v3);
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
      g.obs[F] = __uint_as_float(bsx_row_element(rows + (uint64_t)d * RW, f - d * numel, W, R::KIND));
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
