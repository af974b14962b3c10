/* Test shim: the packed wide rows of memory_chain / umbrella_chain (bsuite_amd/csrc/bsx_rows.h, the header the HIP
 * kernels compile) evaluated on the host by gcc — the store stream of csrc/row_stream.h walked workgroup by workgroup,
 * thread by thread, with the device's index arithmetic restated beside it. */
#include <stdint.h>
#include <string.h>
#include "../../bsuite_amd/csrc/bsx_rows.h"

int shim_row_words(int numel, int kind) { return bsx_row_words_of(numel, kind); }
int shim_plane_words(int numel, int kind) { return bsx_row_plane_words(numel, kind); }

/* Packs one lane's row the way the lane's thread does: HEAD floats, then bsx_row_put for every 32-bit piece of every
 * plane.  bits0 / bits1: the lane's plane bits, one per byte (nbits = numel - HEAD). */
void shim_pack_row(uint32_t* row, int numel, int kind, const uint32_t* head, const uint8_t* bits0, const uint8_t* bits1) {
  const int HEAD = bsx_rows_head(kind), nbits = numel - HEAD, W = bsx_row_plane_words(numel, kind);
  for (int k = 0; k < HEAD; ++k) row[k] = head[k];
  for (int p = 0; p < bsx_rows_planes(kind); ++p) {
    const uint8_t* bits = p ? bits1 : bits0;
    for (int k = 0; 32 * k < nbits; ++k) {
      const int n = nbits - 32 * k < 32 ? nbits - 32 * k : 32;
      uint32_t w = 0xFFFFFFFFu << (n < 32 ? n : 0);          /* garbage above the n bits: put() must mask it */
      if (n == 32) w = 0;
      for (int b = 0; b < n; ++b) w |= (uint32_t)(bits[32 * k + b] & 1u) << b;
      bsx_row_put(row + HEAD, (uint32_t)W, p, k, w, n);
    }
  }
}

/* The whole store stream over [n_lanes x numel] floats with K chunks per thread and 256 threads per workgroup:
 * out[f] = f32 bit pattern of flat element f.  `magic` / (dv_m, dv_s) are the host-built division magics the kernel
 * gets (bsx_div_magic / bsx_make_div64), so the test covers them too.  Returns the number of workgroups. */
int64_t shim_row_stream(const uint32_t* rows, int64_t n_lanes, int numel, int kind, int K, uint32_t magic,
                        uint64_t dv_m, uint32_t dv_s, uint32_t* out) {
  const uint32_t RW = (uint32_t)bsx_row_words_of(numel, kind), W = (uint32_t)bsx_row_plane_words(numel, kind);
  const uint64_t total = (uint64_t)n_lanes * (uint64_t)numel;
  const uint64_t per_block = (uint64_t)K * 4u * 256u;
  const int64_t blocks = (int64_t)((total + per_block - 1) / per_block);
  for (int64_t blk = 0; blk < blocks; ++blk) {
    const uint64_t F0 = (uint64_t)blk * per_block;
    const uint64_t lane_b = (uint64_t)(((unsigned __int128)F0 * dv_m) >> 64) >> dv_s;
    const uint32_t r_b = (uint32_t)(F0 - lane_b * (uint64_t)numel);
    const uint32_t* base = rows + lane_b * RW;
    for (uint32_t tid = 0; tid < 256u; ++tid)
      for (int u = 0; u < K; ++u) {
        const uint32_t c = (tid >> 6) * (uint32_t)(K * 64) + (uint32_t)u * 64u + (tid & 63u);
        const uint32_t f = r_b + (c << 2);
        const uint32_t dl = (uint32_t)(((uint64_t)f * magic) >> 32);
        if (!(F0 + ((uint64_t)c << 2) + 3 < total)) continue;
        const bsx_row_chunk_in q = bsx_row_chunk_load(base + (uint64_t)dl * RW, RW, W, (uint32_t)numel, f - dl * (uint32_t)numel, kind);
        uint32_t* o = out + F0 + ((uint64_t)c << 2);
        bsx_row_chunk_decode(&q, (uint32_t)numel, kind, &o[0], &o[1], &o[2], &o[3]);
      }
    const uint64_t tail0 = total & ~3ull;
    if (tail0 != total && tail0 >= F0 && tail0 < F0 + per_block)
      for (uint32_t tid = 0; tid < 3u; ++tid) {
        const uint64_t F = tail0 + tid;
        if (F < total) {
          const uint32_t f = r_b + (uint32_t)(F - F0);
          const uint32_t d = (uint32_t)(((uint64_t)f * magic) >> 32);
          out[F] = bsx_row_element(base + (uint64_t)d * RW, f - d * (uint32_t)numel, W, kind);
        }
      }
  }
  return blocks;
}
