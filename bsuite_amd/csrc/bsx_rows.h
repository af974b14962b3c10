// bsx_rows.h — the PACKED form of a wide observation row of memory_chain / umbrella_chain and its decoder, chunk by
// chunk (ABI v12, bsx_call_t.row_scratch; the kernels around it: row_stream.h, small_obs.h).  Plain C99 + BSX_HD so that
// the CPU tests compile the very same code with gcc (tests/csrc/rows_shim.c): a wrong index here is silent wrong output
// on the device.
//
//   bsuite/environments/memory_chain.py:60-70    obs = [1 - t/L, query (t == L-1), +-1 by context bit (t == 0) ...]
//   bsuite/environments/umbrella_chain.py:60-66  obs = [need, has, 1 - t/L, Bernoulli(0.5) x n_distractor]
//
// A packed row of `row_words` uint32 (a multiple of 4: rows start on 16-byte boundaries):
//     row[0 .. HEAD)                 the HEAD floats, as their f32 bit patterns
//     row[HEAD + p*W + k], k < W     word k of bit plane p; bit b of a plane belongs to element HEAD + b
// with W = ceil((numel - HEAD) / 32).  At most 48 bytes per lane.
#ifndef BSX_ROWS_H_
#define BSX_ROWS_H_

#include <stdint.h>

#include "../../include/bsx_stream.h"   // BSX_HD

#define BSX_ROWS_UMBRELLA 0   // HEAD = [need, has, time]; element 3+b = distractor bit b as 0.0 / 1.0 (one plane)
#define BSX_ROWS_MEMORY 1     // HEAD = [time, query]; element 2+b = 0 unless t == 0, then +-1 by context bit b:
                              // plane 0 says "non-zero", plane 1 carries the context bit

BSX_HD int bsx_rows_head(int kind) { return kind == BSX_ROWS_MEMORY ? 2 : 3; }
BSX_HD int bsx_rows_planes(int kind) { return kind == BSX_ROWS_MEMORY ? 2 : 1; }
// the f32 bit pattern of a bit element (integer selects: no branches)
BSX_HD uint32_t bsx_rows_decode(int kind, uint32_t b0, uint32_t b1) {
  return kind == BSX_ROWS_MEMORY ? ((0u - b0) & (0xBF800000u ^ (b1 << 31))) : ((0u - b0) & 0x3F800000u);
}
// words per bit plane / per packed row of a row of `numel` floats
BSX_HD int bsx_row_plane_words(int numel, int kind) { return (numel - bsx_rows_head(kind) + 31) / 32; }
BSX_HD int bsx_row_words_of(int numel, int kind) {
  return (bsx_rows_head(kind) + bsx_rows_planes(kind) * bsx_row_plane_words(numel, kind) + 3) & ~3;
}

// What a 16-byte chunk — elements r0 .. r0+3 of lane A's row, running over into the first elements of the NEXT lane's
// row where r0 + 3 >= numel (numel >= 9: at most one row boundary in a chunk) — needs from the packed rows.  (Scalar
// members and straight-line code: an array member indexed in a loop kept the whole struct in scratch memory.)
typedef struct {
  uint32_t r0;
  uint32_t lo0, hi0;         // plane 0: the word that holds the chunk's first bit element, and the one after it
  uint32_t lo1, hi1;         // plane 1 (BSX_ROWS_MEMORY)
  uint32_t ha0, ha1, ha2;    // lane A's HEAD floats (when the chunk starts inside them)
  uint32_t hb0, hb1, hb2;    // the next lane's HEAD floats (when the chunk runs over)
  uint32_t nb0, nb1;         // ... and word 0 of its planes (HEAD < 3: its third element is a bit element)
} bsx_row_chunk_in;

// ra: lane A's packed row (the next lane's is ra + RW; read only when the chunk runs over).
BSX_HD bsx_row_chunk_in bsx_row_chunk_load(const uint32_t* ra, uint32_t RW, uint32_t W, uint32_t numel,
                                           uint32_t r0, int kind) {
  bsx_row_chunk_in v;
  bsx_row_chunk_in* q = &v;
  const uint32_t HEAD = (uint32_t)bsx_rows_head(kind);
  const int two = bsx_rows_planes(kind) > 1;
  // the chunk's (up to four) bit elements of lane A start at plane bit max(r0 - HEAD, 0)
  const uint32_t bs = r0 > HEAD ? r0 - HEAD : 0u;
  const uint32_t wi = bs >> 5, wj = wi + 1u < W ? wi + 1u : wi;
  q->r0 = r0;
  q->lo0 = ra[HEAD + wi]; q->hi0 = ra[HEAD + wj];
  q->lo1 = 0u; q->hi1 = 0u;
  if (two) { q->lo1 = ra[HEAD + W + wi]; q->hi1 = ra[HEAD + W + wj]; }
  q->ha0 = 0u; q->ha1 = 0u; q->ha2 = 0u;
  if (r0 < HEAD) { q->ha0 = ra[0]; q->ha1 = ra[1]; if (HEAD > 2u) q->ha2 = ra[2]; }
  q->hb0 = 0u; q->hb1 = 0u; q->hb2 = 0u; q->nb0 = 0u; q->nb1 = 0u;
  if (numel - r0 < 4u) {
    q->hb0 = ra[RW]; q->hb1 = ra[RW + 1u];
    if (HEAD > 2u) q->hb2 = ra[RW + 2u];
    else { q->nb0 = ra[RW + HEAD]; if (two) q->nb1 = ra[RW + HEAD + W]; }
  }
  return v;
}

// element j (0..3) of the chunk, as its f32 bit pattern
BSX_HD uint32_t bsx_row_chunk_element(const bsx_row_chunk_in* q, uint32_t numel, int kind, uint32_t nib0, uint32_t nib1,
                                      uint32_t bs, uint32_t j) {
  const uint32_t HEAD = (uint32_t)bsx_rows_head(kind);
  const uint32_t r = q->r0 + j;
  if (r < numel) {
    // (masks, not a chain of selects between neighbouring members: the compiler turns that into ONE load at a computed
    // address and the struct stays in scratch memory)
    if (r < HEAD) return (r == 0u ? q->ha0 : 0u) | (r == 1u ? q->ha1 : 0u) | (r == 2u ? q->ha2 : 0u);
    const uint32_t b = r - HEAD - bs;                                  // 0..3
    return bsx_rows_decode(kind, (nib0 >> b) & 1u, (nib1 >> b) & 1u);
  }
  const uint32_t rn = r - numel;                                       // 0..2 of the next lane's row
  if (rn < HEAD) return (rn == 0u ? q->hb0 : 0u) | (rn == 1u ? q->hb1 : 0u) | (rn == 2u ? q->hb2 : 0u);
  return bsx_rows_decode(kind, q->nb0 & 1u, q->nb1 & 1u);
}

// out[j] = the f32 bit pattern of the chunk's element j
BSX_HD void bsx_row_chunk_decode(const bsx_row_chunk_in* q, uint32_t numel, int kind, uint32_t* o0, uint32_t* o1,
                                 uint32_t* o2, uint32_t* o3) {
  const uint32_t HEAD = (uint32_t)bsx_rows_head(kind);
  const uint32_t bs = q->r0 > HEAD ? q->r0 - HEAD : 0u;
  const uint32_t sh = bs & 31u;
  // the 32 plane bits from bit `bs` on (v_alignbit_b32)
  const uint32_t nib0 = (uint32_t)(((((uint64_t)q->hi0) << 32) | (uint64_t)q->lo0) >> sh);
  const uint32_t nib1 = bsx_rows_planes(kind) > 1 ? (uint32_t)(((((uint64_t)q->hi1) << 32) | (uint64_t)q->lo1) >> sh) : 0u;
  *o0 = bsx_row_chunk_element(q, numel, kind, nib0, nib1, bs, 0u);
  *o1 = bsx_row_chunk_element(q, numel, kind, nib0, nib1, bs, 1u);
  *o2 = bsx_row_chunk_element(q, numel, kind, nib0, nib1, bs, 2u);
  *o3 = bsx_row_chunk_element(q, numel, kind, nib0, nib1, bs, 3u);
}

// element r of one packed row (ragged tails; the reference form the chunk decoder is tested against)
BSX_HD uint32_t bsx_row_element(const uint32_t* row, uint32_t r, uint32_t W, int kind) {
  const uint32_t HEAD = (uint32_t)bsx_rows_head(kind);
  if (r < HEAD) return row[r];
  const uint32_t b = r - HEAD;
  const uint32_t b0 = (row[HEAD + (b >> 5)] >> (b & 31u)) & 1u;
  const uint32_t b1 = bsx_rows_planes(kind) > 1 ? (row[HEAD + W + (b >> 5)] >> (b & 31u)) & 1u : 0u;
  return bsx_rows_decode(kind, b0, b1);
}

// A lane's thread hands the bits [32k, 32k + n) of plane p of its row (n in 1..32, the low n bits of w) to the packed row.
BSX_HD void bsx_row_put(uint32_t* planes /* row + HEAD */, uint32_t W, int p, int k, uint32_t w, int n) {
  planes[(uint32_t)p * W + (uint32_t)k] = n < 32 ? (w & ((1u << n) - 1u)) : w;
}

#endif  // BSX_ROWS_H_
