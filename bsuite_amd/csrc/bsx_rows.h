// bsx_rows.h — the wide observation rows of memory_chain / umbrella_chain as FLAT BIT PLANES in device memory, and their
// decoder, chunk by chunk (ABI v12, bsx_call_t.row_scratch; the kernels around it: row_stream.h, small_obs.h).  Plain C99
// + BSX_HD so that the CPU tests compile the very same code with gcc (tests/csrc/rows_shim.c): a wrong index here is
// silent wrong output on the device.
//
//   bsuite/environments/memory_chain.py:60-70    obs = [1 - t/L, query (t == L-1), +-1 by context bit (t == 0) ...]
//   bsuite/environments/umbrella_chain.py:60-66  obs = [need, has, 1 - t/L, Bernoulli(0.5) x n_distractor]
//
// The scratch of one call, n_lanes lanes of numel floats each:
//     plane p  (p < PLANES)   bsx_rows_plane_words(n_lanes, numel) uint32: bit e of a plane belongs to ELEMENT e of the
//                             flat [n_lanes x numel] observation array (e = lane * numel + r) — a 16-byte chunk of the
//                             observation array is one aligned nibble of each plane, whatever rows it touches.  A wave's
//                             64 lanes own 64 * numel bits = 2 * numel whole words of every plane.
//     float head k (k < NF)   n_lanes f32: the row elements that are genuine floats (the time fraction, memory_chain's
//                             query), one column each; their bits in the planes are 0.
// (Round 5's first form kept one packed record per lane — HEAD floats + the lane's own bit words, 16-48 bytes — and
// decoded it per ROW: every chunk paid for a division, the row boundary inside it and up to three selects per element,
// 125 vector instructions per chunk, and the stream ran at 4.0 TB/s: profiles/r05/ab_wide_rows_v1_packed_rows.log.)
#ifndef BSX_ROWS_H_
#define BSX_ROWS_H_

#include <stdint.h>

#include "../../include/bsx_stream.h"   // BSX_HD

#define BSX_ROWS_UMBRELLA 0   // [need, has, time, distractors...]: need / has / distractor b = plane-0 bits 0, 1, 3+b (0.0 / 1.0);
                              // float head 0 = time at position 2
#define BSX_ROWS_MEMORY 1     // [time, query, context...]: context element 2+b = 0 unless t == 0, then +-1 by context bit b:
                              // plane 0 says "non-zero", plane 1 carries the bit; float heads 0, 1 = time, query at 0, 1

BSX_HD int bsx_rows_planes(int kind) { return kind == BSX_ROWS_MEMORY ? 2 : 1; }
BSX_HD int bsx_rows_nf(int kind) { return kind == BSX_ROWS_MEMORY ? 2 : 1; }                      // float heads
BSX_HD uint32_t bsx_rows_fpos(int kind, int k) { return kind == BSX_ROWS_MEMORY ? (uint32_t)k : 2u; }   // ... and where they sit
// the f32 bit pattern of a bit-coded element (integer selects: no branches)
BSX_HD uint32_t bsx_rows_decode(int kind, uint32_t b0, uint32_t b1) {
  return kind == BSX_ROWS_MEMORY ? ((0u - b0) & (0xBF800000u ^ (b1 << 31))) : ((0u - b0) & 0x3F800000u);
}
// words of one plane: whole waves of 64 lanes, 2 * numel words each
BSX_HD uint64_t bsx_rows_plane_words(int64_t n_lanes, int numel) {
  return (uint64_t)((n_lanes + 63) / 64) * (uint64_t)(2 * numel);
}
// uint32 words of the whole scratch
BSX_HD uint64_t bsx_rows_scratch_words(int kind, int64_t n_lanes, int numel) {
  return (uint64_t)bsx_rows_planes(kind) * bsx_rows_plane_words(n_lanes, numel) + (uint64_t)bsx_rows_nf(kind) * (uint64_t)n_lanes;
}

// Float head at row position `pos`, seen from a chunk that starts at row offset t (0 <= t < numel; numel >= 9, so a chunk
// touches at most two rows and holds a given position at most once): the chunk element (0..3) that holds it — 4 or more:
// none — and whether that is the NEXT lane's row (the chunk runs over the end of the row it starts in).
BSX_HD uint32_t bsx_rows_head_slot(uint32_t pos, uint32_t t, uint32_t numel, uint32_t* next_lane) {
  *next_lane = t > pos ? 1u : 0u;
  return t > pos ? numel - t + pos : pos - t;
}

// The four elements of a chunk as f32 bit patterns: nib0 / nib1 = the chunk's nibble of plane 0 / 1 (bit j = element j),
// (j0, h0) / (j1, h1) = slot and value of float head 0 / 1 (slot >= 4: not in this chunk, value ignored).
BSX_HD void bsx_rows_chunk(int kind, uint32_t nib0, uint32_t nib1, uint32_t j0, uint32_t h0, uint32_t j1, uint32_t h1,
                           uint32_t* o0, uint32_t* o1, uint32_t* o2, uint32_t* o3) {
  uint32_t v0 = bsx_rows_decode(kind, nib0 & 1u, nib1 & 1u);
  uint32_t v1 = bsx_rows_decode(kind, (nib0 >> 1) & 1u, (nib1 >> 1) & 1u);
  uint32_t v2 = bsx_rows_decode(kind, (nib0 >> 2) & 1u, (nib1 >> 2) & 1u);
  uint32_t v3 = bsx_rows_decode(kind, (nib0 >> 3) & 1u, (nib1 >> 3) & 1u);
  v0 = j0 == 0u ? h0 : v0; v1 = j0 == 1u ? h0 : v1; v2 = j0 == 2u ? h0 : v2; v3 = j0 == 3u ? h0 : v3;
  if (bsx_rows_nf(kind) > 1) { v0 = j1 == 0u ? h1 : v0; v1 = j1 == 1u ? h1 : v1; v2 = j1 == 2u ? h1 : v2; v3 = j1 == 3u ? h1 : v3; }
  *o0 = v0; *o1 = v1; *o2 = v2; *o3 = v3;
}

// one element (ragged tails; the reference form the chunk decoder is tested against): e = its flat index, r = e mod numel
BSX_HD uint32_t bsx_rows_element(int kind, const uint32_t* planes, uint64_t plane_words, const uint32_t* heads, int64_t n_lanes,
                                 uint64_t e, uint64_t lane, uint32_t r) {
  int k;
  for (k = 0; k < bsx_rows_nf(kind); ++k)
    if (r == bsx_rows_fpos(kind, k)) return heads[(uint64_t)k * (uint64_t)n_lanes + lane];
  const uint32_t b0 = (planes[e >> 5] >> (e & 31u)) & 1u;
  const uint32_t b1 = bsx_rows_planes(kind) > 1 ? (planes[plane_words + (e >> 5)] >> (e & 31u)) & 1u : 0u;
  return bsx_rows_decode(kind, b0, b1);
}

#endif  // BSX_ROWS_H_
