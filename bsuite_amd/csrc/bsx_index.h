// bsx_index.h — workgroup index arithmetic shared by the kernels.  Plain C99 + BSX_HD so that the CPU
// tests compile the very same code with gcc (tests/csrc/index_shim.c): a wrong index here is silent wrong
// output on the device.
#ifndef BSX_INDEX_H_
#define BSX_INDEX_H_

#include <stdint.h>

#include "../../include/bsx_stream.h"   // BSX_HD

// Which workgroups of a pipelined launch advance lanes: `place` 0 = the first adv_blocks of the grid (default),
// 1 = the last, 2 = spread evenly through the grid (one every grid/adv_blocks workgroups).  Measured
// (profiles/r02/ab_pipelined_rollout.log, ab_sweep_pipelined.log): first is best; last leaves the latency-bound
// advance alone at the end of the launch; spread is far worse than no pipelining at all (catch 54 us against 43.5,
// deep_sea 625 against 594) — anything that interrupts the address-ordered store stream costs more than it hides.
// Whatever the placement, the advance workgroups get indices 0..adv_blocks-1 and the stream workgroups
// 0..grid-adv_blocks-1, each in grid order.
typedef struct { int adv; uint32_t index; } bsx_pipe_role;
BSX_HD bsx_pipe_role bsx_pipe_role_of(uint32_t b, uint32_t grid, uint32_t adv_blocks, uint32_t place) {
  bsx_pipe_role r;
  if (place == 0u) { r.adv = b < adv_blocks; r.index = r.adv ? b : b - adv_blocks; return r; }
  const uint32_t str_blocks = grid - adv_blocks;
  if (place == 1u) { r.adv = b >= str_blocks; r.index = r.adv ? b - str_blocks : b; return r; }
  const uint32_t every = grid / (adv_blocks > 0u ? adv_blocks : 1u);     // >= 1; uniform
  const uint32_t j = b / every;
  r.adv = j < adv_blocks && b == j * every;
  uint32_t before = (b + every - 1u) / every;                            // advance workgroups in front of b
  before = before < adv_blocks ? before : adv_blocks;
  r.index = r.adv ? j : b - before;
  return r;
}

// Rows a lane's own thread stores (at most 8 floats: 8-byte stores for an even length, one 12-byte store for 3, 4-byte
// stores for 1, 5, 7); longer rows of the families with a parametric row length go through the bit-plane tile
// (small_obs.h), whose lane-owned HEAD chunks need numel >= 9 to be disjoint between neighbouring lanes.
BSX_HD int bsx_small_direct_shape(int numel) { return numel <= 8; }

// Split of one 32-bit piece of a lane's bit string for the flat bit planes of a tile: the piece starts at flat
// bit `pos` and has n bits (1..32, the low n bits of w).  Word index, the part that goes into that word and the
// part (possibly empty: *has_hi = 0) that spills into the next one.
BSX_HD void bsx_plane_split(uint32_t pos, uint32_t w, int n, uint32_t* word, uint32_t* lo, uint32_t* hi, int* has_hi) {
  if (n < 32) w &= (1u << n) - 1u;
  const uint32_t sh = pos & 31u;
  *word = pos >> 5;
  *lo = w << sh;
  *has_hi = sh + (uint32_t)n > 32u;
  *hi = *has_hi ? w >> (32u - sh) : 0u;
}

#endif  // BSX_INDEX_H_
