/* Test shim: the workgroup index arithmetic of bsuite_amd/csrc/bsx_index.h (the header the HIP kernels
 * compile), evaluated on the host by gcc. */
#include <stdint.h>
#include <string.h>
#include "../../bsuite_amd/csrc/bsx_index.h"

void shim_pipe_roles(uint32_t grid, uint32_t adv_blocks, uint32_t place, int32_t* adv, uint32_t* index) {
  for (uint32_t b = 0; b < grid; ++b) {
    const bsx_pipe_role r = bsx_pipe_role_of(b, grid, adv_blocks, place);
    adv[b] = r.adv;
    index[b] = r.index;
  }
}

int shim_direct_shape(int numel) { return bsx_small_direct_shape(numel); }

/* One 256-lane tile the way small_obs_body builds and reads it: every lane ORs its `nbits` bits (bits[l*words_per_lane
 * + k], 32 to a word) into the flat plane at bit l*numel + head; then element f of the tile is bit f of the plane.
 * out[f] = that bit for every f in [0, lanes*numel) (head positions stay 0). */
void shim_plane_roundtrip(int lanes, int numel, int head, int nbits, const uint32_t* bits, int words_per_lane,
                          uint32_t* plane, int plane_words, uint8_t* out) {
  memset(plane, 0, (size_t)plane_words * 4);
  for (int l = 0; l < lanes; ++l) {
    const uint32_t base = (uint32_t)(l * numel + head);
    for (int k = 0; 32 * k < nbits; ++k) {
      const int n = nbits - 32 * k < 32 ? nbits - 32 * k : 32;
      uint32_t word, lo, hi;
      int has_hi;
      bsx_plane_split(base + 32u * (uint32_t)k, bits[l * words_per_lane + k], n, &word, &lo, &hi, &has_hi);
      plane[word] |= lo;
      if (has_hi) plane[word + 1] |= hi;
    }
  }
  for (int f = 0; f < lanes * numel; ++f) out[f] = (uint8_t)((plane[f >> 5] >> (f & 31)) & 1u);
}
