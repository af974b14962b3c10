/* Test shim: the flat bit planes of the chains' wide rows (bsuite_amd/csrc/bsx_rows.h, the header the HIP kernels
 * compile) evaluated on the host by gcc — the store stream of csrc/row_stream.h walked workgroup by workgroup, thread
 * by thread, with the device's index arithmetic restated beside it. */
#include <stdint.h>
#include <string.h>
#include "../../bsuite_amd/csrc/bsx_rows.h"

int64_t shim_scratch_words(int kind, int64_t n_lanes, int numel) { return (int64_t)bsx_rows_scratch_words(kind, n_lanes, numel); }
int64_t shim_plane_words(int64_t n_lanes, int numel) { return (int64_t)bsx_rows_plane_words(n_lanes, numel); }
int shim_planes(int kind) { return bsx_rows_planes(kind); }
int shim_nf(int kind) { return bsx_rows_nf(kind); }
int shim_fpos(int kind, int k) { return (int)bsx_rows_fpos(kind, k); }

/* The whole store stream over [n_lanes x numel] floats with K chunks per thread and 256 threads per workgroup:
 * out[f] = f32 bit pattern of flat element f.  `magic` / (dv_m, dv_s) are the host-built division magics the kernel
 * gets (bsx_div_magic / bsx_make_div64), so the test covers them too.  Returns the number of workgroups. */
int64_t shim_row_stream(const uint32_t* scratch, int64_t n_lanes, int numel, int kind, int K, uint32_t magic,
                        uint64_t dv_m, uint32_t dv_s, uint32_t* out) {
  const uint64_t pw = bsx_rows_plane_words(n_lanes, numel);
  const int PL = bsx_rows_planes(kind), NF = bsx_rows_nf(kind);
  const uint64_t total = (uint64_t)n_lanes * (uint64_t)numel;
  const uint64_t per_block = (uint64_t)K * 4u * 256u;
  const int64_t blocks = (int64_t)((total + per_block - 1) / per_block);
  for (int64_t blk = 0; blk < blocks; ++blk) {
    const uint64_t F0 = (uint64_t)blk * per_block;
    const uint64_t lane_b = (uint64_t)(((unsigned __int128)F0 * dv_m) >> 64) >> dv_s;
    const uint32_t r_b = (uint32_t)(F0 - lane_b * (uint64_t)numel);
    const uint32_t* pl = scratch + (F0 >> 5);
    const uint32_t* hd = scratch + (uint64_t)PL * pw + lane_b;
    for (uint32_t tid = 0; tid < 256u; ++tid)
      for (int u = 0; u < K; ++u) {
        const uint32_t c = (tid >> 6) * (uint32_t)(K * 64) + (uint32_t)u * 64u + (tid & 63u);
        const uint32_t f = r_b + (c << 2);
        const uint32_t dl = (uint32_t)(((uint64_t)f * magic) >> 32);
        const uint32_t t = f - dl * (uint32_t)numel;
        if (!(F0 + ((uint64_t)c << 2) + 3 < total)) continue;
        const uint32_t w0 = pl[c >> 3], w1 = PL > 1 ? pl[pw + (c >> 3)] : 0u;
        uint32_t j0 = 4u, j1 = 4u, h0 = 0xDEADBEEFu, h1 = 0xDEADBEEFu, nx;
        j0 = bsx_rows_head_slot(bsx_rows_fpos(kind, 0), t, (uint32_t)numel, &nx);
        if (j0 < 4u) h0 = hd[dl + nx];
        if (NF > 1) {
          j1 = bsx_rows_head_slot(bsx_rows_fpos(kind, 1), t, (uint32_t)numel, &nx);
          if (j1 < 4u) h1 = hd[(uint64_t)n_lanes + dl + nx];
        }
        const uint32_t sh = (c & 7u) << 2;
        uint32_t* o = out + F0 + ((uint64_t)c << 2);
        bsx_rows_chunk(kind, w0 >> sh, w1 >> sh, j0, h0, j1, h1, &o[0], &o[1], &o[2], &o[3]);
      }
    const uint64_t tail0 = total & ~3ull;
    if (tail0 != total && tail0 >= F0 && tail0 < F0 + per_block)
      for (uint32_t tid = 0; tid < 3u; ++tid) {
        const uint64_t F = tail0 + tid;
        if (F < total) {
          const uint32_t f = r_b + (uint32_t)(F - F0);
          const uint32_t d = (uint32_t)(((uint64_t)f * magic) >> 32);
          out[F] = bsx_rows_element(kind, scratch, pw, scratch + (uint64_t)PL * pw, n_lanes, F, lane_b + d, f - d * (uint32_t)numel);
        }
      }
  }
  return blocks;
}
