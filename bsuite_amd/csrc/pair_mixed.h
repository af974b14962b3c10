// pair_mixed.h — host interface of the mixed groups of a heterogeneous sweep:
//   BSX_FAM_PAIR_MIXED   deep_sea + catch + mnist segments (pair_mixed.hip): one advance launch, one stream launch
//   BSX_FAM_SWEEP_MIXED  segments of ALL families: phase 0 = every lane of the sweep advanced by ONE launch
//                        (small_obs.hip, sweep_phase0_kernel), phase 1 = the store stream of pair_mixed.hip
#ifndef BSX_PAIR_MIXED_H_
#define BSX_PAIR_MIXED_H_

#include "bsx_host.h"
#include "catch_fam.h"
#include "deep_sea_fam.h"
#include "mnist_fam.h"
#include "row_stream.h"

#define BSX_MIXED_ADV_STRIDE 1024      // phase-0 argument slot (advance args of a pair family, or a small family's args)
#define BSX_MIXED_STR_STRIDE 1280      // phase-1 argument slot (observation stream args of a pair family)

// Records one segment in a BSX_FAM_PAIR_MIXED / BSX_FAM_SWEEP_MIXED group.  `adv`: the segment's phase-0
// argument struct; `str`: its observation-stream argument struct (NULL for the small-observation families,
// which have no phase 1); both are copied verbatim into fixed-stride slots of the group's device tables.
// `lds`: dynamic LDS the segment's phase-0 workgroups need (packed observation records, small_obs.hip).
int bsx_mixed_put(bsx_group* g, int32_t family, int32_t index, const bsx_call_t* call,
                  const void* adv, size_t adv_size, const void* str, size_t str_size,
                  uint64_t blocks1, uint64_t blocks2, size_t lds);

// launch of the mixed observation stream kernel (pair_mixed.hip) over a group's phase-1 tables
int bsx_mixed_launch_stream(bsx_group* g, hipStream_t st);
// phase 0 of a BSX_FAM_SWEEP_MIXED group (small_obs.hip)
int bsx_sweep_launch_phase0(bsx_group* g, hipStream_t st);
// the split closed-loop step of a whole-sweep group (sweep_mixed.hip)
int bsx_sweep_launch_split(bsx_group* g, hipStream_t st);
// one launch: the phase-1 store stream of `streams_of` beside phase 0 of `advances_of` (small_obs.hip)
int bsx_sweep_launch_pipelined(bsx_group* streams_of, bsx_group* advances_of, hipStream_t st);

// One workgroup of the mixed observation store stream: `block` of the phase-1 grid runs its segment's
// family stream body (deep_sea 4 x 4 KiB, catch 2 x 4 KiB, mnist 4 x 4 KiB, wide chain rows 2 x 4 KiB runs per workgroup).
// mnist: 4 KiB-runs per wave, its stand-alone optimum (mnist.hip) — but only TOGETHER with the non-temporal one-hot stores
// below.  Same call, three / four repetitions each, closed-loop sweep step (profiles/r06/ab_sweep_mnist_k.log,
// ab_nontemporal_stores*.log): with ordinary one-hot stores 4: 169-171 us, 5: 165-166, 6: 158.8-159.6, 7: 161-164, 8: 162-165
// (r05 body at 8: 161-162); with non-temporal one-hot stores 3: 165-168, **4: 153.8-157.5**, 5: 154-162, 6: 159.6-163.5 on one
// box and 157.8-161.9 (4) against 163.0-165.5 (6) and 162.9-164.4 (ordinary stores, 6) on a slower one.
#ifndef PAIR_MNIST_K
#define PAIR_MNIST_K 4
#endif
#ifndef PAIR_DEEP_SEA_K
#define PAIR_DEEP_SEA_K 4
#endif
#ifndef PAIR_CATCH_K
#define PAIR_CATCH_K 2
#endif
// non-temporal stores for the one-hot bodies of the MIXED stream (bsx_hot_stream_body<..., NT>): what the other workgroups of the
// launch and the next launch want to find in cache — state columns, action ring, tables, the small families' columns — is no
// longer evicted by 420 MB of deep_sea / catch observations per sweep step.  Not for the mnist body (its image gathers like
// ordinary neighbours: 165-171 us with nt on mnist alone, 157-169 with both; profiles/r06/ab_nontemporal_stores*.log)
#ifndef PAIR_HOT_NT
#define PAIR_HOT_NT true
#endif
// (The segments' pointers arrive through the argument table, so the compiler emits FLAT loads and stores here.  Typed as
// global memory — global_store_dwordx4, no lgkmcnt traffic — the closed-loop sweep step is SLOWER: 159.4-161.9 us against
// 151.2-159.5, same call, four repetitions; profiles/r06/ab_sweep_global_pointers.log.  Left as the compiler has it.)
__device__ __forceinline__ void pair_mixed_stream_body(const uint8_t* __restrict__ table, const int32_t* __restrict__ family,
                                                       const bsx_group_index& gi, uint32_t block, float* s_lut) {
  const bsx_group_slot w = bsx_group_find(gi, (int)block);
  const uint8_t* slot = table + (size_t)w.seg * BSX_MIXED_STR_STRIDE;
  switch (w.tag >= 0 ? w.tag : (family[w.seg] & 0xFF)) {   // uniform per workgroup
    case BSX_FAM_DEEP_SEA: {
      const bsx_stream_seg<deep_sea_hot>& g = *reinterpret_cast<const bsx_stream_seg<deep_sea_hot>*>(slot);
      bsx_hot_stream_body<deep_sea_hot, PAIR_DEEP_SEA_K, BSX_BLOCK, PAIR_HOT_NT>(g.obs, g.state, g.n_lanes, g.cells, g.cells_magic, g.dv, g.fn, w.block);
      break;
    }
    case BSX_FAM_CATCH: {
      const bsx_stream_seg<catch_hot>& g = *reinterpret_cast<const bsx_stream_seg<catch_hot>*>(slot);
      bsx_hot_stream_body<catch_hot, PAIR_CATCH_K, BSX_BLOCK, PAIR_HOT_NT>(g.obs, g.state, g.n_lanes, g.cells, g.cells_magic, g.dv, g.fn, w.block);
      break;
    }
    case BSX_FAM_MNIST:
      mnist_observe_body<PAIR_MNIST_K>(*reinterpret_cast<const mnist_observe_args*>(slot), w.block, s_lut);
      break;
    // wide rows of the chains, left packed by phase 0 (whole-sweep groups; row_stream.h)
    case BSX_FAM_MEMORY_CHAIN:
      bsx_row_stream_body<memory_rows, BSX_ROW_STREAM_K>(*reinterpret_cast<const bsx_row_seg*>(slot), w.block);
      break;
    case BSX_FAM_UMBRELLA_CHAIN:
      bsx_row_stream_body<umbrella_rows, BSX_ROW_STREAM_K>(*reinterpret_cast<const bsx_row_seg*>(slot), w.block);
      break;
    default: break;
  }
}

static inline bool bsx_is_mixed_pair_group(const bsx_group* g) {
  return g->family == BSX_FAM_PAIR_MIXED || g->family == BSX_FAM_SWEEP_MIXED;
}

#endif  // BSX_PAIR_MIXED_H_
