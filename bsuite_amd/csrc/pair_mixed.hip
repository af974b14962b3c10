// pair_mixed.hip — ONE launch pair for all the two-kernel families of a heterogeneous sweep.
//
// deep_sea, catch and the mnist bandit each step as advance kernel + observation stream kernel.  In a
// sweep (BASELINE config 5: 468 bsuite_ids as lane segments) their three grouped pairs are six kernels;
// the timeline of a captured sweep step (profiles/r02/sweep_graph_timeline_before.txt) shows what that
// costs: every dependent node of a HIP graph starts 6-14 us after its predecessor ends, the three
// advance kernels run one after another in front of "their" streams, and two store-bound kernels that
// overlap slow each other down.  A BSX_FAM_PAIR_MIXED group holds segments of all three families in
// fixed-stride argument slots next to a family tag (like BSX_FAM_SMALL_MIXED for the small families):
//   phase 0  pair_mixed_advance_kernel   every lane of every segment advances in one ~8 us launch;
//   phase 1  pair_mixed_stream_kernel    ONE store stream over all observation arrays (~850 of the
//                                        sweep's 886 MB), each workgroup running its family's stream
//                                        body (deep_sea 4 x 4 KiB, catch 2 x 4 KiB, mnist 4 x 4 KiB runs).
#include "catch_fam.h"
#include "deep_sea_fam.h"
#include "mnist_fam.h"
#include "pair_mixed.h"

#define PAIR_ADV_STRIDE BSX_MIXED_ADV_STRIDE
#define PAIR_STR_STRIDE BSX_MIXED_STR_STRIDE

static_assert(sizeof(deep_sea_fam::args) <= PAIR_ADV_STRIDE && sizeof(catch_fam::args) <= PAIR_ADV_STRIDE &&
              sizeof(mnist_args) <= PAIR_ADV_STRIDE, "advance argument struct exceeds the mixed-group slot");
static_assert(sizeof(bsx_stream_seg<deep_sea_hot>) <= PAIR_STR_STRIDE && sizeof(bsx_stream_seg<catch_hot>) <= PAIR_STR_STRIDE &&
              sizeof(mnist_observe_args) <= PAIR_STR_STRIDE && sizeof(bsx_row_seg) <= PAIR_STR_STRIDE,
              "stream argument struct exceeds the mixed-group slot");

__global__ void __launch_bounds__(BSX_BLOCK) pair_mixed_advance_kernel(const uint8_t* __restrict__ table,
                                                                       const int32_t* __restrict__ family,
                                                                       const bsx_group_index gi) {
  __shared__ deep_sea_fam::shared s_ds;
  __shared__ catch_fam::shared s_ca;
  __shared__ unsigned int s_cnt[2];
  const bsx_group_slot w = bsx_group_find(gi, (int)blockIdx.x);
  const uint8_t* slot = table + (size_t)w.seg * PAIR_ADV_STRIDE;
  switch (w.tag >= 0 ? w.tag : (family[w.seg] & 0xFF)) {   // uniform per workgroup
    case BSX_FAM_DEEP_SEA: bsx_advance_body<deep_sea_fam>(*reinterpret_cast<const deep_sea_fam::args*>(slot), w.block, s_ds, s_cnt); break;
    case BSX_FAM_CATCH: bsx_advance_body<catch_fam>(*reinterpret_cast<const catch_fam::args*>(slot), w.block, s_ca, s_cnt); break;
    case BSX_FAM_MNIST: mnist_advance_body(*reinterpret_cast<const mnist_args*>(slot), w.block, s_cnt); break;
    default: break;
  }
}

__global__ void __launch_bounds__(BSX_BLOCK) pair_mixed_stream_kernel(const uint8_t* __restrict__ table,
                                                                      const int32_t* __restrict__ family,
                                                                      const bsx_group_index gi) {
  __shared__ float s_lut[MNIST_LUT_FLOATS];
  pair_mixed_stream_body(table, family, gi, blockIdx.x, s_lut);
}

int bsx_mixed_launch_stream(bsx_group* g, hipStream_t st) {
  if (g->total_blocks2 > 0)
    pair_mixed_stream_kernel<<<dim3((unsigned)g->total_blocks2), dim3(BSX_BLOCK), 0, st>>>(
        (const uint8_t*)g->d_args2, g->d_tags, g->index2());
  return (int)hipGetLastError();
}

static int pair_mixed_launch(bsx_group* g, int phase, hipStream_t st) {
  if (phase != 1)
    pair_mixed_advance_kernel<<<dim3((unsigned)g->total_blocks), dim3(BSX_BLOCK), 0, st>>>(
        (const uint8_t*)g->d_args, g->d_tags, g->index1());
  if (phase != 0) return bsx_mixed_launch_stream(g, st);
  return (int)hipGetLastError();
}

// BSX_FAM_SWEEP_MIXED: phase 0 advances every lane of every family (and bumps the shared call counter
// itself, see small_obs.hip), phase 1 is the same store stream.
static int sweep_mixed_launch(bsx_group* g, int phase, hipStream_t st) {
  int rc = 0;
  if (phase != 1) rc = bsx_sweep_launch_phase0(g, st);
  if (rc == 0 && phase != 0) rc = bsx_mixed_launch_stream(g, st);
  return rc;
}

int bsx_mixed_put(bsx_group* g, int32_t family, int32_t index, const bsx_call_t* call,
                  const void* adv, size_t adv_size, const void* str, size_t str_size,
                  uint64_t blocks1, uint64_t blocks2, size_t lds) {
  if (g == nullptr || !bsx_is_mixed_pair_group(g)) return BSX_EINVAL;
  if (adv_size > PAIR_ADV_STRIDE || str_size > PAIR_STR_STRIDE) return BSX_EINVAL;
  if (g->family == BSX_FAM_PAIR_MIXED && str == nullptr) return BSX_EINVAL;   // small families: sweep groups only
  int rc = bsx_group_check_set(g, g->family, index, call, PAIR_ADV_STRIDE, PAIR_STR_STRIDE, 0);
  if (rc != 0) return rc;
  if (blocks1 > 0x3FFFFFFFull || blocks2 > 0x3FFFFFFFull) return BSX_EINVAL;
  // the whole-sweep launches are compiled without the MT19937-exact generators (a seeded small-batch mode: such
  // segments go into per-family groups or step eagerly)
  if (g->family == BSX_FAM_SWEEP_MIXED && call->stream.mt_state != nullptr) return BSX_EMODE;
  if (g->family == BSX_FAM_SWEEP_MIXED) {           // the group bumps ONE call counter: all segments must share it
    if (g->shared_counter == nullptr) g->shared_counter = const_cast<uint64_t*>(call->stream.step_base);
    else if (g->shared_counter != call->stream.step_base) return BSX_EINVAL;
  }
  memcpy(&g->args[(size_t)index * PAIR_ADV_STRIDE], adv, adv_size);
  if (str != nullptr) memcpy(&g->args2[(size_t)index * PAIR_STR_STRIDE], str, str_size);
  if (g->tags.empty()) g->tags.assign((size_t)g->n, -1);
  g->tags[index] = family;
  g->blocks[index] = (int32_t)blocks1; g->blocks2[index] = (int32_t)blocks2;
  // (per segment, re-evaluated at every put and summed up at commit: ADVICE r04 — a segment set again WITH its second
  // column no longer blocks the pipelined step.)  The chains' wide rows need no second column: each group of a
  // pipelined pair brings its own row scratch (checked in bsx_group_step_pipelined).
  const bool row_seg = family == BSX_FAM_MEMORY_CHAIN || family == BSX_FAM_UMBRELLA_CHAIN;
  if (g->needs_alt.empty()) { g->needs_alt.assign((size_t)g->n, 0); g->row_scratch.assign((size_t)g->n, nullptr); }
  g->needs_alt[index] = (str != nullptr && blocks2 > 0 && !row_seg && call->state_alt == nullptr) ? 1 : 0;
  g->row_scratch[index] = (row_seg && str != nullptr) ? call->row_scratch : nullptr;
  if (lds > g->lds_bytes) g->lds_bytes = lds;
  g->is_set[index] = 1;
  g->launch = g->family == BSX_FAM_SWEEP_MIXED ? sweep_mixed_launch : pair_mixed_launch;
  g->n_phases = 2;
  return 0;
}
