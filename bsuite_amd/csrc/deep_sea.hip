// deep_sea.hip — batched DeepSea for gfx950.  Replaces bsuite/environments/deep_sea.py:103-144
// (`_get_observation`, `_reset`, `_step`) plus the auto-reset of bsuite/environments/base.py:54-65.
//
// Shape of the work at the headline config (N=30, B=2^20): 21 B of scalar traffic and 3600 B of
// observation stores per lane per call — a pure store stream.  Two launches per call:
//   advance  bsx_advance_kernel<deep_sea_fam>: the N*N-bit action mapping goes kernarg -> LDS once
//            per block (lane-divergent `mapping[row,col]` lookup); every thread advances one
//            lane (coalesced column loads/stores of action / packed state / reward / discount /
//            step_type), LAST/FIRST masks by wavefront ballot;
//   observe  bsx_hot_stream_kernel<deep_sea_hot,4,256>: a pure store stream over [B x N*N] f32 —
//            block b writes floats [b*4096,(b+1)*4096) as 4 lane-interleaved 16-byte stores per
//            thread, hot cells recomputed from the packed state column (4 B/lane, L2-resident).
// (With bsx_call_t.obs_paint — the delta observation mode — the call is ONE launch instead,
//  bsx_advance_delta_kernel: the advancing thread patches its lane's persistent board in place.)
// Decoupling the two is what makes the store stream fast: single-kernel variants that advance
// lanes, synchronise and then store (per-block 230 KB tiles, or flat 16 KiB runs with ping-pong
// state) measured 5.1-5.5 TB/s against 6.2-6.4 TB/s for this pair (profiles/r01/ab_*.log).
#include "bsx_host.h"
#include "deep_sea_fam.h"
#include "pair_mixed.h"

// Validates one call's arguments and fills the kernel argument struct (shared by step and group).
static int deep_sea_make(const bsx_deep_sea_t* cfg, const bsx_call_t* call, const int32_t* action,
                         int32_t* state, bsx_timestep_t out, double* info, deep_sea_fam::args* a) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out, /*delta_ok=*/true);
  if (rc != 0) return rc;
  if (cfg->size < 1 || cfg->size > BSX_DEEP_SEA_MAX_SIZE) return BSX_ERANGE;
  if (call->stream.mt_state != nullptr && !cfg->deterministic && call->stream.mt_gauss == nullptr)
    return BSX_ENULL;                      // the stochastic variant draws randn: needs the gauss cache columns
  if (call->n_lanes > 0 && (state == nullptr || info == nullptr)) return BSX_ENULL;
  a->ctl = bsx_make_ctl(call);
  a->action = action; a->state = state; a->out = out; a->info = info;
  a->move_cost = cfg->move_cost; a->inv_size = cfg->inv_size;
  a->size = cfg->size; a->deterministic = cfg->deterministic;
  for (int w = 0; w < DS_MAP_WORDS; ++w) a->mapping_bits[w] = cfg->mapping_bits[w];
  return 0;
}

extern "C" int bsx_deep_sea_step(const bsx_deep_sea_t* cfg, const bsx_call_t* call,
                                 const int32_t* action, int32_t* state, bsx_timestep_t out,
                                 double* info) {
  deep_sea_fam::args a;
  int rc = deep_sea_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  const uint32_t cells = (uint32_t)(cfg->size * cfg->size);
  return bsx_pair_call<deep_sea_fam, deep_sea_hot, 4>(a, call, action, state, out, cells, deep_sea_hot{cfg->size});
}

extern "C" int bsx_group_set_deep_sea(bsx_group_t* g, int32_t index, const bsx_deep_sea_t* cfg,
                                      const bsx_call_t* call, const int32_t* action, int32_t* state,
                                      bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  int rc;
  deep_sea_fam::args a;
  if (bsx_is_mixed_pair_group(g)) {            // one segment of the mixed two-kernel group
    rc = deep_sea_make(cfg, call, action, state, out, info, &a);
    if (rc != 0) return rc;
    a.ctl.state_in = call->state_alt;            // pipelined sweeps: the advance reads the other column
    const uint32_t cells = (uint32_t)(cfg->size * cfg->size);
    if (cells < 4u) return BSX_ERANGE;
    bsx_stream_seg<deep_sea_hot> sg;
    sg.obs = out.observation; sg.state = state; sg.n_lanes = a.ctl.n_lanes; sg.cells = cells;
    sg.cells_magic = bsx_div_magic(cells); sg.dv = bsx_make_div64(cells); sg.fn = deep_sea_hot{cfg->size};
    return bsx_mixed_put(g, BSX_FAM_DEEP_SEA, index, call, &a, sizeof(a), &sg, sizeof(sg),
                              (uint64_t)(a.ctl.n_lanes + BSX_BLOCK - 1) / BSX_BLOCK,
                              bsx_flat_blocks((uint64_t)a.ctl.n_lanes * cells, 4), 0);
  }
  rc = bsx_group_check_set(g, BSX_FAM_DEEP_SEA, index, call, sizeof(deep_sea_fam::args),
                           sizeof(bsx_stream_seg<deep_sea_hot>), 0);
  if (rc != 0) return rc;
  rc = deep_sea_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  g->launch = bsx_group_launch_pair<deep_sea_fam, deep_sea_hot, 4>;
  g->n_phases = 2;
  return bsx_group_put_pair<deep_sea_fam, deep_sea_hot>(g, index, a, out.observation, state,
                                                        (uint32_t)(cfg->size * cfg->size), deep_sea_hot{cfg->size}, 4);
}
