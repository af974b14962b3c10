// catch.hip — batched Catch for gfx950.  Replaces bsuite/environments/catch.py:68-114
// (`_reset`, `_step`, `_observation`) including its private auto-reset (:80-81).
//
// 221 B per lane per call at 10x5: 21 B of scalar columns + a 200 B board with at most two ones.
// The board is never materialised anywhere but in the output: a lane is its two hot cell indices
// (ball, paddle), decoded from the packed state.  Two launches per call:
//   advance  bsx_advance_kernel<catch_fam>: one lane per thread (paddle moves before the ball
//            drops, action ignored on the auto-reset call, `randint(columns)` on reset);
//   observe  bsx_hot_stream_kernel<catch_hot,2,256>: pure store stream over [B x rows*cols] f32.
//            rows*cols = 50 is not a multiple of 4, so a 16-byte chunk may straddle two lanes'
//            boards (one division per chunk, spill-over elements patched from lane l+1's state).
#include "bsx_host.h"
#include "catch_fam.h"
#include "pair_mixed.h"

static int catch_make(const bsx_catch_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state,
                      bsx_timestep_t out, double* info, catch_fam::args* a) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out, /*delta_ok=*/true);
  if (rc != 0) return rc;
  if (cfg->rows < 2 || cfg->rows > 64 || cfg->columns < 1 || cfg->columns > 64) return BSX_ERANGE;
  if (call->n_lanes > 0 && (state == nullptr || info == nullptr)) return BSX_ENULL;
  a->ctl = bsx_make_ctl(call);
  a->action = action; a->state = state; a->out = out; a->info = info;
  a->rows = cfg->rows; a->columns = cfg->columns;
  a->tile_cells_magic = 0; a->_pad = 0;
  return 0;
}

extern "C" int bsx_catch_step(const bsx_catch_t* cfg, const bsx_call_t* call, const int32_t* action,
                              int32_t* state, bsx_timestep_t out, double* info) {
  catch_fam::args a;
  int rc = catch_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  const uint32_t cells = (uint32_t)(cfg->rows * cfg->columns);
  return bsx_pair_call<catch_fam, catch_hot, 2>(a, call, action, state, out, cells, catch_hot{cfg->rows, cfg->columns});
}

extern "C" int bsx_group_set_catch(bsx_group_t* g, int32_t index, const bsx_catch_t* cfg, const bsx_call_t* call,
                                   const int32_t* action, int32_t* state, bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  int rc;
  catch_fam::args a;
  if (bsx_is_mixed_pair_group(g)) {            // one segment of the mixed two-kernel group
    rc = catch_make(cfg, call, action, state, out, info, &a);
    if (rc != 0) return rc;
    a.ctl.state_in = call->state_alt;            // pipelined sweeps: the advance reads the other column
    const uint32_t cells = (uint32_t)(cfg->rows * cfg->columns);
    if (cells < 4u) return BSX_ERANGE;
    bsx_stream_seg<catch_hot> sg;
    sg.obs = out.observation; sg.state = state; sg.n_lanes = a.ctl.n_lanes; sg.cells = cells;
    sg.cells_magic = bsx_div_magic(cells); sg.dv = bsx_make_div64(cells); sg.fn = catch_hot{cfg->rows, cfg->columns};
    // Whole-sweep group, small boards, one state column (no state_alt): phase 0 — latency-bound, its memory pipe
    // idle — writes the segment's boards itself as fused tiles; the segment leaves the phase-1 store stream, where
    // its 8 KiB runs went at 3.7 TB/s (tools/sweep_stream_parts.py: 27 MB in 7.3 us of a 145 us stream).
    // (any lane count: the tile of workgroup b starts b*256*cells floats into the 16-byte-aligned array, and
    // bsx_tile_stream writes the < 4 floats behind the last whole chunk one by one.  The predicate is the one
    // sweep_batch.prepare_groups uses to decide whether the segment gets a second state column: BSX_FUSED_CATCH_MAX_CELLS.)
    const bool fused = g->family == BSX_FAM_SWEEP_MIXED && call->state_alt == nullptr && cells <= BSX_FUSED_CATCH_MAX_CELLS;
    if (fused) a.tile_cells_magic = sg.cells_magic;
    return bsx_mixed_put(g, BSX_FAM_CATCH, index, call, &a, sizeof(a), &sg, sizeof(sg),
                              (uint64_t)(a.ctl.n_lanes + BSX_BLOCK - 1) / BSX_BLOCK,
                              fused ? 0 : bsx_flat_blocks((uint64_t)a.ctl.n_lanes * cells, PAIR_CATCH_K), 0);
  }
  rc = bsx_group_check_set(g, BSX_FAM_CATCH, index, call, sizeof(catch_fam::args),
                           sizeof(bsx_stream_seg<catch_hot>), 0);
  if (rc != 0) return rc;
  rc = catch_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  g->launch = bsx_group_launch_pair<catch_fam, catch_hot, 2>;
  g->n_phases = 2;
  return bsx_group_put_pair<catch_fam, catch_hot>(g, index, a, out.observation, state,
                                                  (uint32_t)(cfg->rows * cfg->columns),
                                                  catch_hot{cfg->rows, cfg->columns}, 2);
}
