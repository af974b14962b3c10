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

// ---------------------------------------------------------------------------------------------
// ONE launch per step (ABI v12: bsx_call_t.flags & BSX_CALL_STATE_TAGGED; lean calls, columns <= 8, batches beyond the
// fused-tile range) — VERDICT r04 next #5, the reader / writer scheme of deep_sea_step1_kernel for catch.  The observation
// store stream, unchanged in shape (workgroup b writes floats [b*K*1024, (b+1)*K*1024) as K 16-byte stores per thread, no
// barrier, no LDS), with nobody advancing the lanes beforehand: every thread loads the state word and the action of the
// (one or two) lanes each of its chunks belongs to and recomputes their transition — paddle move, ball drop, ~20 vector
// instructions; the thread whose chunk holds the FIRST float of a lane's row is that lane's WRITER and alone stores the
// new state word, reward / discount / step_type and the miss count.  A reader may find the word already advanced: bit 7
// (the parity of the next call's index, written by every advance) says which.
// The one thing a reader cannot afford is the reset's Philox block (r01 measured such a stream at 100 us against 42): a
// wave covers ~5 lanes, so on 4 calls in 10 some lane of it resets.  Episodes take exactly `rows` calls, so the call
// index of a lane's next reset is known: on the calls whose index is a multiple of `rows` every WRITER draws the ball
// column of its lane's next episode — word 0 of the stream of THAT call, exactly what the reset would draw — and parks it
// in spare bits of the state word; the threads that meet the reset then just read it.  One call in `rows` pays the
// Philox block, in one thread per lane; a word without a parked value (a state that came from load_state_dict, an
// explicit reset(), any other advance path) makes the few threads that need it draw for themselves, once.
template <int K>
__global__ void __launch_bounds__(BSX_BLOCK) catch_step1_kernel(const catch_fam::args a, const uint32_t cells,
                                                                const uint32_t cells_magic, const bsx_div64 dv) {
  const catch_fam::shared none{};
  const int64_t n_lanes = a.ctl.n_lanes;
  const uint64_t step = bsx_step_of(a.ctl);
  const int32_t tag_new = (int32_t)(((uint32_t)(step + 1) & 1u) << CATCH_TAG_SHIFT);
  const bool park = step % (uint64_t)a.rows == 0;                        // uniform: every `rows`-th call
  const catch_hot fn{a.rows, a.columns};
  const uint64_t total = (uint64_t)n_lanes * cells;                      // a multiple of 4 (the launcher's condition)
  const uint64_t F0 = (uint64_t)blockIdx.x * (uint64_t)(K * 4 * BSX_BLOCK);
  const uint64_t lane_b = __umul64hi(F0, dv.m) >> dv.s;                  // uniform: the lane of the workgroup's first float
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);
  bsx_f4* __restrict__ o4 = reinterpret_cast<bsx_f4*>(a.out.observation + F0);
  const int32_t* __restrict__ st_b = a.state + lane_b;
  const int32_t* __restrict__ act_b = a.action + lane_b + (int64_t)(step & (uint64_t)a.ctl.action_ring_mask) * n_lanes;

  uint32_t dl[K];
  int r0[K];
  bool live[K], two[K];
  int32_t wA[K], wB[K];
  int aA[K], aB[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t c = (threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63);
    const uint32_t f = r_b + (c << 2);
    dl[u] = __umulhi(f, cells_magic);
    r0[u] = (int)(f - dl[u] * cells);
    live[u] = F0 + ((uint64_t)c << 2) + 3 < total;
    two[u] = live[u] && (int)cells - r0[u] < 4;                          // the chunk runs over into the next lane's row
    wA[u] = live[u] ? st_b[dl[u]] : 0;
    aA[u] = live[u] ? act_b[dl[u]] : 0;
    wB[u] = two[u] ? st_b[dl[u] + 1] : 0;
    aB[u] = two[u] ? act_b[dl[u] + 1] : 0;
  }
  // The new packed state of lane lane_b + d given its state word w and action; `writer` publishes it.
  auto resolve = [&](uint32_t d, int32_t w, int act, bool writer, int& type) -> int32_t {
    type = -1;
    if (!writer && (w & CATCH_TAG_BIT) == tag_new) return w;            // the lane's writer has been here already
    const int64_t L = (int64_t)lane_b + d;
    const uint64_t lane = a.ctl.lane_offset + (uint64_t)L;
    int32_t nst; double reward;
    const int ty = catch_fam::advance<true, false, true>(a, none, L, lane, step, w, act, nst, reward, writer, park);
    if (writer) {
      type = ty;
      a.state[L] = nst;
      bsx_emit_at<0, 0, false>(a.ctl, a.out, L, L, lane, step, ty, reward);
    }
    return nst;
  };
  unsigned int n_last = 0, n_first = 0;
#pragma unroll
  for (int u = 0; u < K; ++u) {
    int tA = -1, tB = -1;
    int32_t nA = 0, nB = 0;
    if (live[u]) nA = resolve(dl[u], wA[u], aA[u], r0[u] == 0, tA);       // the chunk holds the first float of lane dl's row
    if (two[u]) nB = resolve(dl[u] + 1, wB[u], aB[u], true, tB);          // ... and the first float of lane dl+1's
    n_last += (unsigned int)__popcll(__ballot(tA == BSX_LAST)) + (unsigned int)__popcll(__ballot(tB == BSX_LAST));
    n_first += (unsigned int)__popcll(__ballot(tA == BSX_FIRST)) + (unsigned int)__popcll(__ballot(tB == BSX_FIRST));
    if (!live[u]) continue;
    int ha, hb;
    fn(nA, ha, hb);
    const int a0 = ha - r0[u], b0 = hb - r0[u];
    bsx_f4 v;
    v.x = (a0 == 0 || b0 == 0) ? 1.0f : 0.0f;
    v.y = (a0 == 1 || b0 == 1) ? 1.0f : 0.0f;
    v.z = (a0 == 2 || b0 == 2) ? 1.0f : 0.0f;
    v.w = (a0 == 3 || b0 == 3) ? 1.0f : 0.0f;
    if (two[u]) {                             // elements j >= over belong to the next lane's row
      const int over = (int)cells - r0[u];
      int na, nb;
      fn(nB, na, nb);
      const int a1 = na + over, b1 = nb + over;
      if (over <= 1) v.y = (a1 == 1 || b1 == 1) ? 1.0f : 0.0f;
      if (over <= 2) v.z = (a1 == 2 || b1 == 2) ? 1.0f : 0.0f;
      v.w = (a1 == 3 || b1 == 3) ? 1.0f : 0.0f;
    }
    o4[(threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63)] = v;
  }
  // one pair of sharded atomics per WAVE that saw an episode end or begin
  if (a.ctl.counters != nullptr && (threadIdx.x & 63u) == 0 && (n_last | n_first) != 0u) {
    unsigned long long* shard = (unsigned long long*)a.ctl.counters +
                                (size_t)((blockIdx.x * (BSX_BLOCK / BSX_WAVE) + (threadIdx.x >> 6)) & (BSX_COUNTER_SHARDS - 1)) * BSX_COUNTER_STRIDE;
    if (n_last) atomicAdd(&shard[0], (unsigned long long)n_last);
    if (n_first) atomicAdd(&shard[1], (unsigned long long)n_first);
  }
}

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
  // ONE launch (catch_step1_kernel) where the caller vouches for the state words' parity tags, the call is lean and the
  // batch lies beyond the range of the fused one-launch tile step (bsx_pair_call: up to 128 MiB of boards per step); an
  // explicit reset() (every lane draws NOW), an odd float count (ragged tail) and boards with more than 8 columns (the
  // parked draw has 3 bits) keep the lane advance + store stream.
  static const int step1_env = bsx_env_int("BSX_CATCH_STEP1", 1);
  static const int step1_min_mib = bsx_env_int("BSX_CATCH_STEP1_MIN_MIB", 128);
  const uint64_t total = (uint64_t)call->n_lanes * cells;
  if (step1_env != 0 && (call->flags & BSX_CALL_STATE_TAGGED) && call->n_steps <= 1 && !call->force_reset && bsx_ctl_lean(a.ctl) &&
      call->obs_paint == nullptr && cfg->columns <= 8 && cells >= 4u && (total & 3ull) == 0 &&
      (int64_t)total * 4 > ((int64_t)step1_min_mib << 20)) {
    constexpr int K = 2;
    const uint64_t blocks = (total + (uint64_t)K * 4 * BSX_BLOCK - 1) / ((uint64_t)K * 4 * BSX_BLOCK);
    if (blocks > 0x7FFFFFFFull) return BSX_EINVAL;
    catch_step1_kernel<K><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, (hipStream_t)call->hip_stream>>>(
        a, cells, bsx_div_magic(cells), bsx_make_div64(cells));
    return bsx_launch_status();
  }
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
                              fused ? 0 : bsx_flat_blocks((uint64_t)a.ctl.n_lanes * cells, 2), 0);
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
