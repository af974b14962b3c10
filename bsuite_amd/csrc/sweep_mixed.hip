// sweep_mixed.hip — the mixed-family launches of a heterogeneous sweep (BASELINE config 5):
//   BSX_FAM_SMALL_MIXED   every small-observation family in one launch
//   BSX_FAM_SWEEP_MIXED   phase 0: every lane of EVERY family advanced by one launch (the lane advance of
//                         deep_sea / catch / mnist + the whole step of the small-observation families), and the
//                         software-pipelined form: that phase beside the previous step's observation store stream.
#include "small_obs.h"

// ------------------------------------------------------------------------------ mixed-family group
__global__ void __launch_bounds__(BSX_BLOCK) small_obs_mixed_group_kernel(const uint8_t* __restrict__ table,
                                                                          const int32_t* __restrict__ family,
                                                                          const bsx_group_index gi) {
  extern __shared__ __attribute__((aligned(16))) float s_obs[];
  __shared__ unsigned int s_cnt[2];
  const bsx_group_slot w = bsx_group_find(gi, (int)blockIdx.x);
  const int seg = w.seg;
  const uint32_t blk = w.block;
  const uint8_t* slot = table + (size_t)seg * SMALL_MIXED_STRIDE;
#define SMALL_MIXED_CASE(FAM, ENV) \
  case FAM: small_obs_group_body<ENV>(*reinterpret_cast<const ENV::args*>(slot), blk, s_obs, s_cnt); break;
  switch (w.tag >= 0 ? w.tag : family[seg]) {      // uniform per workgroup
    SMALL_MIXED_CASE(BSX_FAM_BANDIT, bandit_env)
    SMALL_MIXED_CASE(BSX_FAM_MEMORY_CHAIN, memory_chain_env)
    SMALL_MIXED_CASE(BSX_FAM_UMBRELLA_CHAIN, umbrella_chain_env)
    SMALL_MIXED_CASE(BSX_FAM_DISCOUNTING_CHAIN, discounting_chain_env)
    SMALL_MIXED_CASE(BSX_FAM_CARTPOLE, cartpole_env)
    SMALL_MIXED_CASE(BSX_FAM_MOUNTAIN_CAR, mountain_car_env)
    default: break;
  }
#undef SMALL_MIXED_CASE
}

int bsx_small_mixed_launch(bsx_group* g, int phase, hipStream_t st) {
  if (phase == 1) return 0;
  const dim3 grid((unsigned)g->total_blocks), block(BSX_BLOCK);
  small_obs_mixed_group_kernel<<<grid, block, g->lds_bytes, st>>>((const uint8_t*)g->d_args, (const int32_t*)g->d_args2, g->index1());
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------ whole-sweep group, phase 0
// BSX_FAM_SWEEP_MIXED: ONE launch advances every lane of a heterogeneous sweep — the lane-advance of
// deep_sea / catch / mnist segments (whose observation stream follows as phase 1, pair_mixed.hip) and the
// complete step of every small-observation segment — except that a memory_chain / umbrella_chain segment with a wide
// row and a row scratch (ABI v12) only leaves its rows PACKED here; phase 1 decodes them (row_stream.h).  All of this is latency-bound work that moves a
// few percent of the sweep's bytes; as separate launches (advance, two small-family groups, counter
// bump) it cost ~45 us of a ~185 us sweep step whether serialised or spread over HIP
// streams (cross-queue waits cost ~10 us each; kernels sharing the machine with the store stream
// stretch it: profiles/r02/ab_sweep_*.log, sweep_*_timeline*.txt).  The last workgroup to retire bumps
// the call counter the segments share, so no other kernel has to.
// (Tried, profiles/r04/ab_sweep_state_warmers.log: this launch takes ~20 us back to back and ~30 us behind the 850 MB
// store stream, so the stream launch was ended by one "warmer" workgroup per small-observation workgroup of this one,
// loading — and dropping — the state words it is going to read into the same XCD's L2.  166.6 -> 192.6 us per sweep
// step: reads placed among the stream's draining stores wait just as long and slow the stores down too.  Not adopted.)
__device__ __forceinline__ void sweep_phase0_body(const uint8_t* __restrict__ table, const int32_t* __restrict__ tags,
                                                  const bsx_group_index& gi, uint64_t* counter, uint32_t* ticket,
                                                  const uint32_t block, const uint32_t n_blocks, float* s_obs,
                                                  unsigned int* s_cnt, deep_sea_fam::shared& s_ds, catch_fam::shared& s_ca,
                                                  int32_t* s_tile_state, const uint32_t ticket_index = 0xFFFFFFFFu) {
  BSX_LIFE(0);
  const bsx_group_slot w = bsx_group_find(gi, (int)block);
  const int tag = w.tag >= 0 ? w.tag : tags[w.seg];  // uniform per workgroup
  const uint32_t blk = w.block;
  const uint8_t* slot = table + (size_t)w.seg * BSX_MIXED_ADV_STRIDE;
  BSX_LIFE_AFTER_S(1, tag);                          // the map entry has arrived
#define SWEEP_SMALL_CASE(FAM, ENV) \
  case FAM: small_obs_group_body<ENV, 0, true>(*reinterpret_cast<const ENV::args*>(slot), blk, s_obs, s_cnt); break;
  switch (tag) {
    case BSX_FAM_DEEP_SEA: {
      const deep_sea_fam::args& a = *reinterpret_cast<const deep_sea_fam::args*>(slot);
      if (bsx_ctl_lean(a.ctl)) bsx_advance_body<deep_sea_fam, true>(a, blk, s_ds, s_cnt);
      else bsx_advance_body<deep_sea_fam, false, 0>(a, blk, s_ds, s_cnt);
      break;
    }
    case BSX_FAM_CATCH: {
      const catch_fam::args& a = *reinterpret_cast<const catch_fam::args*>(slot);
      int32_t* tile = a.tile_cells_magic != 0u ? s_tile_state : nullptr;      // uniform: boards written right here
      if (bsx_ctl_lean(a.ctl)) bsx_advance_body<catch_fam, true>(a, blk, s_ca, s_cnt, tile);
      else bsx_advance_body<catch_fam, false, 0>(a, blk, s_ca, s_cnt, tile);
      if (tile != nullptr) {                   // (the advance ended with a barrier: the tile's states are complete)
        const int64_t lane0 = (int64_t)blk * BSX_BLOCK, left = a.ctl.n_lanes - lane0;
        const uint32_t cells = (uint32_t)(a.rows * a.columns);
        // (write-through chunks like every eager output: the closed schedule 168-171 -> 166.5-168.4 us per sweep step, split and
        // pipelined within noise; profiles/r06/ab_catch_tile256_write_through.log)
        bsx_tile_stream<catch_hot, BSX_ST_WT>(a.out.observation + lane0 * (int64_t)cells, tile, left < BSX_BLOCK ? (int)left : BSX_BLOCK, cells,
                        a.tile_cells_magic, catch_hot{a.rows, a.columns});
      }
      break;
    }
    case BSX_FAM_MNIST: mnist_advance_body<0>(*reinterpret_cast<const mnist_args*>(slot), blk, s_cnt); break;
    SWEEP_SMALL_CASE(BSX_FAM_BANDIT, bandit_env)
    SWEEP_SMALL_CASE(BSX_FAM_MEMORY_CHAIN, memory_chain_env)
    SWEEP_SMALL_CASE(BSX_FAM_UMBRELLA_CHAIN, umbrella_chain_env)
    SWEEP_SMALL_CASE(BSX_FAM_DISCOUNTING_CHAIN, discounting_chain_env)
    SWEEP_SMALL_CASE(BSX_FAM_CARTPOLE, cartpole_env)
    SWEEP_SMALL_CASE(BSX_FAM_MOUNTAIN_CAR, mountain_car_env)
    default: break;
  }
#undef SWEEP_SMALL_CASE
  // Every workgroup read the call counter when it started; the one that retires last moves it on.
  // Two-level ticket (64 shards, one 128-byte line each, then one word): several thousand arrivals on ONE
  // word would serialise at ~12 ns each (the lesson of the episode counters, bsx_device.h).
  bsx_final_barrier();
  // (No fence: a workgroup's reads of the counter completed before its barrier, and a release fence here
  // would write back this XCD's whole L2 once per workgroup — measured 165 us instead of 25.)
  if (threadIdx.x == 0 && counter != nullptr) {
    // (`n_blocks` workgroups take part, numbered 0 .. n_blocks-1 by ticket_index — `block` itself unless the launch runs a
    // sub-range of the phase)
    const uint32_t shard = (ticket_index != 0xFFFFFFFFu ? ticket_index : block) & 63u;
    const uint32_t in_shard = (n_blocks - shard + 63u) >> 6;            // workgroups with this shard id
    uint32_t* word = ticket + 32u * (shard + 1u);
    if (atomicAdd(word, 1u) == in_shard - 1u) {
      *word = 0u;
      const uint32_t live_shards = n_blocks < 64u ? n_blocks : 64u;
      if (atomicAdd(ticket, 1u) == live_shards - 1u) {
        *ticket = 0u;
        *counter += 1ull;
      }
    }
  }
  BSX_LIFE(7);
}

// block_base: the launch runs the workgroups [block_base, block_base + gridDim.x) of the group's phase-0 grid (the split
// schedule launches the tail of the grid on its own; 0 = the whole phase).
__global__ void __launch_bounds__(BSX_BLOCK) sweep_phase0_kernel(
    const uint8_t* __restrict__ table, const int32_t* __restrict__ tags, const bsx_group_index gi, uint64_t* counter,
    uint32_t* ticket, uint64_t* trace, const uint32_t block_base) {
  extern __shared__ __attribute__((aligned(16))) float s_obs[];
  __shared__ unsigned int s_cnt[2];
  __shared__ deep_sea_fam::shared s_ds;
  __shared__ catch_fam::shared s_ca;
  __shared__ int32_t s_tile_state[BSX_BLOCK];
  const uint32_t block = block_base + blockIdx.x;
  if (trace != nullptr && threadIdx.x == 0) trace[3 * block] = wall_clock64();        // bsx_group_trace
  // (the retirement ticket counts THIS launch's workgroups: shards by blockIdx.x, not by the phase-wide index)
  sweep_phase0_body(table, tags, gi, counter, ticket, block, gridDim.x, s_obs, s_cnt, s_ds, s_ca, s_tile_state, blockIdx.x);
  if (trace != nullptr && threadIdx.x == 0) {
    trace[3 * block + 1] = wall_clock64();
    trace[3 * block + 2] = (uint64_t)tags[bsx_group_find(gi, (int)block).seg];
  }
}

int bsx_sweep_launch_phase0(bsx_group* g, hipStream_t st) {
#ifdef BSX_TRACE_LIFE
  {
    uint64_t* life = g->trace != nullptr ? g->trace + 3 * (size_t)g->total_blocks : nullptr;   // bsx_group_trace: 11 words per workgroup
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(bsx_life_trace_ptr), &life, sizeof(life), 0, hipMemcpyHostToDevice, st);
  }
#endif
  sweep_phase0_kernel<<<dim3((unsigned)g->total_blocks), dim3(BSX_BLOCK), g->lds_bytes, st>>>(
      (const uint8_t*)g->d_args, g->d_tags, g->index1(), g->shared_counter, g->d_ticket, g->trace, 0u);
  return (int)hipGetLastError();
}

// Software-pipelined sweep step: ONE launch = the observation store stream of sweep step s (group
// `streams_of`) beside phase 0 — every lane's advance — of step s+1 (group `advances_of`).  The two groups
// hold the same segments with the two-kernel families' state columns swapped (bsx_call_t.state_alt) and
// their own TimeStep buffers, so nothing in the launch depends on anything else in it: the stream reads the
// column phase 0 of step s wrote in the previous launch, phase 0 of step s+1 reads it too and writes the
// other one.  The latency-bound phase 0 (~26 us alone) hides beside the ~140 us store stream.
__global__ void __launch_bounds__(BSX_BLOCK) sweep_pipelined_kernel(
    const uint8_t* __restrict__ adv_table, const int32_t* __restrict__ adv_tags, const bsx_group_index adv_gi,
    uint64_t* counter, uint32_t* ticket, const uint32_t adv_blocks, const uint32_t place,
    const uint8_t* __restrict__ str_table, const int32_t* __restrict__ str_tags, const bsx_group_index str_gi) {
  extern __shared__ __attribute__((aligned(16))) float s_obs[];
  __shared__ unsigned int s_cnt[2];
  __shared__ deep_sea_fam::shared s_ds;
  __shared__ catch_fam::shared s_ca;
  __shared__ float s_lut[MNIST_LUT_FLOATS];
  __shared__ int32_t s_tile_state[BSX_BLOCK];
  const bsx_pipe_role r = bsx_pipe_role_of(blockIdx.x, gridDim.x, adv_blocks, place);   // uniform per workgroup
  if (r.adv) sweep_phase0_body(adv_table, adv_tags, adv_gi, counter, ticket, r.index, adv_blocks, s_obs, s_cnt, s_ds, s_ca, s_tile_state);
  else pair_mixed_stream_body(str_table, str_tags, str_gi, r.index, s_lut);
}

int bsx_sweep_launch_pipelined(bsx_group* streams_of, bsx_group* advances_of, hipStream_t st) {
  const uint64_t blocks = (uint64_t)advances_of->total_blocks + (uint64_t)streams_of->total_blocks2;
  if (blocks == 0 || blocks > 0x7FFFFFFFull) return BSX_EINVAL;
  static const int place = bsx_env_int("BSX_PIPELINED_PLACE", 0);       // bsx_pipe_role_of: first (measured best)
  sweep_pipelined_kernel<<<dim3((unsigned)blocks), dim3(BSX_BLOCK), advances_of->lds_bytes, st>>>(
      (const uint8_t*)advances_of->d_args, advances_of->d_tags, advances_of->index1(), advances_of->shared_counter,
      advances_of->d_ticket, (uint32_t)advances_of->total_blocks, (uint32_t)place, (const uint8_t*)streams_of->d_args2, streams_of->d_tags,
      streams_of->index2());
  return (int)hipGetLastError();
}

// Split closed-loop sweep step (bsx_group_step_split): TWO launches like bsx_group_step, cut differently.  Phase 0 holds
// two kinds of workgroups: those whose segment has a share of the phase-1 store stream (the lane advance of deep_sea /
// mnist / large catch boards, the packed rows of the chains) — the stream depends on them — and those that are a
// small-observation segment's whole step, on which nothing in the step depends.  With the segments ordered so that the
// second kind comes first in the phase-0 grid (g->split_block = the first workgroup of the first kind):
//   launch 1   phase-0 workgroups [split, total), split <= split_block: what the stream waits for (920 workgroups of the
//              4213 at 2^20 lanes) + what else fits the same dispatch round (below);
//   launch 2   the store stream beside phase-0 workgroups [0, split) (sweep_pipelined_kernel with both halves
//              taken from THIS group): the latency-bound small families hide beside the 850 MB of stores instead of
//              standing in front of them; their last workgroup to retire bumps the call counter.
// Everything in a step reads the actions of that step only, so the schedule is closed-loop: the TimeSteps of step s are
// complete when launch 2 ends.
// Workgroups of launch 1 when phase 0 does not fit one dispatch round: the machine's resident workgroup slots for THIS kernel
// with THIS group's dynamic LDS (CUs x hipOccupancyMaxActiveBlocksPerMultiprocessor: 256 x 8 = 2048 on MI355X for the whole
// sweep) plus the 18 % the dispatcher places while the first workgroups retire — the margin measured at 2^20 lanes
// (profiles/r05/ab_sweep_split_point*.log: 164.2 us per sweep step with no top-up, 163.0 with 1200 workgroups, 162.0 with 1500,
// 163.2 with 1800, noise from 2200 = slots x 1.07 up to 3000 = slots x 1.46; with ALL of phase 0 in launch 1 168).  Derived
// per group and device, not a constant: another mix of families (more LDS per workgroup = fewer slots) or a part with fewer
// CUs gets its own round.  -DBSX_SPLIT_ROUND_DEFAULT=<n> / BSX_SPLIT_ROUND (tuning build) pin it; 0 = the lane advance only.
static int64_t sweep_split_round(bsx_group* g) {
#ifdef BSX_SPLIT_ROUND_DEFAULT
  static const int pinned = bsx_env_int("BSX_SPLIT_ROUND", BSX_SPLIT_ROUND_DEFAULT);
#else
  static const int pinned = bsx_env_int("BSX_SPLIT_ROUND", -1);
#endif
  if (pinned >= 0) return pinned;
  if (g->split_round >= 0) return g->split_round;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sweep_phase0_kernel, BSX_BLOCK, g->lds_bytes) != hipSuccess ||
      cus <= 0 || per_cu <= 0) {
    (void)hipGetLastError();
    return 0;                               // unknown machine: the lane advance only (always correct, a little slower)
  }
  const int64_t slots = (int64_t)cus * per_cu;
  g->split_round = slots + slots * 18 / 100;
  return g->split_round;
}

int bsx_sweep_launch_split(bsx_group* g, hipStream_t st) {
  if (g->split_block < 0 || g->split_block > g->total_blocks) return BSX_EMODE;
  // Launch 1 is one dispatch round whatever it holds, and the lane advance fills less than half of it: when phase 0 is MORE
  // than one round, launch 1 is topped up with small-observation workgroups — the last, lightest ones of their part of the
  // grid — which then no longer compete with the store stream in launch 2 (sweep_split_round above).  On a second box, three
  // repetitions (ab_sweep_split_order.log): 168.5 -> 166.9 (launch 1 = 2420 workgroups), 166.7 (3000), 168.3 (3600); with
  // the small segments in narrow-rows-first order, i.e. the WIDE rows topping launch 1 up, 167.1 -> 167.4 / 170.6 / 169.1.
  const int64_t round = sweep_split_round(g);
  int64_t extra = 0;
  if (g->total_blocks > round) extra = round - (g->total_blocks - g->split_block);
  if (extra < 0) extra = 0;
  const int64_t split = g->split_block > extra ? g->split_block - extra : 0;
  const int64_t n_tail = g->total_blocks - split;
  if (n_tail > 0) {
    sweep_phase0_kernel<<<dim3((unsigned)n_tail), dim3(BSX_BLOCK), g->lds_bytes, st>>>(
        (const uint8_t*)g->d_args, g->d_tags, g->index1(), split == 0 ? g->shared_counter : nullptr, g->d_ticket, nullptr, (uint32_t)split);
    const int rc1 = (int)hipGetLastError();       // a failed launch 1 is reported as such, and launch 2 (which would bump the
    if (rc1 != 0) return rc1;                     // call counter over lanes that never advanced) is not issued
  }
  const uint64_t blocks = (uint64_t)split + (uint64_t)g->total_blocks2;
  if (blocks == 0) return (int)hipGetLastError();
  if (blocks > 0x7FFFFFFFull) return BSX_EINVAL;
  static const int place = bsx_env_int("BSX_SPLIT_PLACE", 0);            // bsx_pipe_role_of: 0 = the phase-0 workgroups first
  sweep_pipelined_kernel<<<dim3((unsigned)blocks), dim3(BSX_BLOCK), g->lds_bytes, st>>>(
      (const uint8_t*)g->d_args, g->d_tags, g->index1(), g->shared_counter, g->d_ticket, (uint32_t)split, (uint32_t)place,
      (const uint8_t*)g->d_args2, g->d_tags, g->index2());
  return (int)hipGetLastError();
}
