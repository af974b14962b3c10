// bsx_host.h — host-side glue shared by the C-ABI entry points (argument checks, launch math).
#ifndef BSX_HOST_H_
#define BSX_HOST_H_

#include <stdlib.h>
#include <string.h>

#include <vector>

#include "bsx_device.h"

static inline int bsx_check_call(const bsx_call_t* call, const void* action, const bsx_timestep_t& out,
                                 bool delta_ok = false) {
  if (call == nullptr) return BSX_ENULL;
  if (call->obs_paint != nullptr && (!delta_ok || call->n_steps > 1)) return BSX_EMODE;
  if (call->n_lanes < 0 || call->n_lanes > ((int64_t)1 << 40)) return BSX_EINVAL;
  if (call->n_lanes == 0) return 0;
  if (out.reward == nullptr || out.discount == nullptr || out.step_type == nullptr ||
      out.observation == nullptr)
    return BSX_ENULL;
  if (action == nullptr && !call->force_reset) return BSX_ENULL;
  if ((reinterpret_cast<uintptr_t>(out.observation) & 15u) != 0) return BSX_EALIGN;
  if (call->wrap.kind < BSX_WRAP_NONE || call->wrap.kind > BSX_WRAP_NOISE_SCALE) return BSX_EINVAL;
  if (call->n_steps < 0 || (call->n_steps > 1 && call->force_reset)) return BSX_EINVAL;
  // action ring: a power of two of rows, for single-step calls only (a rollout already takes [T,B] actions)
  if (call->action_ring < 0 || (call->action_ring & (call->action_ring - 1)) != 0) return BSX_EINVAL;
  if (call->action_ring > 1 && (call->n_steps > 1 || call->obs_paint != nullptr)) return BSX_EMODE;
  if ((call->stream.mt_state == nullptr) != (call->stream.mt_pos == nullptr)) return BSX_ENULL;
  if ((call->stream.mt_gauss == nullptr) != (call->stream.mt_has_gauss == nullptr)) return BSX_ENULL;
  if (call->stream.mt_state != nullptr && call->wrap.kind >= BSX_WRAP_NOISE &&
      (call->wrap.mt_state == nullptr || call->wrap.mt_pos == nullptr || call->wrap.mt_gauss == nullptr ||
       call->wrap.mt_has_gauss == nullptr))
    return BSX_ENULL;                      // MT19937-exact RewardNoise needs the wrapper's own generator
  if (call->logging != nullptr) {
    const bsx_logging_t* g = call->logging;
    if (g->steps == nullptr || g->episode == nullptr || g->total_return == nullptr || g->episode_len == nullptr ||
        g->episode_return == nullptr || g->rows == nullptr || g->n_rows == nullptr)
      return BSX_ENULL;
    if (g->n_info < 0 || g->max_rows < 0 || g->n_log_points < 0) return BSX_EINVAL;
    if (g->n_info > 0 && g->info == nullptr) return BSX_ENULL;
    if (g->n_log_points > 0 && g->log_points == nullptr) return BSX_ENULL;
  }
  return 0;
}

static inline bsx_ctl bsx_make_ctl(const bsx_call_t* call) {
  bsx_ctl c;
  c.n_lanes = call->n_lanes;
  c.seed = call->stream.seed;
  c.lane_offset = call->stream.lane_offset;
  c.step_index = call->stream.step_index;
  c.step_base = call->stream.step_base;
  c.counters = call->counters;
  c.wrap_param = call->wrap.param;
  c.wrap_param2 = call->wrap.param2;
  c.wrap_seed = call->wrap.seed;
  c.wrap_kind = call->wrap.kind;
  c.wrap_mul = call->wrap.kind == BSX_WRAP_SCALE ? call->wrap.param : 1.0;
  c.force_reset = call->force_reset;
  c.action_ring_mask = call->action_ring > 1 ? (uint32_t)call->action_ring - 1u : 0u;
  c._pad = 0;
  c.mt_state = call->stream.mt_state;
  c.mt_pos = call->stream.mt_pos;
  c.mt_gauss = call->stream.mt_gauss;
  c.mt_has_gauss = call->stream.mt_has_gauss;
  const bool wrap_mt = call->stream.mt_state != nullptr && call->wrap.kind >= BSX_WRAP_NOISE;
  c.wrap_mt_state = wrap_mt ? call->wrap.mt_state : nullptr;
  c.wrap_mt_pos = wrap_mt ? call->wrap.mt_pos : nullptr;
  c.wrap_mt_gauss = wrap_mt ? call->wrap.mt_gauss : nullptr;
  c.wrap_mt_has_gauss = wrap_mt ? call->wrap.mt_has_gauss : nullptr;
  c.reward_f64 = call->reward_f64;
  c.state_in = nullptr;
  if (call->logging != nullptr) c.log = *call->logging;
  else c.log = bsx_logging_t{};
  return c;
}

static inline int bsx_n_steps(const bsx_call_t* call) { return call->n_steps > 1 ? call->n_steps : 1; }

// magic for q = n / d via __umulhi(n, magic): exact for n < 2^20, d <= 4096
static inline uint32_t bsx_div_magic(uint32_t d) { return (uint32_t)((0x100000000ull / d) + 1ull); }

// A/B knobs.  The product library never reads the environment: the knobs exist only in the
// tuning build (`python -m bsuite_amd.build --tuning` -> libbsuite_amd_tuning.so, compiled with -DBSX_TUNING and
// loaded through BSX_NATIVE_LIB by the A/B scripts under tools/ and by the tests that cover the non-default
// settings); everywhere else every knob is its measured-best default, fixed at compile time.
static inline int bsx_env_int(const char* name, int dflt) {
#ifdef BSX_TUNING
  const char* v = getenv(name);
  return (v != nullptr && *v != '\0') ? atoi(v) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

// Launches the lane-per-thread advance kernel of a two-kernel family.
template <class Fam>
static inline int bsx_launch_advance(const typename Fam::args& a, hipStream_t st) {
  const int64_t blocks = (a.ctl.n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  if (blocks > 0x7FFFFFFF) return BSX_EINVAL;
  static const int lean_env = bsx_env_int("BSX_ADVANCE_LEAN", 1);
  const bool lean = lean_env != 0 && bsx_ctl_lean(a.ctl);
  // From two dispatch rounds of one-lane workgroups up (2^20 lanes): two lanes per thread, both lanes' loads issued up
  // front — ONE round.  Same call (profiles/r04/ab_advance_two_lanes.log): catch/0 42.5 -> 41.6 us per step, deep_sea -0.5 us;
  // equal at 2^19 lanes, 79.8 -> 78.8 at 2^21.  (0 = never)
  static const int lpt2_min_blocks = bsx_env_int("BSX_ADVANCE_LPT2_MIN_BLOCKS", 4096);
  if (lean && lpt2_min_blocks > 0 && blocks >= lpt2_min_blocks) {
    bsx_advance2_kernel<Fam><<<dim3((unsigned)((blocks + 1) / 2)), dim3(BSX_BLOCK), 0, st>>>(a);
    return 0;
  }
  if (lean) bsx_advance_kernel<Fam, true><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(a);
  // the wrapped call on the counter-based stream (the common one): the MT19937-exact generators compiled out
  else if (a.ctl.mt_state == nullptr) bsx_advance_kernel<Fam, false, 0><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(a);
  else bsx_advance_kernel<Fam, false><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(a);
  return 0;
}

// Exact 64-bit magic for n / d (4 <= d <= 4096, n < 2^52): s = floor(log2 d) - 1.
static inline bsx_div64 bsx_make_div64(uint32_t d) {
  bsx_div64 r;
  uint32_t lg = 0;
  while ((2u << lg) <= d) ++lg;             // lg = floor(log2 d)
  r.s = lg - 1;
  const unsigned __int128 num = (unsigned __int128)1 << (64 + r.s);
  r.m = (uint64_t)(num / d) + 1;
  return r;
}

// Launches the split-phase observation writer: K stores per thread, 256 threads per workgroup — each family's measured
// optimum (profiles/r01/sweep_stream_*.log), the only shape the product library contains.  The tuning build compiles the
// whole K x block-size matrix and picks by BSX_STREAM_K / BSX_STREAM_BS / BSX_STREAM_WAVE_CONTIG.
template <class HotFn, int K>
static inline int bsx_launch_hot_stream(float* obs, const int32_t* state, int64_t n_lanes, uint32_t cells,
                                        uint32_t cells_magic, HotFn fn, hipStream_t st) {
  const uint64_t total = (uint64_t)n_lanes * cells;
  // 4-byte stores for degenerate boards and for an observation slice that does not start on a 16-byte
  // boundary (rollout slice t of an odd B x cells: t*B*cells*4 bytes into the [T,B,cells] array)
  if (cells < 4u || (reinterpret_cast<uintptr_t>(obs) & 15u) != 0) {
    const uint64_t blocks = (total + BSX_BLOCK - 1) / BSX_BLOCK;
    if (blocks > 0x7FFFFFFFull) return BSX_EINVAL;
    bsx_hot_stream_tiny_kernel<HotFn><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(obs, state, n_lanes, cells, fn);
    return 0;
  }
  const bsx_div64 dv = bsx_make_div64(cells);
#ifndef BSX_TUNING
  const uint64_t per_block = (uint64_t)K * 4 * BSX_BLOCK;
  const uint64_t blocks = (total + per_block - 1) / per_block;
  if (blocks > 0x7FFFFFFFull) return BSX_EINVAL;
  bsx_hot_stream_kernel<HotFn, K, BSX_BLOCK><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(obs, state, n_lanes, cells, cells_magic, dv, fn, 1);
  return 0;
#else
  static const int k_env = bsx_env_int("BSX_STREAM_K", 0);
  const int k = k_env > 0 ? k_env : K;
  static const int bs_env = bsx_env_int("BSX_STREAM_BS", 256);
  static const int ks[] = {1, 2, 3, 4, 5, 6, 8, 12, 16};
  int kk = 1;
  for (int i = 0; i < 9; ++i) if (ks[i] <= k) kk = ks[i];
  const int bs = bs_env >= 1024 ? 1024 : bs_env >= 512 ? 512 : bs_env >= 256 ? 256 : bs_env >= 128 ? 128 : 64;
  const uint64_t per_block = (uint64_t)kk * 4 * bs;
  const uint64_t blocks = (total + per_block - 1) / per_block;
  if (blocks > 0x7FFFFFFFull) return BSX_EINVAL;
  const dim3 g((unsigned)blocks);
  // BSX_STREAM_LDS: dynamic LDS nobody uses = fewer workgroups per CU; BSX_STREAM_PACE: s_sleep rounds before the stores
  static const int wave_contig = bsx_env_int("BSX_STREAM_WAVE_CONTIG", 1) | (bsx_env_int("BSX_STREAM_PACE", 0) << 8);
  static const int lds = bsx_env_int("BSX_STREAM_LDS", 0);
#define BSX_HS(KK, BS) case KK: bsx_hot_stream_kernel<HotFn, KK, BS><<<g, dim3(BS), (size_t)lds, st>>>(obs, state, n_lanes, cells, cells_magic, dv, fn, wave_contig); break
#define BSX_HS_ALL(BS) switch (kk) { BSX_HS(1, BS); BSX_HS(2, BS); BSX_HS(3, BS); BSX_HS(4, BS); BSX_HS(5, BS); BSX_HS(6, BS); BSX_HS(8, BS); BSX_HS(12, BS); BSX_HS(16, BS); default: return BSX_EINVAL; }
  if (bs == 256) { BSX_HS_ALL(256) }
  else if (bs == 128) { BSX_HS_ALL(128) }
  else if (bs == 512) { BSX_HS_ALL(512) }
  else if (bs == 1024) { BSX_HS_ALL(1024) }
  else { BSX_HS_ALL(64) }
#undef BSX_HS_ALL
#undef BSX_HS
  return 0;
#endif
}

// ---------------------------------------------------------------------------------------------
// Grouped launch (bsx_group_t): host-side container.  Each family file fills `args`/`args2` (its
// kernel argument structs, one per segment) and the per-segment block counts, and installs `launch`.
struct bsx_group {
  int32_t family = -1;
  int32_t n = 0;
  int32_t klass = -1;                   // family-specific launch class (lanes per workgroup of small_obs), -1 unset
  size_t arg_size = 0, arg2_size = 0;
  std::vector<uint8_t> args, args2;     // n * arg_size  /  n * arg2_size (second kernel of a pair)
  std::vector<int32_t> blocks, blocks2; // workgroups of each segment in kernel 1 / kernel 2
  std::vector<uint8_t> is_set;
  std::vector<int32_t> tags;            // BSX_FAM_PAIR_MIXED: family of each segment (empty otherwise)
  int32_t* d_tags = nullptr;
  uint64_t* shared_counter = nullptr;   // BSX_FAM_SWEEP_MIXED: the call counter every segment reads; phase 0 bumps it
  uint32_t* d_ticket = nullptr;         //   ... when its last workgroup retires (device word, zero between launches)
  size_t lds_bytes = 0;                 // max dynamic LDS over segments (kernel 1)
  uint64_t* trace = nullptr;            // diagnostics (bsx_group_trace): per-workgroup [start, end, tag] of phase 0
  void* d_args = nullptr;
  void* d_args2 = nullptr;
  int32_t* d_start = nullptr;           // [n+1] exclusive prefix of blocks
  int32_t* d_start2 = nullptr;
  int2* d_map = nullptr;                // [total_blocks] (segment, local block) per workgroup, or null
  int2* d_map2 = nullptr;
  bsx_group_index index1() const { bsx_group_index gi; gi.start = d_start; gi.map = d_map; gi.n = n; return gi; }
  bsx_group_index index2() const { bsx_group_index gi; gi.start = d_start2; gi.map = d_map2; gi.n = n; return gi; }
  int64_t total_blocks = 0, total_blocks2 = 0;
  bool committed = false;
  // a two-kernel segment that has store-stream workgroups but one state column only: fine for {advance, stream} in
  // order, a race in the pipelined launch (the stream of step s would read the column the advance of s+1 writes)
  bool stream_without_alt = false;      // = any(needs_alt), evaluated at commit
  std::vector<uint8_t> needs_alt;       // per segment (mixed groups)
  std::vector<const void*> row_scratch; // per segment: the row scratch of a chain segment on the row path, else null
  std::vector<uintptr_t> rows_sorted;   // the non-null ones, sorted (frozen at commit: bsx_group_step_pipelined's aliasing check)
  int64_t split_block = -1;             // whole-sweep groups: first phase-0 workgroup of the segments that have a share of
                                        // the store stream, when those segments are the tail of the group (else -1)
  int64_t split_round = -1;             // workgroups of the split step's first launch (sweep_mixed.hip), derived on first use
  // launch(g, phase, stream): phase 0 = the first kernel (the lane advance of a two-kernel family, or the
  // whole step of a small-observation group), phase 1 = the observation stream kernel of a two-kernel
  // family (depends on phase 0 of the same group only), phase < 0 = both in order.
  int (*launch)(bsx_group*, int, hipStream_t) = nullptr;
  int n_phases = 1;
};

static inline uint64_t bsx_flat_blocks(uint64_t total_floats, int k) {
  const uint64_t per_block = (uint64_t)k * 4 * BSX_BLOCK;
  return (total_floats + per_block - 1) / per_block;
}

// Records one segment of a two-kernel family (advance args + stream-kernel segment).
template <class Fam, class HotFn>
static inline int bsx_group_put_pair(bsx_group* g, int32_t index, const typename Fam::args& a, float* obs,
                                     int32_t* state, uint32_t cells, const HotFn& fn, int k) {
  if (cells < 4u) return BSX_ERANGE;                        // degenerate boards: step them singly
  memcpy(&g->args[(size_t)index * sizeof(typename Fam::args)], &a, sizeof(a));
  bsx_stream_seg<HotFn> sg;
  sg.obs = obs; sg.state = state; sg.n_lanes = a.ctl.n_lanes; sg.cells = cells;
  sg.cells_magic = bsx_div_magic(cells); sg.dv = bsx_make_div64(cells); sg.fn = fn;
  memcpy(&g->args2[(size_t)index * sizeof(sg)], &sg, sizeof(sg));
  const uint64_t b1 = (uint64_t)(a.ctl.n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  const uint64_t b2 = bsx_flat_blocks((uint64_t)a.ctl.n_lanes * cells, k);
  if (b1 > 0x3FFFFFFFull || b2 > 0x3FFFFFFFull) return BSX_EINVAL;
  g->blocks[index] = (int32_t)b1; g->blocks2[index] = (int32_t)b2;
  g->is_set[index] = 1;
  return 0;
}

template <class Fam, class HotFn, int K>
static int bsx_group_launch_pair(bsx_group* g, int phase, hipStream_t st) {
  if (phase != 1)
    bsx_advance_group_kernel<Fam><<<dim3((unsigned)g->total_blocks), dim3(BSX_BLOCK), 0, st>>>(
        (const typename Fam::args*)g->d_args, g->index1());
  if (phase != 0)
    bsx_hot_stream_group_kernel<HotFn, K><<<dim3((unsigned)g->total_blocks2), dim3(BSX_BLOCK), 0, st>>>(
        (const bsx_stream_seg<HotFn>*)g->d_args2, g->index2());
  return (int)hipGetLastError();
}

// Common validation + bookkeeping of bsx_group_set_<family>.
static inline int bsx_group_check_set(bsx_group* g, int32_t family, int32_t index, const bsx_call_t* call,
                                      size_t arg_size, size_t arg2_size, int klass) {
  if (g == nullptr || call == nullptr) return BSX_ENULL;
  if (g->committed || g->family != family || index < 0 || index >= g->n) return BSX_EINVAL;
  if (call->stream.step_base == nullptr || call->force_reset || call->n_steps > 1 || call->n_lanes < 1)
    return BSX_EINVAL;                 // static arguments need a device-resident call counter
  if (call->obs_paint != nullptr) return BSX_EMODE;
  if (g->klass >= 0 && g->klass != klass) return BSX_EINVAL;
  g->klass = klass;
  if (g->args.empty()) {
    g->arg_size = arg_size; g->arg2_size = arg2_size;
    g->args.assign((size_t)g->n * arg_size, 0);
    g->args2.assign((size_t)g->n * arg2_size, 0);
  }
  if (g->arg_size != arg_size || g->arg2_size != arg2_size) return BSX_EINVAL;
  return 0;
}

static inline int bsx_launch_status() { return (int)hipGetLastError(); }

// One call of a two-kernel family (deep_sea, catch): step() / reset() / a rollout of T steps with outputs
// [T,B,...].  `a` comes from the family's make(); K = stores per thread of its observation stream.
//   delta mode (obs_paint)          one launch per step: advance + in-place patch
//   dense                           advance + observation stream per step
//   dense rollout with state_alt    software-pipelined: advance(0); {stream(t), advance(t+1)} for t < T-1;
//                                   stream(T-1) — T+1 launches.  The advances alternate between `state`
//                                   and `state_alt` so that the column stream(t) reads is not the one
//                                   advance(t+1) writes; the parity is chosen so that the last advance
//                                   writes `state`.  BSX_ROLLOUT_PIPELINED=0: A/B.
template <class Fam, class HotFn, int K>
static int bsx_pair_call(const typename Fam::args& a0, const bsx_call_t* call, const int32_t* action, int32_t* state,
                         bsx_timestep_t out, uint32_t cells, const HotFn& fn) {
  hipStream_t st = (hipStream_t)call->hip_stream;
  const int T = bsx_n_steps(call);
  const int64_t B = call->n_lanes;
  auto at = [&](int t) {                      // the arguments of step t: slice [t] of every [T,B,...] array
    typename Fam::args s = a0;
    const int64_t off = (int64_t)t * B;
    s.ctl.step_index = call->stream.step_index + (uint64_t)t;
    s.ctl.reward_f64 = call->reward_f64 ? call->reward_f64 + off : nullptr;
    s.action = action ? action + off : action;
    s.out.reward = out.reward + off; s.out.discount = out.discount + off; s.out.step_type = out.step_type + off;
    s.out.observation = out.observation + off * (int64_t)cells;
    return s;
  };
  const uint32_t magic = bsx_div_magic(cells);
  static const int pipe_env = bsx_env_int("BSX_ROLLOUT_PIPELINED", 1);
  static const int place = bsx_env_int("BSX_PIPELINED_PLACE", 0);     // bsx_pipe_role_of: first (measured best)
  // the fused launch uses the 16-byte store stream: every [t] slice must start on a 16-byte boundary
  const bool pipelined = pipe_env != 0 && T > 1 && call->state_alt != nullptr && call->obs_paint == nullptr &&
                         cells >= 4u && (((uint64_t)B * cells) & 3ull) == 0;
  int rc = 0;
  // Boards of at most BSX_FUSED_TILE_MAX_CELLS floats (catch's 50; a workgroup's [256 x cells] tile is then <= 128 KiB):
  // ONE fused launch per step (bsx_fused_tile_kernel) and ONE per rollout (bsx_fused_rollout_kernel) while the
  // observation array of a step is at most BSX_FUSED_TILE_MAX_MIB / BSX_FUSED_ROLLOUT_MAX_MIB = 128 MiB: catch up to
  // 2^19 lanes (105 MB: 20 vs 23 us eager, 20 vs 22 us per rollout step; 2^17: 9.4 vs 11 and 7.1 vs 9.0).  At 2^20
  // lanes (210 MB) the winner depends on the box — fused 41.9 vs 43.5 on one, 44.4-45.2 vs 43.2-43.6 on another; a
  // rollout 36.9 vs 39.6 and 43.4-45.5 vs 39.7-40.9 (its T slices lie 210 MB apart: page-mapping luck) — so the
  // decoupled pair / the pipelined rollout, steady within 2 % everywhere, keep that size (profiles/r03/
  // ab_fused_tile*.log, ab_fused_crossover.log); deep_sea N=30 (900 cells, 0.9 MiB tiles) never fuses.  The tile start
  // block*256*cells*4 is always 16-byte aligned when the slice is.  (A barrier-free variant — every wave its own
  // 64 lanes, neighbour states through ds_bpermute — measured 1-9 % slower: profiles/r03/ab_fused_wave.log; two tiles
  // per workgroup with both tiles' inputs loaded up front, i.e. one dispatch round at 2^20 lanes: 42.7-45.2 vs
  // 41.1-41.7 us for the pair, profiles/r03/ab_fused_tiles_per_wg.log.)
  static const int fused_cells = bsx_env_int("BSX_FUSED_TILE_MAX_CELLS", 128);
  static const int fused_step_mib = bsx_env_int("BSX_FUSED_TILE_MAX_MIB", 128);
  static const int fused_roll_mib = bsx_env_int("BSX_FUSED_ROLLOUT_MAX_MIB", 128);
  // (64-lane tiles up to 2^18 lanes: catch 2^15 7.3 -> 5.6 us, 2^16 8.1 -> 6.0, 2^17 9.1 -> 8.0 (r04, ordinary stores:
  // profiles/r04/ab_catch_fused_tile64.log; 2^18 was 12.3 -> 12.7 then); with their chunks non-temporal (round 6) 2^17 8.0 ->
  // 6.8 and 2^18 12.2 -> 11.55, 2^19 19.3 -> 19.8: profiles/r06/ab_catch_tile64_nt_larger_batches.log)
  static const int64_t tile64_max_lanes = bsx_env_int("BSX_FUSED_TILE64_MAX_LANES", 1 << 18);
  const int64_t step_bytes = B * (int64_t)cells * 4;
  const bool fusable = call->obs_paint == nullptr && cells >= 4u && (int)cells <= fused_cells &&
                       (((uint64_t)B * cells) & 3ull) == 0 && (reinterpret_cast<uintptr_t>(out.observation) & 15u) == 0;
  // A WRAPPED step (RewardNoise / Logging / MT19937-exact draws: catch_noise's lane advance is 18 us against the lean 9) fuses at
  // EVERY batch size: inside the one launch the heavy advance of a tile hides among the other workgroups' tile stores, in front
  // of a stand-alone stream it does not — catch_noise/0 at 2^20 lanes 48.3-48.6 -> 42.8-43.2 us per step, rollout r32 49.0-49.5
  // -> 44.6-45.6, r8 48.9-49.8 -> 41.7-42.4 (profiles/r06/ab_catch_noise_fused_at_2p20.log; the lean step in the same call:
  // 41.0 -> 43.8, ab_catch_fused_at_2p20.log); 1.5 * 2^20 lanes 69.1-70.5 -> 63.5-64.1, 2^21 91.6-92.7 -> 82.9-83.5, 2^22
  // 200.6-202.1 -> 170.8-171.3 (ab_catch_wrapped_fused_larger.log).  (MiB; 2^20 = no limit in practice)
  static const int fused_wrapped_mib = bsx_env_int("BSX_FUSED_WRAPPED_MAX_MIB", 1 << 20);
  const bool lean_f = bsx_ctl_lean(a0.ctl);
  int fused_mib = T > 1 ? fused_roll_mib : fused_step_mib;
  if (!lean_f && fused_wrapped_mib > fused_mib) fused_mib = fused_wrapped_mib;
  const bool fused = fusable && step_bytes <= ((int64_t)fused_mib << 20);
  if (fused && T > 1) {
    const dim3 grid((unsigned)((B + BSX_BLOCK - 1) / BSX_BLOCK)), block(BSX_BLOCK);
    if (lean_f) bsx_fused_rollout_kernel<Fam, true, HotFn><<<grid, block, 0, st>>>(a0, T, out.observation, cells, magic, fn);
    else bsx_fused_rollout_kernel<Fam, false, HotFn><<<grid, block, 0, st>>>(a0, T, out.observation, cells, magic, fn);
    return bsx_launch_status();
  }
  if (!pipelined || fused) {
    for (int t = 0; t < T && rc == 0; ++t) {
      const typename Fam::args s = at(t);
      if (call->obs_paint != nullptr) {
        rc = bsx_launch_advance_delta<Fam, HotFn>(s, fn, call->obs_paint, cells, st);
      } else if (fused && lean_f && B <= tile64_max_lanes) {
        // (64-lane tiles while the 256-lane grid would leave the chip under-filled: bsx_fused_tile64_kernel)
        const dim3 grid((unsigned)((B + BSX_WAVE - 1) / BSX_WAVE)), block(BSX_BLOCK);
        bsx_fused_tile64_kernel<Fam, true, HotFn><<<grid, block, 0, st>>>(s, s.out.observation, cells, magic, fn);
      } else if (fused) {
        const dim3 grid((unsigned)((B + BSX_BLOCK - 1) / BSX_BLOCK)), block(BSX_BLOCK);
        if (lean_f) bsx_fused_tile_kernel<Fam, true, HotFn><<<grid, block, 0, st>>>(s, s.out.observation, cells, magic, fn);
        else bsx_fused_tile_kernel<Fam, false, HotFn><<<grid, block, 0, st>>>(s, s.out.observation, cells, magic, fn);
      } else {
        rc = bsx_launch_advance<Fam>(s, st);
        // stores/thread x 256 threads: a sharp optimum per family (profiles/r01/sweep_stream_*.log)
        if (rc == 0) rc = bsx_launch_hot_stream<HotFn, K>(s.out.observation, state, B, cells, magic, fn, st);
      }
    }
    return rc != 0 ? rc : bsx_launch_status();
  }
  int32_t* const col[2] = {state, call->state_alt};
  auto W = [&](int t) { return col[(T - 1 - t) & 1]; };      // the column advance(t) writes; W(T-1) = state
  const uint64_t adv_blocks = (uint64_t)(B + BSX_BLOCK - 1) / BSX_BLOCK;
  const uint64_t str_blocks = ((uint64_t)B * cells + (uint64_t)K * 4 * BSX_BLOCK - 1) / ((uint64_t)K * 4 * BSX_BLOCK);
  if (adv_blocks + str_blocks > 0x7FFFFFFFull) return BSX_EINVAL;
  const bsx_div64 dv = bsx_make_div64(cells);
  const bool lean = bsx_ctl_lean(a0.ctl);
  typename Fam::args s = at(0);
  s.ctl.state_in = state; s.state = W(0);
  rc = bsx_launch_advance<Fam>(s, st);
  for (int t = 0; t + 1 < T && rc == 0; ++t) {
    s = at(t + 1);
    s.ctl.state_in = W(t); s.state = W(t + 1);
    float* obs_t = out.observation + (int64_t)t * B * (int64_t)cells;
    const dim3 grid((unsigned)(adv_blocks + str_blocks)), block(BSX_BLOCK);
    if (lean) bsx_pipelined_kernel<Fam, true, HotFn, K><<<grid, block, 0, st>>>(s, (uint32_t)adv_blocks, (uint32_t)place, obs_t, W(t), cells, magic, dv, fn);
    else bsx_pipelined_kernel<Fam, false, HotFn, K><<<grid, block, 0, st>>>(s, (uint32_t)adv_blocks, (uint32_t)place, obs_t, W(t), cells, magic, dv, fn);
  }
  if (rc == 0) rc = bsx_launch_hot_stream<HotFn, K>(out.observation + (int64_t)(T - 1) * B * (int64_t)cells, state, B, cells, magic, fn, st);
  return rc != 0 ? rc : bsx_launch_status();
}


#endif  // BSX_HOST_H_
