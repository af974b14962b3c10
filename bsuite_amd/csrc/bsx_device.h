// bsx_device.h — device-side building blocks shared by every environment-family kernel.
//
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, 256-thread workgroups (4 waves = one per
// SIMD), LDS for per-block constants and hot-cell indices, 16-byte cooperative stores for the
// observation stream.  No MFMA anywhere: the path is integer indexing + scalar f32/f64.
#ifndef BSX_DEVICE_H_
#define BSX_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bsuite_amd.h"
#include "../../include/bsx_stream.h"
#include "bsx_index.h"

#define BSX_BLOCK 256
#define BSX_WAVE 64

typedef float bsx_f4 __attribute__((ext_vector_type(4)));
typedef float bsx_f2 __attribute__((ext_vector_type(2)));

// Which per-lane OUTPUT stores take a cache policy other than plain — non-temporal in a fused rollout, write-through in an
// eager step (bsx_st below) — as bits (-DBSX_SMALL_NT=<n> in measurement builds, tools/ab_flag_lib.py):
//   1  reward / discount / step_type columns (a wave's store is one contiguous 256- / 64-byte range)
//   2  observation rows of one or two floats stored by their own thread (contiguous per wave as well)
//   4  rows staged through a wave's LDS and stored as 16-byte chunks (full lines)
//   8  rows of 4 / 6 / 8 floats stored by their own thread as 8-byte pieces at the row stride (partial lines)
//  16  rows of three floats (one 12-byte store per lane, contiguous per wave)
//  32  the 16-byte chunks of the wide rows' LDS bit-plane tiles (memory_chain, umbrella_chain)
#ifndef BSX_SMALL_NT
#define BSX_SMALL_NT 55        // 1 + 2 + 4 + 16 + 32: everything but the partial-line rows (small_obs.h has the measurements)
#endif
// Cache policy of an OUTPUT store.  PLAIN; NT = non-temporal (`nt`): the lines neither stay in L2 nor allocate in the Infinity
// Cache; WT = write-through (`sc1`): they leave L2 for the memory side at once but DO land in the Infinity Cache.
// Which one pays depends on who reads the output next (round 6, profiles/r06/ab_store_cache_policies.log,
// ab_nt_outputs_closed_loop_policy.log; cartpole at 2^20 lanes, us per step: eager alone | fused rollout | closed loop with
// a device-side linear policy reading every observation):  plain 17.6-18.1 | 9.1-9.3 | 115-116.5;  nt 15.6-16.6 | 7.5-7.8 |
// 119.5;  sc1 16.2-16.7 | 9.3-9.8 | 114.5.  A rollout's T x B outputs have no reader inside the call: NT.  An eager step()
// exists to be followed by an agent that READS the observation: its outputs must not be pushed past the Infinity Cache —
// WT keeps most of the step's gain (the step's own inputs are no longer evicted from L2) and costs the reader nothing.
enum { BSX_ST_PLAIN = 0, BSX_ST_NT = 1, BSX_ST_WT = 2 };
// (the write-through stores are inline asm — no builtin sets sc1 alone — and end with `s_nop 1`: the compiler's hazard
// recognizer does not look inside an asm string, and a VALU write of the data registers right behind a store of more than
// 8 bytes is a hazard on gfx9: without it mountain_car's 12-byte rows came out corrupted on a few lanes per wave)
#if defined(__HIP_DEVICE_COMPILE__)
template <int BYTES> struct bsx_wt_store;
template <> struct bsx_wt_store<1> { template <class T> __device__ static __forceinline__ void go(const void* p, T v) { uint32_t w = 0; __builtin_memcpy(&w, &v, 1); asm volatile("global_store_byte %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory"); } };
template <> struct bsx_wt_store<4> { template <class T> __device__ static __forceinline__ void go(const void* p, T v) { uint32_t w; __builtin_memcpy(&w, &v, 4); asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory"); } };
template <> struct bsx_wt_store<8> { template <class T> __device__ static __forceinline__ void go(const void* p, T v) { uint64_t w; __builtin_memcpy(&w, &v, 8); asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory"); } };
template <> struct bsx_wt_store<12> { template <class T> __device__ static __forceinline__ void go(const void* p, T v) { typedef uint32_t u3 __attribute__((ext_vector_type(3))); u3 w; __builtin_memcpy(&w, &v, 12); asm volatile("global_store_dwordx3 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory"); } };
template <> struct bsx_wt_store<16> { template <class T> __device__ static __forceinline__ void go(const void* p, T v) { bsx_f4 w; __builtin_memcpy(&w, &v, 16); asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(w) : "memory"); } };
#endif
template <int POLICY, class P, class V>
__device__ __forceinline__ void bsx_st(P* p, V v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (POLICY == BSX_ST_NT) __builtin_nontemporal_store((P)v, p);
  else if constexpr (POLICY == BSX_ST_WT) bsx_wt_store<sizeof(P)>::go((const void*)p, (P)v);
  else *p = (P)v;
#else
  *p = (P)v;
#endif
}
// policy of an output kind (a bit of BSX_SMALL_NT) in a fused rollout (R) / in an eager step (E)
#define BSX_POLICY_R(bit) ((BSX_SMALL_NT & (bit)) ? BSX_ST_NT : BSX_ST_PLAIN)
#if defined(BSX_AB_EAGER_NT)           // measurement builds only: the eager step's outputs non-temporal as well
#define BSX_POLICY_E(bit) ((BSX_SMALL_NT & (bit)) ? BSX_ST_NT : BSX_ST_PLAIN)
#else
#define BSX_POLICY_E(bit) ((BSX_SMALL_NT & (bit)) ? BSX_ST_WT : BSX_ST_PLAIN)
#endif
#ifndef BSX_TILE64_POLICY
#define BSX_TILE64_POLICY BSX_ST_WT     // the 64-lane tiles of a small catch batch (an eager step: write-through)
#endif

// Per-call values every kernel needs, flattened out of bsx_call_t on the host.
struct bsx_ctl {
  int64_t n_lanes;
  uint64_t seed;
  uint64_t lane_offset;
  uint64_t step_index;
  const uint64_t* step_base;
  uint64_t* counters;
  double wrap_param;
  double wrap_param2;       // stacked wrappers: the outer wrapper's parameter
  double wrap_mul;          // RewardScale alone: its scale; no wrapper: 1.0 (x * 1.0 == x bit for bit) — the NOISE = 0
                            // instantiations multiply unconditionally instead of branching on wrap_kind
  uint64_t wrap_seed;
  int32_t wrap_kind;
  int32_t force_reset;
  uint32_t action_ring_mask; // R - 1 for an action ring of R = 2^k rows [R, n_lanes] (bsx_call_t.action_ring), else 0
  uint32_t _pad;
  uint32_t* mt_state;       // MT19937-exact mode: [624, n_lanes] generator states, else nullptr
  int32_t* mt_pos;          // [n_lanes]
  double* mt_gauss;         // [n_lanes] cached second normal of the env generator (nullable)
  int32_t* mt_has_gauss;
  uint32_t* wrap_mt_state;  // RewardNoise's own generator in MT19937-exact mode (all four or none)
  int32_t* wrap_mt_pos;
  double* wrap_mt_gauss;
  int32_t* wrap_mt_has_gauss;
  double* reward_f64;       // optional f64 copy of the reward column (scalar dm_env view), else nullptr
  const int32_t* state_in;  // two-kernel families: the packed state column the advance READS (nullptr: `state`)
  bsx_logging_t log;        // log.steps == nullptr: logging off
};

// No Logging wrapper, no RewardNoise, counter-based draws, no f64 reward copy (the scalar dm_env view's): the call
// the lean instantiations serve.
__host__ __device__ __forceinline__ bool bsx_ctl_lean(const bsx_ctl& c) {
  return c.log.steps == nullptr && c.wrap_kind < BSX_WRAP_NOISE && c.mt_state == nullptr && c.reward_f64 == nullptr;
}

// bsuite_info accumulators (f64 columns [K, B], a lane owns its slots).  An update can be a NO-RETURN hardware atomic
// (global_atomic_add_f64) where nothing in the launch reads the column back (no fused Logging rows): the same IEEE add,
// executed at the L2, and the wave does not wait for a dependent load of a cold column from HBM before it stores.
// Measured in every family, same call (profiles/r04/ab_info_atomics.log): it pays where a few lanes of a wave update now
// and then — memory_chain with long episodes, memory_len 13.3 -> 12.0 us per step — and costs where updates are dense
// (bandit 9.6 -> 11.2, memory_size 44.1 -> 47.4, umbrella_distract 98 -> 103, deep_sea's headline 589 -> 595) or is
// neutral (cartpole, mountain_car, mnist): adopted for memory_chain at L >= 8 only.  The columns must then be ordinary
// device memory (hardware f64 atomics do not reach fine-grained host mappings): include/bsuite_amd.h says so.
// `quiet`: no Logging wrapper reads the column in this launch (bsx_track snapshots it).
__device__ __forceinline__ void bsx_info_add(bool quiet, double* p, double v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BSX_NO_INFO_ATOMICS)
  if (quiet) { __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)p, v); return; }
#endif
  *p += v;
}
template <int LOG>
__device__ __forceinline__ bool bsx_info_quiet(const bsx_ctl& c) { return LOG == 0 || (LOG == -1 && c.log.steps == nullptr); }

__device__ __forceinline__ uint64_t bsx_step_of(const bsx_ctl& c) {
  return c.step_index + (c.step_base ? *c.step_base : 0ull);
}

// The action of output element `oi` (lane i of a step() call; t*B + i inside a fused rollout) on call `step`.
// With an action ring (bsx_call_t.action_ring = R) the call reads row (step mod R) of `action` [R, B] — the
// offset is uniform per workgroup (scalar unit) and zero without a ring.
__device__ __forceinline__ int bsx_action(const bsx_ctl& c, const int32_t* __restrict__ action, int64_t oi, uint64_t step) {
  return action[oi + (int64_t)(step & (uint64_t)c.action_ring_mask) * c.n_lanes];
}

// Opens lane i's environment draw stream for this call: the counter-based stream, or — in
// MT19937-exact mode — the lane's own RandomState carried in HBM.  bsx_draws_end writes the
// generator position back (the state words are updated in place by the twist).
// MT: -1 decide at run time (c.mt_state != nullptr), 0 the MT19937 mode is compiled out.
template <int MT = -1>
__device__ __forceinline__ void bsx_draws_begin(bsx_draws* d, const bsx_ctl& c, int64_t i, uint64_t lane,
                                                uint64_t step) {
  bsx_draws_init(d, c.seed, lane, step, BSX_STREAM_ENV);
  if (MT != 0 && c.mt_state != nullptr) {
    d->mt = c.mt_state + i;
    d->mt_stride = c.n_lanes;
    d->mt_pos = c.mt_pos[i];
    if (c.mt_gauss != nullptr) { d->mt_has_gauss = c.mt_has_gauss[i]; d->mt_gauss = c.mt_gauss[i]; }
  }
}
template <int MT = -1>
__device__ __forceinline__ void bsx_draws_end(const bsx_draws* d, const bsx_ctl& c, int64_t i) {
  if (MT != 0 && c.mt_state != nullptr) {
    c.mt_pos[i] = d->mt_pos;
    if (c.mt_gauss != nullptr) { c.mt_has_gauss[i] = d->mt_has_gauss; c.mt_gauss[i] = d->mt_gauss; }
  }
}

// Reward epilogue of utils/wrappers.py:275-283 (RewardNoise) and :338-346 (RewardScale): non-FIRST
// lanes only, evaluated in f64 like the reference, result cast to f32 once.
// NOISE = 0 compiles the RewardNoise branch out: its ~100 f64 polynomial constants are otherwise
// hoisted into VGPRs ahead of the T-step rollout loop and cost two thirds of the occupancy.
// MT = 0: the call draws from the counter-based stream (bsx_make_ctl sets wrap_mt_state only in MT19937-exact mode): the
// wrapper's own generator — twist, legacy gauss, libm log — is compiled out.
template <int NOISE = -1, int MT = -1>
__device__ __forceinline__ double bsx_wrap_reward(const bsx_ctl& c, int64_t i, uint64_t lane, uint64_t step,
                                                  double reward) {
  BSX_NO_CONTRACT
  if (NOISE == 0) return reward * c.wrap_mul;        // RewardScale or nothing: no branch (bsx_ctl.wrap_mul)
  if (c.wrap_kind == BSX_WRAP_SCALE) return reward * c.wrap_param;
  if (NOISE != 0 && c.wrap_kind >= BSX_WRAP_NOISE) {
    // RewardNoise alone, or stacked with RewardScale in either order (each wrapper acts on what the one
    // inside it returned): SCALE_NOISE = r*s + sigma*z, NOISE_SCALE = (r + sigma*z)*s
    const double sigma = c.wrap_kind == BSX_WRAP_SCALE_NOISE ? c.wrap_param2 : c.wrap_param;
    if (c.wrap_kind == BSX_WRAP_SCALE_NOISE) reward = reward * c.wrap_param;
    bsx_draws w;
    bsx_draws_init(&w, c.wrap_seed, lane, step, BSX_STREAM_WRAP);
    double z;
    if (MT != 0 && c.wrap_mt_state != nullptr) {       // MT19937-exact mode: the wrapper's own RandomState (wrappers.py:267)
      w.mt = c.wrap_mt_state + i;
      w.mt_stride = c.n_lanes;
      w.mt_pos = c.wrap_mt_pos[i];
      w.mt_has_gauss = c.wrap_mt_has_gauss[i];
      w.mt_gauss = c.wrap_mt_gauss[i];
      z = bsx_normal(&w);
      c.wrap_mt_pos[i] = w.mt_pos;
      c.wrap_mt_has_gauss[i] = w.mt_has_gauss;
      c.wrap_mt_gauss[i] = w.mt_gauss;
    } else {
      z = bsx_normal(&w);
    }
    reward = reward + sigma * z;
    if (c.wrap_kind == BSX_WRAP_NOISE_SCALE) reward = reward * c.wrap_param2;
    return reward;
  }
  return reward;
}

// `Logging._track` + `_log_bsuite_data` of bsuite/utils/wrappers.py:85-125 for one lane.  `reward`
// is the f64 reward the outermost wrapper returned (0.0 on FIRST, where the reference adds
// `timestep.reward or 0.0`).  The family's step function has already applied this call's updates
// to its info columns, so the snapshot sees the same bsuite_info() the reference logs.
__device__ __forceinline__ void bsx_track(const bsx_ctl& c, int64_t i, int type, double reward) {
  BSX_NO_CONTRACT
  const bsx_logging_t& g = c.log;
  int64_t steps = g.steps[i], episode = g.episode[i], ep_len = g.episode_len[i];
  double total = g.total_return[i], ep_ret = g.episode_return[i];
  if (type != BSX_FIRST) { steps += 1; ep_len += 1; }                    // :87-89
  if (type == BSX_LAST) episode += 1;                                     // :90-91
  ep_ret += reward;                                                       // :92
  total += reward;                                                        // :93
  bool log = false;
  const int64_t key = g.log_by_step ? steps : episode;
  if (g.log_by_step || type == BSX_LAST) {                                // :96-102
    log = g.log_every != 0;
    int lo = 0, hi = g.n_log_points;                                      // _logarithmic_logging :140-147
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const int64_t v = g.log_points[mid];
      if (v == key) { log = true; break; }
      if (v < key) lo = mid + 1; else hi = mid;
    }
  }
  if (log) {                                                              // _log_bsuite_data :112-125
    const int n = g.n_rows[i];
    g.n_rows[i] = n + 1;
    if (n < g.max_rows) {
      const int w = 5 + g.n_info;
      double* row = g.rows + ((int64_t)i * g.max_rows + n) * w;
      row[0] = (double)steps; row[1] = (double)episode; row[2] = total;
      row[3] = (double)ep_len; row[4] = ep_ret;
      for (int k = 0; k < g.n_info; ++k) row[5 + k] = g.info[(int64_t)k * c.n_lanes + i];
    }
  }
  if (type == BSX_LAST) { ep_len = 0; ep_ret = 0.0; }                     // :105-107
  g.steps[i] = steps; g.episode[i] = episode; g.episode_len[i] = ep_len;
  g.total_return[i] = total; g.episode_return[i] = ep_ret;
}

// The scalar TimeStep fields of one lane: wrapper epilogue + Logging bookkeeping, values only.
// LOG: -1 decide at run time (c.log.steps != nullptr), 0 logging compiled out, 1 always track.
// `oi` is the output element (== i for step(); t*B + i inside a fused rollout).
// F64 = false: the lean instantiations (bsx_ctl_lean: reward_f64 == nullptr) compile the f64 reward copy out.
template <int LOG = -1, int NOISE = -1, bool F64 = true, int MT = -1>
__device__ __forceinline__ void bsx_emit_values(const bsx_ctl& c, int64_t i, int64_t oi, uint64_t lane, uint64_t step,
                                                int type, double reward, float& r, float& d) {
  r = 0.0f; d = 1.0f;         // FIRST: dm_env.restart has reward/discount None -> 0 / 1 in a batch
  double wrapped = 0.0;
  if (type != BSX_FIRST) {
    wrapped = bsx_wrap_reward<NOISE, MT>(c, i, lane, step, reward);
    r = (float)wrapped;
    d = (type == BSX_LAST) ? 0.0f : 1.0f;
  }
  if (F64 && c.reward_f64 != nullptr) c.reward_f64[oi] = wrapped;
  if (LOG == 1 || (LOG == -1 && c.log.steps != nullptr)) bsx_track(c, i, type, wrapped);
}

// Writes the scalar TimeStep fields of one lane (coalesced: lane i -> element oi of each column;
// oi == i for step(), oi == t*B + i inside a fused T-step rollout).
// POLICY: other than plain stores (BSX_SMALL_NT bit 1) ONLY where every lane of a wave emits, lane by lane (the small-observation
// kernels): a wave's store is then one contiguous range.  A lone emitting thread (the writer threads of deep_sea's
// single-launch step: one lane per 225 threads) must not — 4-byte non-temporal stores scattered over the grid took that
// kernel from 81 to 125 us at 2^17 lanes.
template <int LOG = -1, int NOISE = -1, bool F64 = true, int MT = -1, int POLICY = BSX_ST_PLAIN>
__device__ __forceinline__ void bsx_emit_at(const bsx_ctl& c, const bsx_timestep_t& out, int64_t i, int64_t oi,
                                            uint64_t lane, uint64_t step, int type, double reward) {
  float r, d;
  bsx_emit_values<LOG, NOISE, F64, MT>(c, i, oi, lane, step, type, reward, r, d);
  bsx_st<POLICY>(&out.reward[oi], r);
  bsx_st<POLICY>(&out.discount[oi], d);
  bsx_st<POLICY>(&out.step_type[oi], (int8_t)type);
}
__device__ __forceinline__ void bsx_emit(const bsx_ctl& c, const bsx_timestep_t& out, int64_t i,
                                         uint64_t lane, uint64_t step, int type, double reward) {
  bsx_emit_at(c, out, i, i, lane, step, type, reward);
}

// Termination / restart masks by wavefront ballot.  Each wave popcounts its LAST / FIRST masks
// into two LDS words (s_cnt, zeroed by the caller before the phase); after the block's barrier one
// thread flushes them with one global atomic per mask into the block's shard of the counter array
// (bsx_flush_counts).  A single device-wide word would serialise ~16k same-address atomics on the
// steps where every lane terminates (~200 us measured on catch at B=2^20).  Inactive lanes (beyond
// n_lanes) must pass type = -1.
__device__ __forceinline__ void bsx_count_types(const bsx_ctl& c, int type, unsigned int* s_cnt) {
  if (c.counters == nullptr) return;
  unsigned long long last = __ballot(type == BSX_LAST);
  unsigned long long first = __ballot(type == BSX_FIRST);
  if ((threadIdx.x & (BSX_WAVE - 1)) == 0) {
    if (last) atomicAdd(&s_cnt[0], (unsigned int)__popcll(last));
    if (first) atomicAdd(&s_cnt[1], (unsigned int)__popcll(first));
  }
}

// Measurement builds only (-DBSX_TRACE_LIFE, tools/sweep_phase0_trace.py --life): thread 0 of every workgroup of the
// sweep's phase 0 stamps the wall clock at up to 8 points of its life into bsx_life_trace_ptr[8 * blockIdx.x + k].
#ifdef BSX_TRACE_LIFE
static __device__ uint64_t* bsx_life_trace_ptr;
#define BSX_LIFE(k) do { if (threadIdx.x == 0 && bsx_life_trace_ptr != nullptr) bsx_life_trace_ptr[8 * (size_t)blockIdx.x + (k)] = wall_clock64(); } while (0)
#define BSX_LIFE_AFTER_V(k, v) do { asm volatile("" :: "v"(v)); BSX_LIFE(k); } while (0)
#define BSX_LIFE_AFTER_S(k, v) do { asm volatile("" :: "s"(v)); BSX_LIFE(k); } while (0)
#else
#define BSX_LIFE(k) do {} while (0)
#define BSX_LIFE_AFTER_V(k, v) do {} while (0)
#define BSX_LIFE_AFTER_S(k, v) do {} while (0)
#endif

// The LAST barrier of a workgroup, in front of bsx_flush_counts: it has to order the waves' LDS counter updates and
// nothing else.  __syncthreads() is a fence + barrier, and on gfx9 its release half waits for vmcnt(0) — every wave
// sat through the acknowledgements of its final stores (1-3 us behind a saturated memory system) before it could
// arrive, and the workgroup's slot stayed taken for that long; the waves may simply END with their stores in flight
// (the kernel's completion covers them).  BSX_FINAL_BARRIER_FENCED restores the old form for A/B builds.
__device__ __forceinline__ void bsx_final_barrier() {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BSX_FINAL_BARRIER_FENCED)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
  __syncthreads();
#endif
}

// Call after a __syncthreads() that follows every wave's bsx_count_types.
__device__ __forceinline__ void bsx_flush_counts(const bsx_ctl& c, const unsigned int* s_cnt,
                                                 uint32_t block_id = 0xFFFFFFFFu) {
  if (c.counters == nullptr || threadIdx.x != 0) return;
  if (block_id == 0xFFFFFFFFu) block_id = blockIdx.x;
  unsigned long long* shard = (unsigned long long*)c.counters +
                              (size_t)(block_id & (BSX_COUNTER_SHARDS - 1)) * BSX_COUNTER_STRIDE;
  if (s_cnt[0]) atomicAdd(&shard[0], (unsigned long long)s_cnt[0]);
  if (s_cnt[1]) atomicAdd(&shard[1], (unsigned long long)s_cnt[1]);
}

// Error word: the batched kernels never fault on an action outside the action_spec —
// bandit / discounting_chain clamp, catch moves the paddle by (action - 1) and clips — where the
// reference raises IndexError; every such lane-step is counted in word 2 of the lane's counter shard
// so a caller can assert `invalid_action_count() == 0` without a per-step host check.  Rare path:
// a plain global atomic.
__device__ __forceinline__ void bsx_note_invalid_action(const bsx_ctl& c, int64_t i) {
  if (c.counters == nullptr) return;
  unsigned long long* shard = (unsigned long long*)c.counters +
                              (size_t)((uint64_t)(i >> 8) & (BSX_COUNTER_SHARDS - 1)) * BSX_COUNTER_STRIDE;
  atomicAdd(&shard[2], 1ull);
}

// Grouped launch: which segment a workgroup belongs to, and its index inside that segment.
// `map` (device memory, one (segment, local block) pair per workgroup of the launch) answers with ONE
// scalar load; without it (launches too large to tabulate) the segment is the largest s with
// start[s] <= b, found by binary search over the exclusive prefix sums `start` (n+1 entries) — a
// chain of ~log2(n) dependent loads in front of every workgroup, which cost the store-stream
// kernels 10-35 % when the sweep's 16 KiB workgroups each paid it (profiles/r01/ab_sweep_modes.log).
struct bsx_group_index {
  const int32_t* start;
  const int2* map;
  int n;
};

struct bsx_group_slot { int seg; uint32_t block; int tag; };   // tag: the segment's family in a mixed group, -1 = look it up

__device__ __forceinline__ bsx_group_slot bsx_group_find(const bsx_group_index& gi, int b) {
  bsx_group_slot r;
  if (gi.map != nullptr) {
    const int2 v = gi.map[b];
    r.seg = v.x & 0x00FFFFFF; r.block = (uint32_t)v.y; r.tag = ((v.x >> 24) & 0x7F) - 1;
    return r;
  }
  r.tag = -1;
  int lo = 0, hi = gi.n;         // invariant: start[lo] <= b < start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gi.start[mid] <= b) lo = mid; else hi = mid;
  }
  r.seg = lo; r.block = (uint32_t)(b - gi.start[lo]);
  return r;
}

// Fam::advance<LEAN> — with the family's MT19937-exact draws compiled out as well when NOMT (deep_sea's extra template
// parameter sits between the two).
template <class Fam, bool LEAN, bool NOMT>
__device__ __forceinline__ int bsx_fam_advance(const typename Fam::args& a, const typename Fam::shared& s, int64_t i, uint64_t lane,
                                               uint64_t step, int32_t st, int act, int32_t& nst, double& reward) {
  return Fam::template advance_nomt<LEAN, NOMT>(a, s, i, lane, step, st, act, nst, reward);
}

// Advance kernel of the two-kernel families (deep_sea, catch): one lane per thread, coalesced
// column loads/stores.  At B=2^20 it moves only 22 MB and sits at the ~8 us launch/latency floor of
// any 2^20-lane kernel; a 4-lanes-per-thread variant with 16-byte column accesses measured the
// same for catch and slower for deep_sea, whose per-lane Philox draw then runs 4x serially
// (profiles/r01/ab_advance_vec4.log).
//
// Fam provides: struct args { bsx_ctl ctl; const int32_t* action; int32_t* state; bsx_timestep_t out;
//                             double* info; ... };  struct shared;  static stage(args, shared&);
//   static int advance(args, shared, i, lane, step, st, act, nst&, reward&)
// LEAN: no Logging wrapper, no RewardNoise, counter-based draws — those branches are compiled out (the
// launcher picks it when the call has none of them).
// MT = 0 (with LEAN = false): a wrapped call on the counter-based stream — the MT19937-exact generators of the
// environment and of RewardNoise are compiled out (the whole-sweep group, which holds no MT19937-exact segment).
// LPT = 2: TWO lanes per thread — lanes b*512 + t and b*512 + 256 + t, the loads of both issued before the first use:
// half as many workgroups, i.e. ONE dispatch round at 2^20 lanes instead of two.
template <class Fam, bool LEAN = false, int MT = -1, int LPT = 1>
__device__ __forceinline__ void bsx_advance_body(const typename Fam::args& a, uint32_t block_id,
                                                 typename Fam::shared& s_fam, unsigned int* s_cnt,
                                                 int32_t* s_state = nullptr) {
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  Fam::stage(a, s_fam);
  __syncthreads();
  if constexpr (LPT == 2) {
    const uint64_t step = bsx_step_of(a.ctl);
    int64_t i[2];
    bool mine[2];
    int act[2], type[2] = {-1, -1};
    int32_t st[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      i[h] = (int64_t)block_id * (2 * BSX_BLOCK) + h * BSX_BLOCK + threadIdx.x;
      mine[h] = i[h] < a.ctl.n_lanes;
      act[h] = 0; st[h] = 0;
      if (mine[h]) {
        if (!a.ctl.force_reset) act[h] = bsx_action(a.ctl, a.action, i[h], step);
        st[h] = a.ctl.state_in != nullptr ? a.ctl.state_in[i[h]] : a.state[i[h]];
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (mine[h]) {
        const uint64_t lane = a.ctl.lane_offset + (uint64_t)i[h];
        int32_t nst; double reward;
        type[h] = bsx_fam_advance<Fam, LEAN, MT == 0>(a, s_fam, i[h], lane, step, st[h], act[h], nst, reward);
        a.state[i[h]] = nst;
        if (LEAN) bsx_emit_at<0, 0, false>(a.ctl, a.out, i[h], i[h], lane, step, type[h], reward);
        else bsx_emit_at<-1, -1, true, MT>(a.ctl, a.out, i[h], i[h], lane, step, type[h], reward);
      }
      bsx_count_types(a.ctl, type[h], s_cnt);
    }
    bsx_final_barrier();
    bsx_flush_counts(a.ctl, s_cnt, block_id);
    return;
  }
  const int64_t i = (int64_t)block_id * BSX_BLOCK + threadIdx.x;
  int type = -1;
  if (i < a.ctl.n_lanes) {
    const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;
    const uint64_t step = bsx_step_of(a.ctl);
    int32_t nst; double reward;
    BSX_LIFE_AFTER_S(2, (uint32_t)step);                    // the argument slot and the call counter have arrived
    const int act = a.ctl.force_reset ? 0 : bsx_action(a.ctl, a.action, i, step);
    const int32_t st = a.ctl.state_in != nullptr ? a.ctl.state_in[i] : a.state[i];
    BSX_LIFE_AFTER_V(3, st + act);                              // ... the lane's state and action
    type = bsx_fam_advance<Fam, LEAN, MT == 0>(a, s_fam, i, lane, step, st, act, nst, reward);
    BSX_LIFE_AFTER_V(4, nst);                                   // ... computed
    a.state[i] = nst;
    if (s_state != nullptr) s_state[threadIdx.x] = nst;     // fused small-batch step: the tile streamer reads it from LDS
    if (LEAN) bsx_emit_at<0, 0, false>(a.ctl, a.out, i, i, lane, step, type, reward);
    else bsx_emit_at<-1, -1, true, MT>(a.ctl, a.out, i, i, lane, step, type, reward);
  }
  BSX_LIFE(5);                                                  // stores issued
  bsx_count_types(a.ctl, type, s_cnt);
  bsx_final_barrier();
  BSX_LIFE(6);
  bsx_flush_counts(a.ctl, s_cnt, block_id);
}

template <class Fam, bool LEAN = false, int MT = -1>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_advance_kernel(const typename Fam::args a) {
  __shared__ typename Fam::shared s_fam;
  __shared__ unsigned int s_cnt[2];
  bsx_advance_body<Fam, LEAN, MT>(a, blockIdx.x, s_fam, s_cnt);
}
// (the WRAPPED call with two lanes per thread — Logging / RewardNoise instantiation — measured in round 6 and not kept:
// catch_noise/0 49.4-50.0 -> 51.4-51.7 us per step, its normal draws then run twice in series; deep_sea under Logging equal;
// profiles/r06/ab_wrapped_advance_two_lanes.log, catch_noise_kernel_stats.csv: the wrapped advance is 18.0 us, the lean one 9.1)
template <class Fam>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_advance2_kernel(const typename Fam::args a) {     // two lanes per thread, lean
  __shared__ typename Fam::shared s_fam;
  __shared__ unsigned int s_cnt[2];
  bsx_advance_body<Fam, true, -1, 2>(a, blockIdx.x, s_fam, s_cnt);
}

template <class Fam>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_advance_group_kernel(const typename Fam::args* __restrict__ table,
                                                                      const bsx_group_index gi) {
  __shared__ typename Fam::shared s_fam;
  __shared__ unsigned int s_cnt[2];
  const bsx_group_slot w = bsx_group_find(gi, (int)blockIdx.x);
  bsx_advance_body<Fam>(table[w.seg], w.block, s_fam, s_cnt);
}

// n / cells for n < 2^20 via the host-built magic (bsx_div_magic); cells == 1 has no 32-bit magic.
__device__ __forceinline__ uint32_t bsx_div_cells(uint32_t n, uint32_t cells, uint32_t cells_magic) {
  return cells == 1u ? n : __umulhi(n, cells_magic);
}

// ---------------------------------------------------------------------------------------------
// Observation stream kernel: a pure store stream over the whole [B x cells] observation array,
// decoupled from the lane-advance kernel.  Block b writes the K*4 KiB run of floats
// [b*K*1024, (b+1)*K*1024): K lane-interleaved 16-byte stores per thread, blocks in address order,
// no loop — the shape of the fastest fill kernels measured on MI355X (profiles/r01/
// store_calibration*.log).  The hot cells are recomputed from the packed state column the advance
// kernel has just written (4 B per lane, L2-resident).
//
// Index math: the block's first float F0 = b*K*1024 is split once per block into (lane, offset)
// with an exact 64-bit magic division on the scalar unit; per-thread offsets stay < 2^14 so a
// 32-bit magic division is exact.
struct bsx_div64 {            // n / d = __umul64hi(n, m) >> s, exact for n*d < 2^(64+s), d >= 4
  uint64_t m;
  uint32_t s;
};

template <class HotFn>
struct bsx_stream_seg {            // one segment's arguments of the observation stream kernel
  float* obs;
  const int32_t* state;
  int64_t n_lanes;
  uint32_t cells;
  uint32_t cells_magic;
  bsx_div64 dv;
  HotFn fn;
};

// NT: non-temporal stores.  Stand-alone they cost the stream 8-10 % (deep_sea 586 -> 636-652 us per step, catch 41.0 -> 42.5:
// r01, and again profiles/r06/ab_nontemporal_stores.log); inside the sweep's mixed stream they are what keeps everything ELSE —
// the state columns the stream itself reads, the small families' columns, actions and tables — in cache while 842 MB of
// observations pass: closed-loop sweep step 161-162.5 -> 157-161.5 us, open-loop 157-158 -> 150-151.6 (pair_mixed.h).
template <class HotFn, int K, int BS, bool NT = false>
__device__ __forceinline__ void bsx_hot_stream_body(float* __restrict__ obs,
                                                    const int32_t* __restrict__ state,
                                                    int64_t n_lanes, uint32_t cells,
                                                    uint32_t cells_magic, bsx_div64 dv,
                                                    const HotFn& fn, uint32_t block_id, int wave_contig = 1) {
#ifdef BSX_TUNING
  const int pace = wave_contig >> 8;                                     // BSX_STREAM_PACE (bsx_launch_hot_stream)
  wave_contig &= 1;
#endif
  const uint64_t total = (uint64_t)n_lanes * cells;                      // floats in the array
  const uint64_t F0 = (uint64_t)block_id * (uint64_t)(K * 4 * BS);
  const uint64_t lane_b = __umul64hi(F0, dv.m) >> dv.s;                  // uniform
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);                  // < cells
  const bool aligned = (cells & 3u) == 0;
  bsx_f4* __restrict__ o4 = reinterpret_cast<bsx_f4*>(obs + F0);
  const int32_t* __restrict__ st = state + lane_b;
  const uint64_t lanes_left = (uint64_t)n_lanes - lane_b;                // lanes at or after lane_b

  uint32_t dl[K];
  int r0[K];
  int32_t s0[K], s1[K];
  bool live[K];
  // (Measured in round 4, profiles/r04/ab_stream_without_state_loads.log / ab_stream_occupancy_pace.log: WITHOUT these loads
  // the stream is 9 % slower; with fewer than 8 resident workgroups per CU it is slower at every step (7: +5 %, 4: +26 %);
  // s_sleep pacing between the loads and the stores never helps.)
  // (One state load per WAVE — lane j fetches row (first row of the wave) + j — handed to the chunks through
  // ds_bpermute instead of one mostly redundant load per chunk: 7 % slower, deep_sea 593 -> 636 us; the K stores then
  // all hang on one load + a cross-lane hop.  profiles/r03/ab_stream_wave_state_load.log)
#pragma unroll
  for (int u = 0; u < K; ++u) {
    // chunk within the block: each wave owns K consecutive KiB (store u of wave w covers KiB
    // w*K + u) — measured +3 % on deep_sea over the block-interleaved order u*BS + tid
    // (profiles/r01/ab_stream_wave_contig.log)
    const uint32_t c = wave_contig ? ((threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63))
                                   : (threadIdx.x + u * BS);
    const uint32_t f = r_b + (c << 2);                                   // float offset from lane_b's row start
    dl[u] = __umulhi(f, cells_magic);
    r0[u] = (int)(f - dl[u] * cells);
    live[u] = F0 + ((uint64_t)c << 2) + 3 < total;
#if defined(BSX_ABLATE_STREAM_LOADS)      // measurement builds only (tools/ab/): the store stream without its state loads
    s0[u] = live[u] ? (int32_t)(dl[u] * 7u + (uint32_t)lane_b) & 0x0F0F : 0;
    s1[u] = s0[u] + 1;
#else
    s0[u] = live[u] ? st[dl[u]] : 0;
    s1[u] = (live[u] && !aligned && (uint64_t)dl[u] + 1 < lanes_left) ? st[dl[u] + 1] : 0;
#endif
  }
#ifdef BSX_TUNING
  for (int q = 0; q < pace; ++q) __builtin_amdgcn_s_sleep(16);           // 16 x 64 clocks per round
#endif
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (!live[u]) continue;
    int ha, hb;
    fn(s0[u], ha, hb);
    const int a0 = ha < 0 ? -1 : ha - r0[u], b0 = hb < 0 ? -1 : hb - r0[u];
    bsx_f4 v;
    v.x = (a0 == 0 || b0 == 0) ? 1.0f : 0.0f;
    v.y = (a0 == 1 || b0 == 1) ? 1.0f : 0.0f;
    v.z = (a0 == 2 || b0 == 2) ? 1.0f : 0.0f;
    v.w = (a0 == 3 || b0 == 3) ? 1.0f : 0.0f;
    const int over = (int)cells - r0[u];
    if (!aligned && over < 4) {             // elements j >= over belong to the next lane's row
      int na, nb;
      fn(s1[u], na, nb);
      const int a1 = na < 0 ? -1 : na + over, b1 = nb < 0 ? -1 : nb + over;
      if (over <= 1) v.y = (a1 == 1 || b1 == 1) ? 1.0f : 0.0f;
      if (over <= 2) v.z = (a1 == 2 || b1 == 2) ? 1.0f : 0.0f;
      v.w = (a1 == 3 || b1 == 3) ? 1.0f : 0.0f;
    }
    {
      bsx_f4* dst = &o4[wave_contig ? ((threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63)) : (threadIdx.x + u * BS)];
      if (NT) __builtin_nontemporal_store(v, dst);
      else *dst = v;
    }
  }
  // ragged tail (< 4 floats) of an odd-sized array: the block that contains the array's end
  const uint64_t tail0 = total & ~3ull;
  if (tail0 != total && tail0 >= F0 && tail0 < F0 + (uint64_t)(K * 4 * BS) && threadIdx.x < 3) {
    const uint64_t F = tail0 + threadIdx.x;
    if (F < total) {
      const uint32_t f = r_b + (uint32_t)(F - F0);
      const uint32_t d = __umulhi(f, cells_magic);
      const int r = (int)(f - d * cells);
      int ha, hb;
      fn(st[d], ha, hb);
      obs[F] = (ha == r || hb == r) ? 1.0f : 0.0f;
    }
  }
}

template <class HotFn, int K, int BS>
__global__ void __launch_bounds__(BS) bsx_hot_stream_kernel(float* __restrict__ obs,
                                                                   const int32_t* __restrict__ state,
                                                                   int64_t n_lanes, uint32_t cells,
                                                                   uint32_t cells_magic, bsx_div64 dv,
                                                                   HotFn fn, int wave_contig) {
  bsx_hot_stream_body<HotFn, K, BS>(obs, state, n_lanes, cells, cells_magic, dv, fn, blockIdx.x, wave_contig);
}

// ---------------------------------------------------------------------------------------------
// Small batches (a rank's share of a strong-scaled batch: 2^20 / 8 lanes of catch are 26 MB of boards): ONE
// launch per step.  A workgroup advances its 256 lanes, leaves their new packed states in LDS, and after one
// barrier streams exactly those 256 boards — [256 x cells] floats, contiguous in HBM — as 16-byte chunks with the
// hot cells decoded from LDS.  At 2^20 lanes the decoupled pair wins (no store waits behind a barrier: the
// barrier'd single-kernel designs lost 15-35 % there); when the whole step is a few microseconds
// the second launch and the state column's round trip through L2 are what is left to remove.
// POLICY: the chunk stores' cache policy (bsx_st).  The fused rollout's tiles are non-temporal (catch at 2^17 lanes: 7.2 -> 5.5
// us per step; 2^18 / 2^19 equal / -4 %).  The 64-lane tiles of an EAGER step of a small batch (up to 2^18 lanes) are
// write-through: catch at 2^17 lanes 7.96 -> 6.32 us per step (non-temporal: 6.77), at 2^18 12.5 -> 10.9 (11.8), and the closed
// loop with a device-side policy reading the boards 37.4 -> 35-36 / 46.2 -> 44.7 us (non-temporal: 48.1, WORSE than plain).
// The 256-lane tiles of an eager step (2^19 lanes) are write-through too: 19.4 -> 18.6 us, closed loop 78.9 -> 77.7 (non-temporal:
// 23.3).  (profiles/r06/ab_nt_wide_rows_and_small_batches.log, ab_eager_output_policy.log, ab_catch_tile256_write_through.log)
template <class HotFn, int POLICY = BSX_ST_PLAIN>
__device__ __forceinline__ void bsx_tile_stream(float* __restrict__ tile, const int32_t* s_state, int lanes_here,
                                                uint32_t cells, uint32_t cells_magic, const HotFn& fn) {
  const uint32_t total = (uint32_t)lanes_here * cells;                  // <= 256 * 4096 floats
  const uint32_t n_chunks = total >> 2;
  const bool aligned = (cells & 3u) == 0;
  bsx_f4* __restrict__ t4 = reinterpret_cast<bsx_f4*>(tile);
  for (uint32_t c = threadIdx.x; c < n_chunks; c += BSX_BLOCK) {
    const uint32_t f = c << 2;
    const uint32_t dl = bsx_div_cells(f, cells, cells_magic);
    const int r0 = (int)(f - dl * cells);
    int ha, hb;
    fn(s_state[dl], ha, hb);
    const int a0 = ha < 0 ? -1 : ha - r0, b0 = hb < 0 ? -1 : hb - r0;
    bsx_f4 v;
    v.x = (a0 == 0 || b0 == 0) ? 1.0f : 0.0f;
    v.y = (a0 == 1 || b0 == 1) ? 1.0f : 0.0f;
    v.z = (a0 == 2 || b0 == 2) ? 1.0f : 0.0f;
    v.w = (a0 == 3 || b0 == 3) ? 1.0f : 0.0f;
    const int over = (int)cells - r0;
    if (!aligned && over < 4) {               // elements j >= over belong to the next lane's row (dl + 1 < lanes_here
      int na, nb;                             // because the chunk lies inside the tile)
      fn(s_state[dl + 1], na, nb);
      const int a1 = na < 0 ? -1 : na + over, b1 = nb < 0 ? -1 : nb + over;
      if (over <= 1) v.y = (a1 == 1 || b1 == 1) ? 1.0f : 0.0f;
      if (over <= 2) v.z = (a1 == 2 || b1 == 2) ? 1.0f : 0.0f;
      v.w = (a1 == 3 || b1 == 3) ? 1.0f : 0.0f;
    }
    bsx_st<POLICY>(&t4[c], v);
  }
  // ragged tail (< 4 floats): only the last, partial workgroup of an odd-sized array can have one
  const uint32_t f = (n_chunks << 2) + threadIdx.x;
  if (threadIdx.x < 3 && f < total) {
    const uint32_t dl = bsx_div_cells(f, cells, cells_magic);
    const int r = (int)(f - dl * cells);
    int ha, hb;
    fn(s_state[dl], ha, hb);
    tile[f] = (ha == r || hb == r) ? 1.0f : 0.0f;
  }
}

template <class Fam, bool LEAN, class HotFn>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_fused_tile_kernel(const typename Fam::args a, float* __restrict__ obs,
                                                                   const uint32_t cells, const uint32_t cells_magic,
                                                                   const HotFn fn) {
  __shared__ typename Fam::shared s_fam;
  __shared__ unsigned int s_cnt[2];
  __shared__ int32_t s_state[BSX_BLOCK];
  bsx_advance_body<Fam, LEAN>(a, blockIdx.x, s_fam, s_cnt, s_state);    // ends with a barrier: s_state is complete
  const int64_t lane0 = (int64_t)blockIdx.x * BSX_BLOCK;
  const int64_t left = a.ctl.n_lanes - lane0;
  bsx_tile_stream<HotFn, BSX_ST_WT>(obs + lane0 * (int64_t)cells, s_state, left < BSX_BLOCK ? (int)left : BSX_BLOCK, cells, cells_magic, fn);
}

// ... with 64-lane tiles: wave 0 advances the workgroup's 64 lanes, all four waves stream their [64 x cells] boards.  A
// rank's share of a strong-scaled batch (2^17 lanes of catch) is 512 workgroups of the 256-lane kernel — two per CU,
// every wave a chain of {loads, advance, barrier, 13 chunk stores}; 64-lane tiles make it 2048 workgroups (8 per CU)
// whose threads each write 3 chunks.
template <class Fam, bool LEAN, class HotFn>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_fused_tile64_kernel(const typename Fam::args a, float* __restrict__ obs,
                                                                     const uint32_t cells, const uint32_t cells_magic,
                                                                     const HotFn fn) {
  __shared__ typename Fam::shared s_fam;
  __shared__ int32_t s_state[BSX_WAVE];
  Fam::stage(a, s_fam);
  __syncthreads();
  const int64_t lane0 = (int64_t)blockIdx.x * BSX_WAVE;
  const int64_t i = lane0 + threadIdx.x;
  int type = -1;
  if (threadIdx.x < BSX_WAVE && i < a.ctl.n_lanes) {
    const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;
    const uint64_t step = bsx_step_of(a.ctl);
    int32_t nst; double reward;
    const int act = a.ctl.force_reset ? 0 : bsx_action(a.ctl, a.action, i, step);
    const int32_t st = a.ctl.state_in != nullptr ? a.ctl.state_in[i] : a.state[i];
    type = Fam::template advance<LEAN>(a, s_fam, i, lane, step, st, act, nst, reward);
    a.state[i] = nst;
    s_state[threadIdx.x] = nst;
    if (LEAN) bsx_emit_at<0, 0, false>(a.ctl, a.out, i, i, lane, step, type, reward);
    else bsx_emit(a.ctl, a.out, i, lane, step, type, reward);
  }
  if (threadIdx.x < BSX_WAVE && a.ctl.counters != nullptr) {               // one wave: its ballots ARE the workgroup's counts
    const unsigned long long last = __ballot(type == BSX_LAST), first = __ballot(type == BSX_FIRST);
    if (threadIdx.x == 0 && (last | first) != 0ull) {
      unsigned long long* shard = (unsigned long long*)a.ctl.counters + (size_t)(blockIdx.x & (BSX_COUNTER_SHARDS - 1)) * BSX_COUNTER_STRIDE;
      if (last) atomicAdd(&shard[0], (unsigned long long)__popcll(last));
      if (first) atomicAdd(&shard[1], (unsigned long long)__popcll(first));
    }
  }
  __syncthreads();
  const int64_t left = a.ctl.n_lanes - lane0;
  bsx_tile_stream<HotFn, BSX_TILE64_POLICY>(obs + lane0 * (int64_t)cells, s_state, left < BSX_WAVE ? (int)left : BSX_WAVE, cells, cells_magic, fn);
}

// The same for a rollout of T steps: ONE launch.  Lanes never interact, so a workgroup can take its 256 lanes
// through all T steps on its own — packed state in a register, actions prefetched one step ahead, per step one
// barrier (the LDS state tile is double-buffered) and the [256 x cells] tile of slice t streamed while the next
// step's advance is already under way in the faster waves.  No launch boundary, no state round trip.
template <class Fam, bool LEAN, class HotFn>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_fused_rollout_kernel(const typename Fam::args a, const int n_steps,
                                                                      float* __restrict__ obs, const uint32_t cells,
                                                                      const uint32_t cells_magic, const HotFn fn) {
  __shared__ typename Fam::shared s_fam;
  __shared__ unsigned int s_cnt[2];
  __shared__ int32_t s_state[2][BSX_BLOCK];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  Fam::stage(a, s_fam);
  __syncthreads();
  const int64_t B = a.ctl.n_lanes;
  const int64_t lane0 = (int64_t)blockIdx.x * BSX_BLOCK;
  const int64_t i = lane0 + threadIdx.x;
  const bool mine = i < B;
  const int lanes_here = B - lane0 < BSX_BLOCK ? (int)(B - lane0) : BSX_BLOCK;
  const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;
  const uint64_t step0 = bsx_step_of(a.ctl);
  int32_t st = mine ? a.state[i] : 0;
  int act_next = mine ? a.action[i] : 0;
#pragma unroll 1
  for (int t = 0; t < n_steps; ++t) {
    int type = -1;
    const int act = act_next;
    if (mine) {
      if (t + 1 < n_steps) act_next = a.action[(int64_t)(t + 1) * B + i];
      int32_t nst; double reward;
      type = Fam::template advance<LEAN>(a, s_fam, i, lane, step0 + (uint64_t)t, st, act, nst, reward);
      st = nst;
      s_state[t & 1][threadIdx.x] = nst;
      if (LEAN) bsx_emit_at<0, 0, false>(a.ctl, a.out, i, (int64_t)t * B + i, lane, step0 + (uint64_t)t, type, reward);
      else bsx_emit_at(a.ctl, a.out, i, (int64_t)t * B + i, lane, step0 + (uint64_t)t, type, reward);
    }
    bsx_count_types(a.ctl, type, s_cnt);
    // one barrier per step: tile t is read from s_state[t & 1] after it; step t+1 writes the other buffer, and no
    // thread reaches step t+2 (which rewrites this one) before every thread has passed the barrier of step t+1
    __syncthreads();
    bsx_tile_stream<HotFn, BSX_ST_NT>(obs + ((int64_t)t * B + lane0) * (int64_t)cells, s_state[t & 1], lanes_here, cells, cells_magic, fn);
  }
  if (mine) a.state[i] = st;
  bsx_final_barrier();
  bsx_flush_counts(a.ctl, s_cnt, blockIdx.x);
}

// Software-pipelined rollout step of a two-kernel family: ONE launch runs the observation stream of
// step t beside the lane advance of step t+1.  Nothing inside the launch depends on anything else inside
// it — both halves read the packed state column W(t) that the previous launch wrote, the advance writes
// the OTHER column (bsx_call_t.state_alt) — so, unlike a fused {advance, stream} of the same step
// (profiles/r02/ab_step1_fused_single_launch.log), no workgroup ever waits for another.  The first
// adv_blocks of the workgroups advance lanes (bsx_pipe_role_of); the rest are the store stream.
template <class Fam, bool LEAN, class HotFn, int K>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_pipelined_kernel(const typename Fam::args a, const uint32_t adv_blocks,
                                                                  const uint32_t place,
                                                                  float* __restrict__ obs, const int32_t* __restrict__ hot_state,
                                                                  uint32_t cells, uint32_t cells_magic, bsx_div64 dv, HotFn fn) {
  __shared__ typename Fam::shared s_fam;
  __shared__ unsigned int s_cnt[2];
  const bsx_pipe_role r = bsx_pipe_role_of(blockIdx.x, gridDim.x, adv_blocks, place);   // uniform per workgroup
  if (r.adv) bsx_advance_body<Fam, LEAN>(a, r.index, s_fam, s_cnt);
#if defined(BSX_AB_PIPELINED_NT)     // measurement builds only: non-temporal stores in the pipelined rollout's stream half
  else bsx_hot_stream_body<HotFn, K, BSX_BLOCK, true>(obs, hot_state, a.ctl.n_lanes, cells, cells_magic, dv, fn, r.index);
#else
  else bsx_hot_stream_body<HotFn, K, BSX_BLOCK>(obs, hot_state, a.ctl.n_lanes, cells, cells_magic, dv, fn, r.index);
#endif
}

template <class HotFn, int K>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_hot_stream_group_kernel(
    const bsx_stream_seg<HotFn>* __restrict__ table, const bsx_group_index gi) {
  const bsx_group_slot w = bsx_group_find(gi, (int)blockIdx.x);
  const bsx_stream_seg<HotFn>& g = table[w.seg];
  bsx_hot_stream_body<HotFn, K, BSX_BLOCK>(g.obs, g.state, g.n_lanes, g.cells, g.cells_magic, g.dv, g.fn, w.block);
}

// Delta observation mode (bsx_call_t.obs_paint): the observation array persists between calls and
// already shows the hot cells of the packed state recorded in `paint`; the thread that advances a
// lane also clears the cells that went stale and sets the new ones — at most 4 scattered 4-byte
// stores per lane instead of the whole board, in the same launch as the advance.  The array
// afterwards is bit-identical to the dense mode's.
template <class HotFn>
__device__ __forceinline__ void bsx_patch_board(float* __restrict__ board, int32_t was, int32_t now, const HotFn& fn) {
  if (now == was) return;
  int a0 = -1, b0 = -1, a1, b1;
  if (was != -1) fn(was, a0, b0);
  fn(now, a1, b1);
  if (a0 >= 0 && a0 != a1 && a0 != b1) board[a0] = 0.0f;
  if (b0 >= 0 && b0 != a1 && b0 != b1) board[b0] = 0.0f;
  if (a1 >= 0 && a1 != a0 && a1 != b0) board[a1] = 1.0f;
  if (b1 >= 0 && b1 != a0 && b1 != b0) board[b1] = 1.0f;
}

template <class Fam, class HotFn>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_advance_delta_kernel(const typename Fam::args a, const HotFn fn,
                                                                      int32_t* __restrict__ paint, const uint32_t cells) {
  __shared__ typename Fam::shared s_fam;
  __shared__ unsigned int s_cnt[2];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  Fam::stage(a, s_fam);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * BSX_BLOCK + threadIdx.x;
  int type = -1;
  if (i < a.ctl.n_lanes) {
    const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;
    const uint64_t step = bsx_step_of(a.ctl);
    int32_t nst; double reward;
    const int act = a.ctl.force_reset ? 0 : bsx_action(a.ctl, a.action, i, step);
    const int32_t was = paint[i];
    type = Fam::template advance<false>(a, s_fam, i, lane, step, a.state[i], act, nst, reward);
    a.state[i] = nst;
    bsx_patch_board(a.out.observation + i * (int64_t)cells, was, nst, fn);
    paint[i] = nst;
    bsx_emit(a.ctl, a.out, i, lane, step, type, reward);
  }
  bsx_count_types(a.ctl, type, s_cnt);
  bsx_final_barrier();
  bsx_flush_counts(a.ctl, s_cnt, blockIdx.x);
}

template <class Fam, class HotFn>
static inline int bsx_launch_advance_delta(const typename Fam::args& a, const HotFn& fn, int32_t* paint,
                                           uint32_t cells, hipStream_t st) {
  const int64_t blocks = (a.ctl.n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  if (blocks > 0x7FFFFFFF) return BSX_EINVAL;
  bsx_advance_delta_kernel<Fam, HotFn><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(a, fn, paint, cells);
  return 0;
}

// Degenerate boards (cells < 4: a 16-byte chunk spans several lanes): one float per thread.
template <class HotFn>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_hot_stream_tiny_kernel(float* __restrict__ obs,
                                                                        const int32_t* __restrict__ state,
                                                                        int64_t n_lanes, uint32_t cells,
                                                                        HotFn fn) {
  const uint64_t F = (uint64_t)blockIdx.x * BSX_BLOCK + threadIdx.x;
  if (F >= (uint64_t)n_lanes * cells) return;
  const uint64_t lane = F / cells;
  const int r = (int)(F - lane * cells);
  int ha, hb;
  fn(state[lane], ha, hb);
  obs[F] = (ha == r || hb == r) ? 1.0f : 0.0f;
}

#endif  // BSX_DEVICE_H_
