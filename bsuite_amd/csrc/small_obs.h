// small_obs.h — device code of the families whose observation is a short row (1..256 floats per lane):
//   bandit            bsuite/environments/bandit.py:54-64
//   memory_chain      bsuite/environments/memory_chain.py:60-97
//   umbrella_chain    bsuite/environments/umbrella_chain.py:60-92
//   discounting_chain bsuite/environments/discounting_chain.py:63-88
//   cartpole/swingup  bsuite/environments/cartpole.py:37-177,
//                     bsuite/experiments/cartpole_swingup/cartpole_swingup.py:81-150
//   mountain_car      bsuite/environments/mountain_car.py:62-90
// each with the auto-reset of bsuite/environments/base.py:54-65.
//
// One kernel shape for all of them: thread t advances lane (block*256 + t) from coalesced SoA column
// loads and either stores its short observation row itself or — the wide rows of memory_chain /
// umbrella_chain — leaves it as bits in an LDS tile that the whole block streams to HBM as consecutive
// 16-byte chunks (small_obs_body, below).  Physics state is f32 on the device (the reference holds
// Python floats); rewards and the time-fraction observation are formed in f64 exactly as the reference
// forms them and cast once.
#ifndef BSX_SMALL_OBS_H_
#define BSX_SMALL_OBS_H_

#include <type_traits>

#include "bsx_host.h"
#include "bsx_math.h"
#include "catch_fam.h"
#include "deep_sea_fam.h"
#include "mnist_fam.h"
#include "pair_mixed.h"
#include "row_stream.h"

// n_steps == 1 is env.step()/reset(); n_steps = T > 1 is the fused rollout: the same thread advances
// its lane T times inside one launch (actions [T,B], outputs [T,B,...]); per-lane state columns are
// re-read from L2 by the thread that wrote them, so HBM sees only the action/TimeStep streams —
// the tiny families are otherwise bound by one ~8 us launch per step.
// LOG / NOISE / MT: -1 = decide at run time, 0 = compiled out, 1 = always on.  The common call (no
// Logging wrapper, no RewardNoise, counter-based draws) runs the <0,0,0> instantiation: without the
// MT19937 twist, the f64 normal transform and the row snapshots the kernel is a fifth of the size.
//
// Two ways for a row to reach HBM:
//   DIRECT  rows of 1, 3 or an even number <= 8 floats: the thread that advances a lane stores its row
//           itself — no LDS, no barrier.  Every family with a fixed short row (bandit, discounting_chain,
//           cartpole, mountain_car) always takes it.
//   PACKED  the families whose row length is a parameter (memory_chain: nb+2, umbrella_chain: 3+nd, up to
//           256 floats) and whose row is a few floats (HEAD) followed by values that one or two BITS
//           encode.  The block keeps the tile [256 x numel] as flat bit planes in LDS — bit (l*numel + j) of
//           a plane belongs to element j of lane l — which each lane ORs its bits into; after a barrier
//           the block streams the tile to HBM as consecutive 16-byte chunks: chunk c is the nibble at bit
//           4c of each plane, four v_bfe/v_cvt away from a float4 (a lane-per-row store would be a
//           stride-(4*numel) scatter; an f32 tile costs 32x the LDS — up to 32 KiB per workgroup, which
//           capped the resident workgroups per CU — and per-element records cost ~80 VALU instructions
//           per chunk: profiles/r02/ab_packed_records_v*.log).  The HEAD floats stay in the lane's
//           registers and overwrite their (zero) places in the tile after a second barrier.
//           LDS per workgroup: 32*numel bytes per plane (<= 8 KiB).
//   ROWS    the same wide rows when the call brings a scratch (bsx_call_t.row_scratch, ABI v12) and is a single step:
//           every WAVE builds its 64 lanes' flat bit planes in wave-private LDS — no workgroup barrier — and stores
//           them into the scratch; a store stream decodes them into the observation array in a second launch
//           (row_stream.h).
// A lane's handle on the tile's bit planes.
struct bsx_bit_sink {
  uint32_t* planes;        // LDS: PLANES x `stride` words
  int stride;              // words per plane = 8 * numel
  uint32_t base;           // flat bit index of the lane's element HEAD
  const float* tf = nullptr;  // LDS: the family's time fractions 1 - t / L for t = 0..L (small_obs_body, PACKED), or null
  // ORs bits [32k, 32k+n) of the lane's bit string (n in 1..32, the low n bits of w) into plane p
  __device__ __forceinline__ void put(int p, int k, uint32_t w, int n) const {
    uint32_t word, lo, hi;
    int has_hi;
    bsx_plane_split(base + 32u * (uint32_t)k, w, n, &word, &lo, &hi, &has_hi);
    uint32_t* dst = planes + p * stride + word;
    atomicOr(dst, lo);
    if (has_hi) atomicOr(dst + 1, hi);
  }
};

// A table staged in LDS, typed as such: a generic pointer would make every read a FLAT load, which counts in vmcnt
// AND lgkmcnt and drags a full `s_waitcnt vmcnt(0)` (a drain of the wave's stores) in front of its first use.
typedef const float __attribute__((address_space(3)))* bsx_lds_table;

// Reset values of a workgroup's lanes, computed by a few threads on behalf of their owners (small_obs_regs_rollout).
struct bsx_reset_pool {
  float vals[6][BSX_BLOCK];          // [value][owner thread]
  unsigned short list[BSX_BLOCK];    // the threads whose lane begins an episode at this step, in arrival order
  unsigned int n[2];                 // how many; [step parity]
};

// The cache policy of the OUTPUT stores (bsx_st<POLICY>, BSX_SMALL_NT: bsx_device.h; round 6).  What a step or a rollout writes —
// TimeStep columns and observation rows — is never read again by the engine, while what it READS (actions a run ahead, state
// and info columns, tables) is what the stores evict: the fused rollouts were bound by exactly those reads (r04: 6.5 us with,
// 4.8 without the action loads).  A FUSED ROLLOUT's outputs are NON-TEMPORAL wherever a wave's store instruction covers one
// contiguous range: mountain_car r16 6.45 -> 4.45-4.7 us per step, memory_len r16 7.0 -> 5.2, discounting_chain r16 5.2 ->
// 3.8, bandit r16 5.0 -> 4.0, cartpole r16 9.8 -> 7.7 (its rows leave as 16-byte chunks through the wave's LDS), umbrella_length
// r16 23.8 -> 21.4 (profiles/r06/ab_small_families_nt*.log, ab_nt_wide_rows_and_small_batches.log).  An EAGER step's outputs are
// WRITE-THROUGH: non-temporal ones make the step itself faster still (umbrella_length 25.8 -> 21.7 us, mountain_car 8.2 -> 7.3)
// but cost the agent that reads the observation next more than that (ab_nt_outputs_closed_loop_policy.log); write-through:
// cartpole 17.7 -> 15.8 us, discounting_chain 7.05 -> 6.05, memory_len 10.0 -> 9.25, bandit 8.46 -> 7.96, the closed loop equal or
// faster everywhere (ab_eager_output_policy.log).  NEITHER for rows written as 8-byte pieces at the row stride (cartpole
// row-per-lane: 18 -> 25 us non-temporal — partial lines want the L2 to merge them).  deep_sea / catch / mnist / the sweep are
// not touched by this (ab_small_nt_other_paths.log).  Per family: does the fused rollout store reward / discount / step_type
// non-temporal?
template <class Env> struct small_rollout_nt_scalars { static constexpr bool value = true; };

// A lane's own thread stores its short row (<= 8 floats).  A wave's 64 rows are one contiguous range, written by
// back-to-back instructions that the L2 merges line by line.
template <bool ROLLOUT_ST>
__device__ __forceinline__ void small_obs_store_row(float* __restrict__ dst, const float* o, int numel) {
  constexpr int P2 = ROLLOUT_ST ? BSX_POLICY_R(2) : BSX_POLICY_E(2), P8 = ROLLOUT_ST ? BSX_POLICY_R(8) : BSX_POLICY_E(8);
  if ((numel & 1) == 0) {
    bsx_f2* __restrict__ d2 = reinterpret_cast<bsx_f2*>(dst);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (2 * k < numel) {
        bsx_f2 v; v.x = o[2 * k]; v.y = o[2 * k + 1];
        if (numel == 2) bsx_st<P2>(&d2[k], v);
        else bsx_st<P8>(&d2[k], v);
      }
  } else if (numel == 3) {
    // one 12-byte store per lane (global_store_dwordx3): a wave's 64 rows are 768 contiguous bytes
    if (ROLLOUT_ST && (BSX_SMALL_NT & 16)) {
      typedef float row3v __attribute__((ext_vector_type(3), aligned(4)));
      row3v v; v.x = o[0]; v.y = o[1]; v.z = o[2];
      __builtin_nontemporal_store(v, reinterpret_cast<row3v*>(dst));
    } else {
      struct __attribute__((packed, aligned(4))) row3 { float a, b, c; };
      row3 v; v.a = o[0]; v.b = o[1]; v.c = o[2];
      bsx_st<(ROLLOUT_ST ? BSX_ST_PLAIN : BSX_POLICY_E(16))>(reinterpret_cast<row3*>(dst), v);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 7; ++k)                                   // numel == 1, 5, 7: 4-byte stores
      if (k < numel) {
        if (numel == 1) bsx_st<P2>(&dst[k], o[k]);
        else dst[k] = o[k];
      }
  }
}

// An element `off` BYTES (32 bits, per lane) behind a wave-uniform base: the address add is the memory instruction's
// (global_store ... v_off, v_data, s[base:base+1]), not three or four 64-bit VALU instructions per access.
// The pointer is typed as GLOBAL memory (address_space(1)): the slab pointers pass through an opaque asm statement
// every step (see the loop), after which the compiler no longer knows where a generic pointer leads and would emit
// FLAT accesses — which count in vmcnt AND lgkmcnt and have no scalar-base form.
#if defined(__HIP_DEVICE_COMPILE__)
#define BSX_GLOBAL __attribute__((address_space(1)))
#else
#define BSX_GLOBAL              /* (the host pass only parses the device functions) */
#endif
template <class T>
__device__ __forceinline__ BSX_GLOBAL T* bsx_at_off(T* base, uint32_t off) {
  return (BSX_GLOBAL T*)((BSX_GLOBAL char*)base + off);
}
template <class T>
__device__ __forceinline__ const BSX_GLOBAL T* bsx_at_off(const T* base, uint32_t off) {
  return (const BSX_GLOBAL T*)((const BSX_GLOBAL char*)base + off);
}

// The 64 rows of a full wave, through a wave-private LDS staging area, as 16-byte chunks: row-per-lane stores of a
// 24-byte row are 8-byte pieces at stride 24 — every store instruction touches all of the wave's 64-byte segments with
// a third of their bytes, three write requests per segment where one would do, and the fused rollouts of the physics
// families turned out to be bound by exactly that (profiles/r03/exp_store_ablation.log: without the row stores
// cartpole's step takes 8.1 us instead of 13.3).  No workgroup barrier: LDS serves a wave's accesses in order.
// dst = row of the wave's first lane (16-byte aligned: the caller checks), s_wave = 64 * 8 floats, wl = lane in the wave.
template <bool ROLLOUT_ST>
__device__ __forceinline__ void small_obs_store_rows_wave(float* __restrict__ dst, const float* o, int numel, float* s_wave, int wl) {
  float* mine = s_wave + wl * numel;
  if ((numel & 1) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (2 * k < numel) *reinterpret_cast<float2*>(mine + 2 * k) = make_float2(o[2 * k], o[2 * k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (k < numel) mine[k] = o[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int chunks = 16 * numel;                                 // 64 rows x numel floats / 4
  for (int c = wl; c < chunks; c += 64)
    bsx_st<(ROLLOUT_ST ? BSX_POLICY_R(4) : BSX_POLICY_E(4))>(&reinterpret_cast<bsx_f4*>(dst)[c], reinterpret_cast<const bsx_f4*>(s_wave)[c]);
  __builtin_amdgcn_wave_barrier();                               // (the next step's rows are written after these reads)
}
// ... the same with the destination as {uniform slab pointer, byte offset of the wave's first row}
__device__ __forceinline__ void small_obs_store_rows_wave_off(float* slab, uint32_t wave_off, const float* o, int numel, float* s_wave, int wl) {
  float* mine = s_wave + wl * numel;
  if ((numel & 1) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (2 * k < numel) *reinterpret_cast<float2*>(mine + 2 * k) = make_float2(o[2 * k], o[2 * k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (k < numel) mine[k] = o[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int chunks = 16 * numel;
  for (int c = wl; c < chunks; c += 64)
    bsx_st<BSX_POLICY_R(4)>(bsx_at_off(reinterpret_cast<bsx_f4*>(slab), wave_off + 16u * (uint32_t)c), reinterpret_cast<const bsx_f4*>(s_wave)[c]);
  __builtin_amdgcn_wave_barrier();
}

// The same value, opaque to the optimiser: what is computed from it stays where it is written (no hoisting out of loops).
__device__ __forceinline__ uint32_t bsx_fresh(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Fused T-step rollout of a family whose lane state fits in registers (cartpole, swing-up, mountain_car): ONE launch,
// and nothing inside the step loop ever waits for memory —
//  * the state AND the bsuite_info accumulators of the lane live in registers for the T steps (without the Logging
//    wrapper, whose rows snapshot the columns): the episode-end read-modify-writes were `s_waitcnt vmcnt(0)` in the
//    loop, i.e. a drain of every store the wave had in flight, on 6 steps in 10 (some lane of 64 ends an episode);
//  * actions are loaded RUN steps at a time, ahead of the run: on gfx9 stores count in vmcnt too, so waiting for a
//    one-step-ahead prefetch issued before a divergent number of stores also meant vmcnt(0) — once per step;
//  * cartpole's time-fraction table sits in LDS (lgkmcnt) when it fits.
// r02 (prefetch + in-loop RMW): 70 % of the wave cycles were waits, cartpole 15.1 us per step at 53 % VALU-busy
// (profiles/r03/cartpole_rollout16_before_pmc_sq.json).
// V: the family's variant as a compile-time constant (Env::numel_of(V) floats per row, cartpole: 0 classic / 1 swing-up),
// or -1 = read obs_numel / the flags from the arguments.  With the row length a run-time value every `2*k < numel` of
// the row store became a loop-invariant 64-bit condition mask of its own: 119 SGPR spills and ~90 v_readlane reloads
// in each copy of the step loop (r03 build; profiles/r04/cartpole_rollout_loop_isa_before.txt).
// V >= 0 also promises (launch_regs_rollout) that a [B, numel] slab is shorter than 4 GiB: every output of step t is
// then addressed as {uniform slab pointer, advanced once per step on the scalar unit} + {the lane's 32-bit byte offset,
// loop-invariant}, and no 64-bit index t*B + i is formed per step and array.
template <class Env, int LOG, int NOISE, int MT, bool TAB, bool BIG, int V = -1>
__device__ __forceinline__ void small_obs_regs_rollout(const typename Env::args& a, const int n_steps, const uint32_t block_id,
                                                       float* s_dyn, unsigned int* s_cnt, bsx_reset_pool* s_pool, float* s_rows) {
  constexpr bool IREGS = LOG == 0;
  // BIG: the launch fills the chip several times over (launch_small_obs) — the loop is then bound by throughput (vector
  // instructions, write requests) and pools the resets / stages the rows; a small launch is bound by the latency of a
  // wave's own step, to which the pool's two barriers and the LDS round trip only add (2^17 lanes: cartpole 2.1 -> 2.5 us
  // per step, profiles/r03/ab_rows_small_batches.log).
  constexpr bool POOL = BIG && Env::POOLED_RESETS && MT == 0;
  constexpr bool ROWS = BIG && Env::ROWS_VIA_LDS;
  constexpr bool OFF32 = V >= 0;
#ifndef BSX_RUN
#define BSX_RUN 8
#endif
  constexpr int RUN = OFF32 ? BSX_RUN : 8;                // (16: no drain inside a T=16 launch at all, and 3-7 % slower — profiles/r03/ab_rollout_run16.log)
  if (threadIdx.x < 2) {
    s_cnt[threadIdx.x] = 0;
    if constexpr (POOL) s_pool->n[threadIdx.x] = 0;
  }
  // TAB: the family's table is staged in LDS (a compile-time fact inside the loop: were the source of a value
  // decided at run time, the compiler would wait for the global load it MIGHT have been — vmcnt — at every use)
  bsx_lds_table s_tab = (bsx_lds_table)0;
  if constexpr (TAB) s_tab = Env::stage_tables(a, s_dyn);
  __syncthreads();
  const int numel = V >= 0 ? Env::numel_of(V) : a.obs_numel;
  constexpr bool F64 = !(LOG == 0 && NOISE == 0 && MT == 0);       // bsx_ctl_lean: no f64 reward copy
  const int64_t B = a.ctl.n_lanes;
  const int64_t i = (int64_t)block_id * BSX_BLOCK + threadIdx.x;
  const uint32_t iu0 = (uint32_t)i;
  // (Tried: BIG variants for whole workgroups only, i.e. no `mine` mask in the loop — fewer scalar registers, but the
  // scheduler then interleaves across the former block boundary: 65 -> 71 VGPRs for cartpole, 68 -> 85 for swing-up.)
  const bool mine = i < B;
  const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;
  const uint64_t step0 = bsx_step_of(a.ctl);
  // rows leave through the wave's LDS staging area when every row of the wave exists and every step's block of rows
  // starts on a 16-byte boundary (wave-uniform)
  const int wl = (int)(threadIdx.x & 63u);
  const bool rows_via_lds = ROWS && (((int64_t)B * numel) & 3) == 0 && (i - wl + 64) <= B;
  // LAST lanes are counted per thread and pooled once per launch (the per-step wave ballots + LDS atomics of
  // bsx_count_types were ~12 instructions and two LDS round trips of every wave-step); FIRST steps follow from them:
  // every LAST is followed by a FIRST except one at the launch's final step, and a lane that arrives with its reset
  // pending begins with one (no force_reset inside a rollout).
  uint32_t n_last = 0;
  typename Env::regs rg;
  Env::clear(rg);
  if (mine) {
    Env::load(a, i, rg);
    if constexpr (IREGS) Env::template load_info<V>(a, i, rg);
  }
  const uint32_t pending_in = Env::reset_pending(rg) ? 1u : 0u;
  // the slabs of the step under way (OFF32): [B] reward / discount / step_type, [B, numel] observation, [B] action
  float* rp = a.out.reward;
  float* dp = a.out.discount;
  int8_t* sp = a.out.step_type;
  float* op = a.out.observation;
  const int32_t* ap = a.action;
  // (Tried, profiles/r04/ab_rollout_action_prepack.log: a pre-pass that leaves action[t, i] as a BYTE in step_type[t, i] —
  // the slot this loop overwrites after reading it — so that the loop reads 1 byte per lane-step from an array one
  // read-dominated pass has just produced.  The loop does not get faster by what the pre-pass costs: cartpole 9.9-10.2 ->
  // 10.9-11.0 us per step, mountain_car 5.9-6.5 -> 5.8-6.1 at 2^20 lanes but 2.64 -> 2.93 at 2^19 and 5.5 -> 6.2 at
  // T = 64.  Not adopted.
  // Nor did removing the per-run drain: the next run's actions loaded by inline asm at the START of the current run
  // (invisible to the compiler's wait insertion) and awaited at its end with `s_waitcnt vmcnt(8 x stores per step)`,
  // which the loads alone satisfy — correct (rollout tests), mountain_car r16 6.5-6.65 -> 6.25-6.4 us, nothing at
  // T = 32 / 64, cartpole +-1 % at the price of its eighth wave (profiles/r04/ab_rollout_prefetched_runs.log).  The loop
  // is 26-36 % faster without its action loads (exp_action_load_ablation.log) — and so is the bare access pattern
  // (tools/micro/step_stores.hip, same box: 3.95 us without loads, 5.04 in runs of eight, 4.64 with all rows loaded
  // before the first store = the loads' own 0.69 us added; prefetched a run ahead 5.70): reads among a saturating
  // stream of per-lane stores cost at least their own time, in any arrangement.)
#pragma unroll 1
  for (int t0 = 0; t0 < n_steps; t0 += RUN) {
    const int run = n_steps - t0 < RUN ? n_steps - t0 : RUN;             // uniform
    // The run's actions: eight registers for eight 2-bit values were what stood between the pooled cartpole loop and
    // 64 VGPRs (8 waves per SIMD), and picking act[j] cost 7 selects per step.  OFF32 packs them, 4 bits each, into
    // ONE register (unpacked by one v_bfe_u32 with a scalar shift).  Any int32 is a legal action of these families
    // (the reference computes (action - 1) * force with whatever it gets, cartpole.py:48): a run in which some lane
    // of the wave holds an action outside 0..15 re-reads its actions from memory step by step instead (wave-uniform
    // branch; the wait it needs drains the wave's stores — slow, exact, and never taken by in-spec actions).
    int acts[OFF32 ? 1 : RUN];
    typename std::conditional<(RUN > 8), uint64_t, uint32_t>::type packed = 0;
    bool wide = false;
    const int32_t* const ap_run = ap;
    if constexpr (OFF32) {
      if (mine) {
        // (rows beyond the run re-read its last row: no per-row condition, no table of row strides in scalar registers)
        const uint32_t off = bsx_fresh(iu0) * 4u;
#pragma unroll
        for (int j = 0; j < RUN; ++j) {
#if defined(BSX_ABLATE_STORES) && (BSX_ABLATE_STORES & 4)
          const uint32_t aj = (iu0 + (uint32_t)(t0 + j)) % 3u;           // measurement builds: the loop without its action loads
#elif defined(BSX_ABLATE_STORES) && (BSX_ABLATE_STORES & 8)
          uint32_t aj = (uint32_t)*bsx_at_off(ap, off);                  // ... with the loads, but the same periodic actions
          if (aj != 0x7FFFFFF0u) aj = (iu0 + (uint32_t)(t0 + j)) % 3u;
#else
          const uint32_t aj = (uint32_t)*bsx_at_off(ap, off);
#endif
          packed |= (decltype(packed))(aj & 15u) << (4 * j);
          wide |= aj > 15u;
          if (j + 1 < run) ap += B;                                      // uniform
        }
        ap += B;
      }
    } else {
#pragma unroll
      for (int j = 0; j < RUN; ++j) acts[j] = (mine && j < run) ? a.action[(int64_t)(t0 + j) * B + i] : 0;
    }
    // (one scalar pair says both whether and where: the start of the run's actions, or null)
    const int32_t* const ap_wide = (OFF32 && __ballot(wide) != 0ull) ? ap_run : nullptr;     // uniform
    // Everything loaded so far has landed before the run starts (vmcnt(0) lgkmcnt(0)): the compiler's wait insertion
    // then knows that no register is waiting for memory inside the run — otherwise every first use of a
    // conditionally loaded value (the swing-up info columns) gets its own vmcnt(0), and on gfx9 that is a drain of
    // the STORES in flight as well.  One drain per RUN steps instead of several per step.
    __builtin_amdgcn_s_waitcnt(0x0070);
    // (a ROLLED loop: unrolled, the scheduler interleaves the steps and the kernel needs 140 VGPRs instead of ~75 —
    // half the waves per SIMD, no gain, profiles/r03/ab_regs_rollout.log; the run's action is picked by selects)
#pragma unroll 1
    for (int j = 0; j < run; ++j) {
      const int t = t0 + j;
      int type = -1;
      int act;
      if constexpr (OFF32) {
        act = (int)((uint32_t)(packed >> (4 * j)) & 15u);
        if (ap_wide != nullptr) {
          act = mine ? *bsx_at_off(ap_wide + (int64_t)j * B, bsx_fresh(iu0) * 4u) : 0;
          __builtin_amdgcn_s_waitcnt(0x0070);                            // landed: nothing pends beyond this block
        }
      } else {
        act = acts[0];
#pragma unroll
        for (int q = 1; q < RUN; ++q) act = j == q ? acts[q] : act;
      }
      if constexpr (POOL) {
        // The lanes that begin an episode at this step hand the draws of their reset to a pool: some lane of a
        // wave does on 4 steps in 10 (cartpole, random actions), and then the whole wave walks through two Philox
        // blocks and four f64 uniforms for its sake — ~230 vector instructions beside the step's ~150.  Pooled, ONE
        // wave of the workgroup computes all of them in one pass, two threads per lane (a Philox block each), and
        // the waves take turns at it step by step so that the work lands on every SIMD: 306 -> ~215 vector
        // instructions per wave and step, 77 -> 63 VGPRs (profiles/r03/ab_pooled_resets.log).
        const int par = t & 1;
        if (mine && Env::template wants_reset<true>(a, rg)) {
          const unsigned slot = atomicAdd(&s_pool->n[par], 1u);
          s_pool->list[slot] = (unsigned short)threadIdx.x;
        }
        __syncthreads();
        const unsigned n2 = 2u * s_pool->n[par];
        if (threadIdx.x == 0) s_pool->n[par ^ 1] = 0;
        for (unsigned h = (threadIdx.x + 64u * ((unsigned)t + block_id)) & (BSX_BLOCK - 1u); h < n2; h += BSX_BLOCK) {
          const unsigned e = s_pool->list[h >> 1];
          Env::reset_part(a, a.ctl.lane_offset + (uint64_t)block_id * BSX_BLOCK + e, step0 + (uint64_t)t, (int)(h & 1u), e, s_pool);
        }
        __syncthreads();
      }
      if (mine) {
        // (the lane's byte offsets are formed HERE, from a value the compiler cannot trace to the loop's outside: hoisted
        // out of the loop they are four more live registers, and — zero-extended in another basic block — they no longer
        // match the {scalar base + 32-bit vector offset} addressing mode, so every access paid a 64-bit add again)
        const uint32_t iu = bsx_fresh(iu0);
        const int64_t oi = (int64_t)t * B + i;                           // (dead in the lean instantiations)
        double reward = 0.0;
        float o[8];
        type = Env::template core<LOG, MT, IREGS, TAB, POOL, V, true>(a, rg, act, i, lane, step0 + (uint64_t)t, o, reward, s_tab, s_pool);
        // Measurement builds only (-DBSX_ABLATE_STORES, tools/ablate_stores.sh; never the product library): the loop
        // without its observation rows (bit 0), without reward / discount / step_type (bit 1), without action loads (bit 2) — the conditions are
        // never true, the stores stay reachable so that the arithmetic feeding them is not compiled away.
#if defined(BSX_ABLATE_STORES)
        const bool scalars = !(BSX_ABLATE_STORES & 2) || (reward == 123.0 && type == 7);
        const bool rows = !(BSX_ABLATE_STORES & 1) || (o[0] == 123.0f && o[1] == 5.0f && o[2] == 7.0f);
#else
        constexpr bool scalars = true, rows = true;
#endif
        if (scalars) {
          if constexpr (OFF32) {
            float r, d;
            bsx_emit_values<LOG, NOISE, F64, MT>(a.ctl, i, oi, lane, step0 + (uint64_t)t, type, reward, r, d);
            constexpr int NTS = small_rollout_nt_scalars<Env>::value ? BSX_POLICY_R(1) : BSX_ST_PLAIN;
            bsx_st<NTS>(bsx_at_off(rp, iu * 4u), r);
            bsx_st<NTS>(bsx_at_off(dp, iu * 4u), d);
            bsx_st<NTS>(bsx_at_off(sp, iu), (int8_t)type);
          } else {
            bsx_emit_at<LOG, NOISE, F64, MT, BSX_POLICY_R(1)>(a.ctl, a.out, i, oi, lane, step0 + (uint64_t)t, type, reward);
          }
        }
        if (rows) {
          if constexpr (OFF32) {
            if (rows_via_lds) small_obs_store_rows_wave_off(op, (iu - (uint32_t)wl) * (uint32_t)(numel * 4), o, numel, s_rows + (threadIdx.x - wl) * 8, wl);
            else small_obs_store_row<true>(bsx_at_off(op, iu * (uint32_t)(numel * 4)), o, numel);
          } else {
            if (rows_via_lds) small_obs_store_rows_wave<true>(a.out.observation + (oi - wl) * (int64_t)numel, o, numel, s_rows + (threadIdx.x - wl) * 8, wl);
            else small_obs_store_row<true>(a.out.observation + oi * (int64_t)numel, o, numel);
          }
        }
      }
      if constexpr (OFF32) {
        // (a slab is shorter than 4 GiB: 32-bit strides — three scalar registers instead of three pairs)
        const uint32_t b1 = (uint32_t)B;
        rp = (float*)((char*)rp + b1 * 4u); dp = (float*)((char*)dp + b1 * 4u); sp += b1;
        op = (float*)((char*)op + b1 * (uint32_t)(numel * 4));
      } else {
        rp += B; dp += B; sp += B; op += B * (int64_t)numel;              // uniform: the scalar unit's
      }
      if constexpr (OFF32) {
        // (opaque, or loop strength reduction folds the four pointers into ONE running offset t*B*4 and every address
        // becomes base + (offset + lane): not the {scalar base, vector offset} form any more)
        asm volatile("" : "+s"(rp), "+s"(dp), "+s"(sp), "+s"(op));
      }
      n_last += (type == BSX_LAST) ? 1u : 0u;
    }
  }
  const uint32_t n_first = n_last + pending_in - (Env::reset_pending(rg) ? 1u : 0u);
  if (mine) {
    Env::store(a, i, rg);
    if constexpr (IREGS) Env::template store_info<V>(a, i, rg);
  }
  if (a.ctl.counters != nullptr) {
    if (n_last) atomicAdd(&s_cnt[0], n_last);
    if (n_first) atomicAdd(&s_cnt[1], n_first);
  }
  bsx_final_barrier();
  bsx_flush_counts(a.ctl, s_cnt, block_id);
}

// TABSEL: the register-resident families' table (cartpole's time fractions) is staged in LDS (1) or read from device
// memory (0) — the launcher knows; -1 = both loops in the kernel, chosen per launch by table_fits().
template <class Env, bool ROLLOUT, int LOG, int NOISE, int MT, bool DIRECT_ARG, bool BIG = false, int V = -1, int TABSEL = -1,
          bool ROWS_ARG = false>
__device__ __forceinline__ void small_obs_body(const typename Env::args& a, const int n_steps_arg,
                                               const uint32_t block_id, float* s_obs, unsigned int* s_cnt) {
  constexpr bool ROWS = ROWS_ARG && Env::PACKED && !ROLLOUT;         // wide rows, packed, into the call's row scratch
  constexpr bool DIRECT = DIRECT_ARG || !Env::PACKED;
  // (a family whose row length is a parameter — memory_chain — is register-resident in the rows its own thread stores)
  if constexpr (ROLLOUT && Env::HAS_REGS && (DIRECT_ARG || !Env::PACKED)) {
    bsx_reset_pool* pool = nullptr;
    float* rows = nullptr;
    if constexpr (BIG && Env::POOLED_RESETS && MT == 0) {
      __shared__ bsx_reset_pool s_pool;
      pool = &s_pool;
    }
    if constexpr (BIG && Env::ROWS_VIA_LDS) {
      __shared__ __attribute__((aligned(16))) float s_rows[BSX_BLOCK * 8];         // 64 rows of <= 8 floats per wave
      rows = s_rows;
    }
    if constexpr (TABSEL == 1) small_obs_regs_rollout<Env, LOG, NOISE, MT, true, BIG, V>(a, n_steps_arg, block_id, s_obs, s_cnt, pool, rows);
    else if constexpr (TABSEL == 0) small_obs_regs_rollout<Env, LOG, NOISE, MT, false, BIG, V>(a, n_steps_arg, block_id, s_obs, s_cnt, pool, rows);
    else if (Env::table_fits(a)) small_obs_regs_rollout<Env, LOG, NOISE, MT, true, BIG, V>(a, n_steps_arg, block_id, s_obs, s_cnt, pool, rows);   // uniform
    else small_obs_regs_rollout<Env, LOG, NOISE, MT, false, BIG, V>(a, n_steps_arg, block_id, s_obs, s_cnt, pool, rows);
    return;
  }
  constexpr bool F64 = !(LOG == 0 && NOISE == 0 && MT == 0);       // bsx_ctl_lean: no f64 reward copy
  const int n_steps = ROLLOUT ? n_steps_arg : 1;   // the single-step instantiation has no loop: keeping
                                                    // every kernarg live across iterations costs ~120 VGPRs
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int numel = a.obs_numel;
  const int64_t B = a.ctl.n_lanes;
  const int64_t lane0 = (int64_t)block_id * BSX_BLOCK;
  const int64_t remaining = B - lane0;
  const int lanes_here = remaining < BSX_BLOCK ? (int)remaining : BSX_BLOCK;
  const uint64_t step0 = bsx_step_of(a.ctl);
  const bool mine = (int)threadIdx.x < lanes_here;
  const int64_t i = lane0 + threadIdx.x;
  const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;

#pragma unroll 1
  for (int t = 0; t < n_steps; ++t) {
    const int64_t oi = (int64_t)t * B + i;
    int type = -1;
    if constexpr (ROWS) {
      // Wide rows into the call's row scratch (bsx_rows.h) for the store stream of the next launch (row_stream.h): the
      // advance of the PACKED path below without its workgroup barriers.  Every WAVE keeps its own flat bit planes in
      // LDS — 64 lanes x numel bits = 2 * numel whole words per plane, LDS serves a wave's accesses in order — which
      // its lanes OR their bits into and which it then stores as consecutive words of the global planes; the row
      // elements that are genuine floats go to one f32 column each, the 0.0 / 1.0 HEAD elements of umbrella_chain (need,
      // has) are plane bits like the distractors.
      typedef typename Env::rows_t R;
      const int wl = (int)(threadIdx.x & 63u);
      const uint32_t wstride = 2u * (uint32_t)numel;                       // words per plane of one wave
      uint32_t* __restrict__ wplanes = reinterpret_cast<uint32_t*>(s_obs) + (threadIdx.x >> 6) * ((uint32_t)R::PLANES * wstride);
      for (uint32_t w = (uint32_t)wl; w < (uint32_t)R::PLANES * wstride; w += BSX_WAVE) wplanes[w] = 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (mine) {
        double reward = 0.0;
        float o[Env::HEAD];
        const bsx_bit_sink sink{wplanes, (int)wstride, (uint32_t)(wl * numel + Env::HEAD)};
        type = Env::template step<LOG, MT, true>(a, i, oi, lane, step0 + (uint64_t)t, o, reward, &sink);
        bsx_emit_at<LOG, NOISE, F64, MT, (ROLLOUT ? BSX_POLICY_R(1) : BSX_POLICY_E(1))>(a.ctl, a.out, i, oi, lane, step0 + (uint64_t)t, type, reward);
        float* __restrict__ heads = reinterpret_cast<float*>(a.rows + (uint64_t)R::PLANES * (uint64_t)a.row_plane_words);
        uint32_t head_bits = 0u;
#pragma unroll
        for (int k = 0; k < Env::HEAD; ++k) {
          bool is_float = false;
#pragma unroll
          for (int q = 0; q < R::NF; ++q)
            if ((int)bsx_rows_fpos(R::KIND, q) == k) { heads[(int64_t)q * B + i] = o[k]; is_float = true; }
          if (!is_float) head_bits |= (o[k] != 0.0f ? 1u : 0u) << k;     // (0.0 / 1.0: decode(1) == 1.0f for such a family)
        }
        if (R::NF < Env::HEAD) {
          const bsx_bit_sink hs{wplanes, (int)wstride, (uint32_t)(wl * numel)};
          hs.put(0, 0, head_bits, Env::HEAD);
        }
      }
      bsx_count_types(a.ctl, type, s_cnt);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int64_t wave_lane0 = lane0 + (int64_t)(threadIdx.x & ~63u);
      if (wave_lane0 < B) {                                              // (uniform per wave)
        uint32_t* __restrict__ gp = a.rows + (uint64_t)(wave_lane0 >> 6) * wstride;
#pragma unroll
        for (int p = 0; p < R::PLANES; ++p)
          for (uint32_t w = (uint32_t)wl; w < wstride; w += BSX_WAVE) gp[(uint64_t)p * (uint64_t)a.row_plane_words + w] = wplanes[(uint32_t)p * wstride + w];
      }
    } else if constexpr (DIRECT) {
      // the waves of a block (and the steps of a fused rollout) never wait for each other
      if (mine) {
        double reward = 0.0;
        float o[8];
        BSX_LIFE_AFTER_S(2, (uint32_t)step0);                   // the argument slot and the call counter have arrived
        type = Env::template step<LOG, MT>(a, i, oi, lane, step0 + (uint64_t)t, o, reward);
        BSX_LIFE_AFTER_V(4, type);                              // loads + arithmetic (+ the state stores issued)
        bsx_emit_at<LOG, NOISE, F64, MT, (ROLLOUT ? BSX_POLICY_R(1) : BSX_POLICY_E(1))>(a.ctl, a.out, i, oi, lane, step0 + (uint64_t)t, type, reward);
        // (row-per-lane stores also in a big launch: staging the rows like the fused rollout does left the eager step
        // where it was — 17.5 / 17.5 vs 18.0 / 17.4 us at 2^20 lanes — and cost 4 % at 2^18,
        // profiles/r03/ab_eager_rows_via_lds.log: one memory round trip per launch bounds it, not the write requests;
        // pooled resets in the eager step: 17.6 -> 19.1-19.6 us, the barriers wait for the slowest wave's loads,
        // profiles/r03/ab_eager_pooled_resets.log)
        // Rows of 4-8 floats (cartpole, swing-up): a full wave stages its 64 rows in LDS and stores them as NON-TEMPORAL 16-byte
        // chunks (small_obs_store_rows_wave) — row-per-lane they are 8-byte pieces at the row stride, partial lines that must
        // not be non-temporal (18 -> 25 us), and ordinary stores evict the columns the next call reads.  Round 3 had measured the
        // staging alone as equal (ab_eager_rows_via_lds.log); with non-temporal chunks: cartpole/0 17.4-18.7 -> 15.4-16.0 us per
        // step at 2^20 lanes, 7.8 -> 7.2 at 2^18 (profiles/r06/ab_cartpole_eager_rows_lds_nt.log).
        bool staged = false;
        if constexpr (Env::HAS_REGS) {
          if constexpr (Env::ROWS_VIA_LDS && !ROLLOUT) {
            float* s_rows_e = s_obs;                                // dynamic LDS: small_obs_lds() reserves 256 rows x 8 floats
            const int wl_e = (int)(threadIdx.x & 63u);
            if (i - wl_e + 64 <= B) {                              // (uniform per wave: all 64 lanes are in range)
              small_obs_store_rows_wave<false>(a.out.observation + (oi - wl_e) * (int64_t)numel, o, numel, s_rows_e + (threadIdx.x - wl_e) * 8, wl_e);
              staged = true;
            }
          }
        }
        if (!staged) small_obs_store_row<ROLLOUT>(a.out.observation + oi * (int64_t)numel, o, numel);
      }
      bsx_count_types(a.ctl, type, s_cnt);
    } else {
      constexpr int HEAD = Env::HEAD, PLANES = Env::PLANES;          // numel >= 9 here (bsx_small_direct_shape)
      // LDS: PLANES data planes + 1 plane marking the HEAD elements (a fixed pattern: bits l*numel + k, k < HEAD —
      // built once per launch), `stride` words each; then the 256 lanes' HEAD floats (4 floats of padding either side).
      uint32_t* __restrict__ planes = reinterpret_cast<uint32_t*>(s_obs);
      const int stride = numel * (BSX_BLOCK / 32);                       // words per plane
      uint32_t* __restrict__ head_plane = planes + PLANES * stride;
      float* __restrict__ s_head = reinterpret_cast<float*>(head_plane + stride) + 4;
      // ... then the time fractions 1 - t / L, t = 0..L, when the chain is short enough (small_obs_lds): every lane's row
      // holds one, an f64 division per lane-step that the workgroup's first L + 1 threads now do once per launch
      // (umbrella_length: 13 threads of one wave instead of all four waves)
      // (single steps only: inside the fused rollout's step loop the table made umbrella_length r16 9 % SLOWER, 25.1 ->
      // 27.4 us per step, while the eager step gained 2.6 %: profiles/r05/ab_chain_time_fraction_table.log)
      const float* s_tf = nullptr;
      if (!ROLLOUT && Env::tf_table_fits(a)) {
        float* tab = s_head + BSX_BLOCK * HEAD + 4;
        s_tf = tab;
        if (t == 0) {
          BSX_NO_CONTRACT
          for (int k = threadIdx.x; k <= a.L; k += BSX_BLOCK) tab[k] = (float)(1.0 - (double)k / (double)a.L);
        }
      }
      if (t == 0) {
        for (int w = threadIdx.x; w < (PLANES + 1) * stride; w += BSX_BLOCK) planes[w] = 0u;
      } else {
        __syncthreads();                                                 // every chunk of step t-1 has been read
        for (int w = threadIdx.x; w < PLANES * stride; w += BSX_BLOCK) planes[w] = 0u;
      }
      __syncthreads();
      if (mine) {
        double reward = 0.0;
        float head[HEAD];
        const bsx_bit_sink sink{planes, stride, (uint32_t)((int)threadIdx.x * numel + HEAD), s_tf};
        type = Env::template step<LOG, MT, true>(a, i, oi, lane, step0 + (uint64_t)t, head, reward, &sink);
        bsx_emit_at<LOG, NOISE, F64, MT, (ROLLOUT ? BSX_POLICY_R(1) : BSX_POLICY_E(1))>(a.ctl, a.out, i, oi, lane, step0 + (uint64_t)t, type, reward);
#pragma unroll
        for (int k = 0; k < HEAD; ++k) s_head[threadIdx.x * HEAD + k] = head[k];
        if (t == 0) {
          const bsx_bit_sink hs{head_plane, stride, (uint32_t)((int)threadIdx.x * numel)};
          hs.put(0, 0, 0xFFFFFFFFu, HEAD);
        }
      }
      bsx_count_types(a.ctl, type, s_cnt);
      __syncthreads();

      // Stream the tile: [lanes_here x numel] floats, contiguous in HBM, 16-byte aligned start, every byte written
      // exactly once and only by full 16-byte chunks (1 KiB per wave instruction).  Chunk c is the nibble at bit 4c of
      // each plane, four bit-tests away from a float4; where the HEAD plane's nibble is non-zero the lane's HEAD
      // floats are spliced in from LDS: numel >= 9 means at most ONE lane's HEAD run lies in a chunk, so one magic
      // division finds it and element j of the chunk is s_head[base + j].  What this replaces, and why
      // (profiles/r03/ab_wide_rows*.log): r02 streamed zeros into the HEAD places and patched them with a 12-byte
      // store per lane after a third barrier and the store acknowledgements of the whole tile; storing the HEAD
      // chunks from their own lanes instead (no patch, no third barrier) was no faster — the L2 is bound by write
      // REQUESTS here, and a million scattered partial-line stores are a third of all requests: without any HEAD
      // stores the same kernel takes 86 us instead of 116 (umbrella_distract).
      float* __restrict__ tile = a.out.observation + ((int64_t)t * B + lane0) * (int64_t)numel;
      const int total = lanes_here * numel;
      const bool vec = ((((int64_t)t * B * numel) & 3) == 0);            // [t] slice 16-byte aligned?
      const int n_chunks = vec ? total >> 2 : 0;
      const uint32_t numel_magic = a.numel_magic;                        // f / numel = __umulhi(f, magic), f < 2^16
      bsx_f4* __restrict__ t4 = reinterpret_cast<bsx_f4*>(tile);
      for (int ch = threadIdx.x; ch < n_chunks; ch += BSX_BLOCK) {
        const int sh = (ch & 7) << 2;
        const uint32_t n0 = planes[ch >> 3] >> sh;
        const uint32_t n1 = PLANES > 1 ? planes[stride + (ch >> 3)] >> sh : 0u;
        const uint32_t hm = (head_plane[ch >> 3] >> sh) & 0xFu;
        bsx_f4 q;
        q.x = Env::decode(n0 & 1u, n1 & 1u);
        q.y = Env::decode((n0 >> 1) & 1u, (n1 >> 1) & 1u);
        q.z = Env::decode((n0 >> 2) & 1u, (n1 >> 2) & 1u);
        q.w = Env::decode((n0 >> 3) & 1u, (n1 >> 3) & 1u);
        if (hm != 0u) {
          const int j1 = __ffs((int)hm) - 1;                             // first HEAD element of the chunk
          const uint32_t f1 = ((uint32_t)ch << 2) + (uint32_t)j1;
          const uint32_t l = __umulhi(f1, numel_magic);                  // its lane
          const int base = (int)(l * (uint32_t)HEAD + (f1 - l * (uint32_t)numel)) - j1;   // >= -3: padded
          const float h0 = s_head[base], h1 = s_head[base + 1], h2 = s_head[base + 2], h3 = s_head[base + 3];
          q.x = (hm & 1u) ? h0 : q.x;
          q.y = (hm & 2u) ? h1 : q.y;
          q.z = (hm & 4u) ? h2 : q.z;
          q.w = (hm & 8u) ? h3 : q.w;
        }
        bsx_st<(ROLLOUT ? BSX_POLICY_R(32) : BSX_POLICY_E(32))>(&t4[ch], q);
      }
      // elements beyond the 16-byte chunks (an unaligned [t] slice, or the < 4 floats at the end of an odd tile)
      for (int f = (n_chunks << 2) + (int)threadIdx.x; f < total; f += BSX_BLOCK) {
        const uint32_t b0 = (planes[f >> 5] >> (f & 31)) & 1u;
        const uint32_t b1 = PLANES > 1 ? (planes[stride + (f >> 5)] >> (f & 31)) & 1u : 0u;
        float v = Env::decode(b0, b1);
        if ((head_plane[f >> 5] >> (f & 31)) & 1u) {
          const uint32_t l = __umulhi((uint32_t)f, numel_magic);
          v = s_head[l * (uint32_t)HEAD + ((uint32_t)f - l * (uint32_t)numel)];
        }
        tile[f] = v;
      }
    }
  }
  BSX_LIFE(5);
  bsx_final_barrier();
  BSX_LIFE(6);
  bsx_flush_counts(a.ctl, s_cnt, block_id);
}

template <class Env, bool ROLLOUT, int LOG, int NOISE, int MT, bool DIRECT, bool ROWS = false>
__global__ void __launch_bounds__(BSX_BLOCK) small_obs_kernel(const typename Env::args a, const int n_steps) {
  extern __shared__ __attribute__((aligned(16))) float s_obs[];
  __shared__ unsigned int s_cnt[2];
  small_obs_body<Env, ROLLOUT, LOG, NOISE, MT, DIRECT, false, -1, -1, ROWS>(a, n_steps, blockIdx.x, s_obs, s_cnt);
}

// Eager step of a register-resident family, LPT = 2 or 4 lanes per thread (lean calls of 2^19+ lanes): thread t of
// workgroup b advances lanes b*LPT*256 + h*256 + t, the loads of ALL of them issued before the first use.  At 2^20 lanes
// the one-lane kernel is 4096 workgroups = two dispatch rounds of a launch whose every wave is one dependent chain
// {column loads -> step -> stores}; this one is a single round with LPT times the bytes in flight per wave.
// r04 measured cartpole / mountain_car only, called them equal and left it off.  Round 5, same call, two repetitions, with
// bandit, discounting_chain and memory_len now register-resident (profiles/r05/ab_eager_lanes_per_thread.log), 1 -> 2 lanes
// per thread at 2^20 lanes: bandit 9.55 -> 8.65 us, discounting_chain 8.7 -> 7.2, memory_len 12.0 -> 10.2, mountain_car
// 9.95 -> 9.0, cartpole 18.9 -> 18.5 (noise); at 2^19: bandit 6.3 -> 6.1, memory_len 7.75 -> 7.1, mountain_car 6.7 -> 6.4,
// discounting_chain equal, cartpole 10.8 -> 11.2 (SLOWER); at 2^18 all equal.  Four lanes per thread are never better than
// two (discounting_chain 8.1, memory_len 10.4 at 2^20; everything slower at 2^19).  Env::EAGER_LPT_MIN_BLOCKS per family.
template <class Env, int V, int LPT>
__global__ void __launch_bounds__(BSX_BLOCK) small_obs_eager2_kernel(const typename Env::args a) {
  __shared__ unsigned int s_cnt[2];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int numel = Env::numel_of(V);
  const int64_t B = a.ctl.n_lanes;
  const uint64_t step = bsx_step_of(a.ctl);
  int64_t i[LPT];
  bool mine[LPT];
  typename Env::regs rg[LPT];
  int act[LPT], type[LPT];
#pragma unroll
  for (int h = 0; h < LPT; ++h) {
    i[h] = (int64_t)blockIdx.x * (LPT * BSX_BLOCK) + h * BSX_BLOCK + threadIdx.x;
    mine[h] = i[h] < B;
    Env::clear(rg[h]);
    act[h] = 0;
    type[h] = -1;
    if (mine[h]) {
      Env::load(a, i[h], rg[h]);
      if (!a.ctl.force_reset) act[h] = bsx_action(a.ctl, a.action, i[h], step);
    }
  }
#pragma unroll
  for (int h = 0; h < LPT; ++h) {
    if (mine[h]) {
      double reward = 0.0;
      float o[8];
      type[h] = Env::template core<0, 0, false, false, false, V>(a, rg[h], act[h], i[h], a.ctl.lane_offset + (uint64_t)i[h], step, o, reward);
      bsx_emit_at<0, 0, false, -1, BSX_POLICY_E(1)>(a.ctl, a.out, i[h], i[h], a.ctl.lane_offset + (uint64_t)i[h], step, type[h], reward);
      small_obs_store_row<false>(a.out.observation + i[h] * (int64_t)numel, o, numel);
      Env::store(a, i[h], rg[h]);
    }
    bsx_count_types(a.ctl, type[h], s_cnt);
  }
  bsx_final_barrier();
  bsx_flush_counts(a.ctl, s_cnt, blockIdx.x);
}

// The lean fused rollout of a register-resident family (no Logging wrapper, no RewardNoise, counter-based draws):
// one instantiation per (big launch, variant, table in LDS) — everything the step loop would otherwise carry as
// run-time conditions in scalar registers.
// (Tried: holding the pooled cartpole loop, 65 VGPRs, to 64 = 8 waves per SIMD, so that the 16 workgroups a CU runs at
// 2^20 lanes are 8 + 8 instead of 7 + 7 + 2.  amdgpu_waves_per_eu(8, 8) also caps the SCALAR registers at 80 of the
// 102 — the attribute that made sweep_phase0_kernel spill 157 of them in r03 — and the ~35 reloads per step it costs
// here cancel the gain: 10.2-10.3 vs 10.0-10.1 us per step, profiles/r04/ab_rollout_8_waves.log; amdgpu_num_vgpr(64)
// is ignored by this compiler.)
template <class Env, bool BIG, int V, bool TAB>
__global__ void __launch_bounds__(BSX_BLOCK) small_obs_lean_rollout_kernel(const typename Env::args a, const int n_steps) {
  extern __shared__ __attribute__((aligned(16))) float s_obs[];
  __shared__ unsigned int s_cnt[2];
  small_obs_body<Env, true, 0, 0, 0, true, BIG, V, TAB ? 1 : 0>(a, n_steps, blockIdx.x, s_obs, s_cnt);
}

// Dynamic LDS of one workgroup stepping `a`.
template <class Env>
static size_t small_obs_lds(const typename Env::args& a) {
  if constexpr (Env::PACKED)     // PLANES data planes + the HEAD-position plane + the lanes' HEAD floats (padded) + the time fractions
    return bsx_small_direct_shape(a.obs_numel) ? 0 : (size_t)(Env::PLANES + 1) * a.obs_numel * (BSX_BLOCK / 32) * 4 + (size_t)(BSX_BLOCK * Env::HEAD + 8) * 4 +
                                                     (Env::tf_table_fits(a) ? ((size_t)a.L + 1) * 4 : 0);
  else if constexpr (Env::HAS_REGS) return Env::ROWS_VIA_LDS ? (size_t)BSX_BLOCK * 8 * 4 : 0;     // the eager step's row staging (small_obs_body, DIRECT)
  else return 0;
}
// does a single-step call of this segment take the row path (flat bit planes into a.rows + the wide-row store stream)?
template <class Env>
static bool small_obs_rows(const typename Env::args& a) {
  if constexpr (Env::PACKED) return a.rows != nullptr && !bsx_small_direct_shape(a.obs_numel);
  else return false;
}
// dynamic LDS of one workgroup on the row path: every wave's own PLANES x 2 * numel plane words
template <class Env>
static size_t small_obs_rows_lds(const typename Env::args& a) {
  if constexpr (Env::PACKED) return (size_t)(BSX_BLOCK / BSX_WAVE) * Env::PLANES * 2 * (size_t)a.obs_numel * 4;
  else return 0;
}
// the arguments of the wide-row store stream of a segment on the row path
template <class Env>
static bsx_row_seg small_obs_row_seg(const typename Env::args& a) {
  bsx_row_seg g{};
  if constexpr (Env::PACKED) {
    g.obs = a.out.observation; g.planes = a.rows; g.n_lanes = a.ctl.n_lanes; g.plane_words = (uint64_t)a.row_plane_words;
    g.numel = (uint32_t)a.obs_numel; g.numel_magic = bsx_div_magic(g.numel); g.dv = bsx_make_div64(g.numel);
  }
  return g;
}
// ... and of a fused rollout: the tables a register-resident family stages (cartpole: the time fractions)
template <class Env>
static size_t small_obs_rollout_lds(const typename Env::args& a) {
  if constexpr (Env::HAS_REGS && Env::PACKED) return bsx_small_direct_shape(a.obs_numel) ? Env::table_bytes(a) : small_obs_lds<Env>(a);
  else if constexpr (Env::HAS_REGS) return Env::table_bytes(a);
  else return small_obs_lds<Env>(a);
}

// One workgroup of a grouped launch: the single-step body.  A segment without Logging wrapper, RewardNoise and
// MT19937 draws (uniform per workgroup) runs the lean instantiation, like a stand-alone call does: a cartpole
// workgroup then issues a third fewer instructions, and the heavy workgroups are what a sweep's lane advance
// waits for (profiles/r02/sweep_phase0_trace.json).
// MT = 0: the group holds no segment in MT19937-exact mode (the whole-sweep group refuses them, bsx_mixed_put): the
// wrapped segments' bodies are compiled without the generator's twist and numpy's legacy samplers.
template <class Env, bool D, int MT = -1, bool ROWS = false>
__device__ __forceinline__ void small_obs_group_body_d(const typename Env::args& a, const uint32_t blk, float* s_obs,
                                                       unsigned int* s_cnt) {
  if (bsx_ctl_lean(a.ctl)) small_obs_body<Env, false, 0, 0, 0, D, false, -1, -1, ROWS>(a, 1, blk, s_obs, s_cnt);
  else small_obs_body<Env, false, -1, -1, MT, D, false, -1, -1, ROWS>(a, 1, blk, s_obs, s_cnt);
}
// ROWS_OK: the launch is followed by the wide-row store stream (the whole-sweep group's phase 1): a segment that
// brought a row scratch leaves its rows there, packed; everywhere else wide rows go through the LDS bit planes.
template <class Env, int MT = -1, bool ROWS_OK = false>
__device__ __forceinline__ void small_obs_group_body(const typename Env::args& a, const uint32_t blk, float* s_obs,
                                                     unsigned int* s_cnt) {
  if constexpr (Env::PACKED) {
    if (!bsx_small_direct_shape(a.obs_numel)) {                 // uniform per workgroup
      if constexpr (ROWS_OK) {
        if (a.rows != nullptr) {
          small_obs_group_body_d<Env, true, MT, true>(a, blk, s_obs, s_cnt);
          return;
        }
      }
      small_obs_group_body_d<Env, false, MT>(a, blk, s_obs, s_cnt);
      return;
    }
  }
  small_obs_group_body_d<Env, true, MT>(a, blk, s_obs, s_cnt);
}

// Grouped launch: every workgroup looks up its segment and runs the single-step body on that
// segment's argument struct (device memory).
template <class Env>
__global__ void __launch_bounds__(BSX_BLOCK) small_obs_group_kernel(const typename Env::args* __restrict__ table,
                                                                    const bsx_group_index gi) {
  extern __shared__ __attribute__((aligned(16))) float s_obs[];
  __shared__ unsigned int s_cnt[2];
  const bsx_group_slot w = bsx_group_find(gi, (int)blockIdx.x);
  small_obs_group_body<Env>(table[w.seg], w.block, s_obs, s_cnt);
}

template <class Env>
static int small_obs_group_launch(bsx_group* g, int phase, hipStream_t st) {
  if (phase == 1) return 0;                 // one kernel per step: everything happens in phase 0
  const dim3 grid((unsigned)g->total_blocks), block(BSX_BLOCK);
  const typename Env::args* table = (const typename Env::args*)g->d_args;
  small_obs_group_kernel<Env><<<grid, block, g->lds_bytes, st>>>(table, g->index1());
  return (int)hipGetLastError();
}

// A BSX_FAM_SMALL_MIXED group holds segments of ANY of the families in this file: every segment's
// argument struct sits in a fixed-stride slot next to a family tag, and one launch
// advances them all (the kernel switches on the tag per workgroup).  Six ~8 us launches of a
// heterogeneous sweep become one.
#define SMALL_MIXED_STRIDE 1024
// launch of a BSX_FAM_SMALL_MIXED group (sweep_mixed.hip)
int bsx_small_mixed_launch(bsx_group* g, int phase, hipStream_t st);


// Records one segment of a small-observation family in a group (of its own family, or mixed).
template <class Env>
static int small_obs_group_put(bsx_group* g, int32_t family, int32_t index, const bsx_call_t* call,
                               const typename Env::args& a_in) {
  static_assert(sizeof(typename Env::args) <= SMALL_MIXED_STRIDE, "argument struct exceeds the mixed-group slot");
  typename Env::args a = a_in;
  const uint64_t nb = (uint64_t)(a.ctl.n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  if (g != nullptr && g->family == BSX_FAM_SWEEP_MIXED) {     // one segment of the whole-sweep group
    if (small_obs_rows<Env>(a)) {
      // wide rows with a row scratch: phase 0 leaves them packed, the phase-1 store stream writes the observations
      const bsx_row_seg sg = small_obs_row_seg<Env>(a);
      return bsx_mixed_put(g, family, index, call, &a, sizeof(a), &sg, sizeof(sg), nb,
                           bsx_flat_blocks((uint64_t)a.ctl.n_lanes * sg.numel, BSX_ROW_STREAM_K), small_obs_rows_lds<Env>(a));
    }
    return bsx_mixed_put(g, family, index, call, &a, sizeof(a), nullptr, 0, nb, 0, small_obs_lds<Env>(a));   // phase 0 only
  }
  if constexpr (Env::PACKED) a.rows = nullptr;                // single-launch groups: the LDS bit planes
  const size_t lds = small_obs_lds<Env>(a);
  const bool mixed = g != nullptr && g->family == BSX_FAM_SMALL_MIXED;
  int rc = bsx_group_check_set(g, mixed ? BSX_FAM_SMALL_MIXED : family, index, call,
                               mixed ? SMALL_MIXED_STRIDE : sizeof(typename Env::args), mixed ? sizeof(int32_t) : 0, BSX_BLOCK);
  if (rc != 0) return rc;
  memcpy(&g->args[(size_t)index * g->arg_size], &a, sizeof(a));
  if (mixed) memcpy(&g->args2[(size_t)index * sizeof(int32_t)], &family, sizeof(int32_t));
  if (nb > 0x3FFFFFFFull) return BSX_EINVAL;
  g->blocks[index] = (int32_t)nb;
  if (lds > g->lds_bytes) g->lds_bytes = lds;
  g->is_set[index] = 1;
  g->launch = mixed ? bsx_small_mixed_launch : small_obs_group_launch<Env>;
  return 0;
}

// variant 1 of a register-resident family that has one (else the instantiation is the same kernel as variant 0's)
template <class Env>
static constexpr int small_obs_v1() {
  if constexpr (Env::HAS_REGS) return Env::N_VARIANTS > 1 ? 1 : 0;
  else return -1;
}

// does the family's fused rollout have a variant for big launches (small_obs_regs_rollout, BIG)?
template <class Env>
static constexpr bool small_obs_has_big() {
  if constexpr (Env::HAS_REGS) return Env::POOLED_RESETS || Env::ROWS_VIA_LDS;
  else return false;
}

// Launches the lean fused rollout of a register-resident family (small_obs_lean_rollout_kernel).
template <class Env>
static void launch_regs_rollout(const typename Env::args& a, int n_steps, int v, bool big, dim3 g, dim3 b, size_t lds, hipStream_t st) {
  if constexpr (Env::HAS_REGS) {
    constexpr bool HB = small_obs_has_big<Env>();
    constexpr int V1 = small_obs_v1<Env>();
    const bool tab = Env::table_fits(a);
#define REGS_ROLLOUT(BIGV, VV)                                                                                     \
    {                                                                                                              \
      if (tab) small_obs_lean_rollout_kernel<Env, BIGV, VV, true><<<g, b, lds, st>>>(a, n_steps);                  \
      else small_obs_lean_rollout_kernel<Env, BIGV, VV, false><<<g, b, lds, st>>>(a, n_steps);                     \
    }
    if (v == 1 && big) REGS_ROLLOUT(HB, V1)
    else if (v == 1) REGS_ROLLOUT(false, V1)
    else if (big) REGS_ROLLOUT(HB, 0)
    else REGS_ROLLOUT(false, 0)
#undef REGS_ROLLOUT
  }
}

// LPT lanes per thread for the lean eager step of a register-resident family (Env::EAGER_LPT_MIN_BLOCKS: from this many
// one-lane workgroups up, 0 = never; Env::EAGER_LPT)
template <class Env, int LPT>
static void launch_eager_lpt(const typename Env::args& a, int v, hipStream_t st) {
  if constexpr (Env::HAS_REGS) {
    const dim3 g((unsigned)((a.ctl.n_lanes + LPT * BSX_BLOCK - 1) / (LPT * BSX_BLOCK))), b(BSX_BLOCK);
    if (v == 1) small_obs_eager2_kernel<Env, small_obs_v1<Env>(), LPT><<<g, b, 0, st>>>(a);
    else small_obs_eager2_kernel<Env, 0, LPT><<<g, b, 0, st>>>(a);
  }
}
template <class Env>
static void launch_eager2(const typename Env::args& a, int v, int lpt, hipStream_t st) {
#ifdef BSX_TUNING
  if (lpt == 4) { launch_eager_lpt<Env, 4>(a, v, st); return; }      // measured: never better than 2 (see below)
#endif
  (void)lpt;
  launch_eager_lpt<Env, 2>(a, v, st);
}

template <class Env>
static int launch_small_obs(const typename Env::args& a, int n_steps, void* hip_stream) {
  hipStream_t st = (hipStream_t)hip_stream;
  if (n_steps < 1) return BSX_EINVAL;
  const bool logging = a.ctl.log.steps != nullptr, noise = a.ctl.wrap_kind >= BSX_WRAP_NOISE;
  // (the wrapped rollouts are compiled with and without the MT19937-exact draws: the common wrapped run draws from the
  // counter-based stream, and the generator's twist + numpy's legacy samplers cost the loop 50-70 VGPRs)
  const bool mt = a.ctl.mt_state != nullptr;
  const bool lean = bsx_ctl_lean(a.ctl);
  // the register-resident families' lean rollouts are compiled per variant (row length, swing-up): -1 = none
  // (and address every [B, numel] slab with 32-bit byte offsets: a slab of 4 GiB or more takes the generic loop)
  int regs_v = -1, Env_variant = 0;
  bool eager2 = false;
  int eager_lpt = 2;
  if constexpr (Env::HAS_REGS) {
    Env_variant = Env::variant_of(a);
    if (a.ctl.n_lanes * (int64_t)a.obs_numel * 4 < ((int64_t)1 << 32)) regs_v = Env_variant;
    if (Env::PACKED && !bsx_small_direct_shape(a.obs_numel)) regs_v = -1;      // wide rows: the LDS bit planes, step by step
    static const int eager2_min_blocks = bsx_env_int("BSX_EAGER2_MIN_BLOCKS", Env::EAGER_LPT_MIN_BLOCKS);
    static const int eager_lpt_knob = bsx_env_int("BSX_EAGER_LPT", Env::EAGER_LPT);
    eager2 = eager2_min_blocks > 0 && Env_variant >= 0 && (a.ctl.n_lanes + BSX_BLOCK - 1) / BSX_BLOCK >= eager2_min_blocks;
    eager_lpt = eager_lpt_knob;
  }
  const int64_t blocks = (a.ctl.n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  if (blocks > 0x7FFFFFFF) return BSX_EINVAL;
  const size_t lds = n_steps > 1 ? small_obs_rollout_lds<Env>(a) : small_obs_lds<Env>(a);
  const dim3 g((unsigned)blocks), b(BSX_BLOCK);
  // the register-resident families' fused rollout: pooled resets / staged rows from 2^19 lanes (8 workgroups per CU) up
  static const int big_min_blocks = bsx_env_int("BSX_REGS_ROLLOUT_BIG_MIN_BLOCKS", 2048);
  const bool big = small_obs_has_big<Env>() && blocks >= big_min_blocks;
  // Per-thread stores (4-byte / 12-byte / 8-byte stores, each wave writing one contiguous range) against
  // an f32 LDS tile, measured on bandit, discounting_chain, cartpole, mountain_car, memory_len: eager equal
  // or 2-4 % faster, fused rollout 3-15 % faster (profiles/r02/ab_small_direct_stores.log).  Three 4-byte
  // stores at stride 12 for the 3-float rows were 5-8 % SLOWER than the tile; one global_store_dwordx3 is
  // faster.
#define SMALL_OBS_LAUNCH(D)                                                                                \
  {                                                                                                        \
    if (n_steps == 1 && lean && eager2) launch_eager2<Env>(a, Env_variant, eager_lpt, st);                 \
    else if (n_steps == 1 && lean) small_obs_kernel<Env, false, 0, 0, 0, D><<<g, b, lds, st>>>(a, 1);      \
    else if (n_steps == 1) small_obs_kernel<Env, false, -1, -1, -1, D><<<g, b, lds, st>>>(a, 1);           \
    else if (logging && noise && !mt) small_obs_kernel<Env, true, 1, 1, 0, D><<<g, b, lds, st>>>(a, n_steps); \
    else if (logging && !mt) small_obs_kernel<Env, true, 1, 0, 0, D><<<g, b, lds, st>>>(a, n_steps);       \
    else if (noise && !mt) small_obs_kernel<Env, true, 0, 1, 0, D><<<g, b, lds, st>>>(a, n_steps);         \
    else if (logging && noise) small_obs_kernel<Env, true, 1, 1, -1, D><<<g, b, lds, st>>>(a, n_steps);    \
    else if (logging) small_obs_kernel<Env, true, 1, 0, -1, D><<<g, b, lds, st>>>(a, n_steps);             \
    else if (noise) small_obs_kernel<Env, true, 0, 1, -1, D><<<g, b, lds, st>>>(a, n_steps);               \
    else if (lean && regs_v >= 0) launch_regs_rollout<Env>(a, n_steps, regs_v, big, g, b, lds, st);        \
    else if (lean) small_obs_kernel<Env, true, 0, 0, 0, D><<<g, b, lds, st>>>(a, n_steps);                 \
    else small_obs_kernel<Env, true, 0, 0, -1, D><<<g, b, lds, st>>>(a, n_steps);                          \
  }
  if constexpr (Env::PACKED) {
    if (!bsx_small_direct_shape(a.obs_numel)) {
      if (n_steps == 1 && small_obs_rows<Env>(a)) {
        // wide rows, a single step, a row scratch: lane advance (rows packed into the scratch) + the store stream
        // that decodes them (row_stream.h) instead of the one launch with the LDS bit planes and its three barriers
        const size_t rlds = small_obs_rows_lds<Env>(a);
        if (lean) small_obs_kernel<Env, false, 0, 0, 0, true, true><<<g, b, rlds, st>>>(a, 1);
        else small_obs_kernel<Env, false, -1, -1, -1, true, true><<<g, b, rlds, st>>>(a, 1);
        const bsx_row_seg sg = small_obs_row_seg<Env>(a);
        static const int row_k = bsx_env_int("BSX_ROW_STREAM_K", BSX_ROW_STREAM_K);
        const int k = row_k == 1 || row_k == 4 ? row_k : 2;
        const uint64_t sblocks = bsx_flat_blocks((uint64_t)a.ctl.n_lanes * sg.numel, k);
        if (sblocks > 0x7FFFFFFFull) return BSX_EINVAL;
        const dim3 gs((unsigned)sblocks);
        if (k == 1) bsx_row_stream_kernel<typename Env::rows_t, 1><<<gs, b, 0, st>>>(sg);
        else if (k == 4) bsx_row_stream_kernel<typename Env::rows_t, 4><<<gs, b, 0, st>>>(sg);
        else bsx_row_stream_kernel<typename Env::rows_t, 2><<<gs, b, 0, st>>>(sg);
        return bsx_launch_status();
      }
      SMALL_OBS_LAUNCH(false)
      return bsx_launch_status();
    }
  }
  SMALL_OBS_LAUNCH(true)
#undef SMALL_OBS_LAUNCH
  return bsx_launch_status();
}

// ------------------------------------------------------------------------------ bandit
// A family with a one-word state and no table, register-resident in a fused rollout (small_obs_regs_rollout): what the
// interface needs beyond regs / load / store / core.
struct small_regs_defaults {
  static constexpr bool POOLED_RESETS = false, ROWS_VIA_LDS = false;
  static constexpr int N_VARIANTS = 1, TABLE_MAX_BYTES = 0;
  static constexpr int EAGER_LPT_MIN_BLOCKS = 2048, EAGER_LPT = 2;   // lean eager step: lanes per thread (launch_small_obs)
  __device__ static __forceinline__ void reset_part(const void*, uint64_t, uint64_t, int, unsigned, bsx_reset_pool*) {}
};

struct bandit_env : small_regs_defaults {
  // (register-resident in a fused rollout: the generic loop re-read the reset flag from L2 behind a drain of the previous
  // step's stores and read-modify-wrote the f64 regret column on every second step — 16 of the step's 33 bytes)
  static constexpr bool HAS_REGS = true, PACKED = false;
  __host__ __device__ static constexpr int numel_of(int) { return 1; }
  struct regs { int32_t st; double inf0; };                    // inf0: total_regret in a fused rollout
  struct args {
    bsx_ctl ctl; const int32_t* action; int32_t* state; bsx_timestep_t out; double* info;
    int32_t obs_numel; int32_t num_actions; double rewards[BSX_BANDIT_MAX_ACTIONS];
  };
  static int variant_of(const args&) { return 0; }
  template <bool NOFORCE = false>
  __device__ static __forceinline__ bool wants_reset(const args&, const regs&) { return false; }
  __device__ static __forceinline__ void clear(regs& r) { r.st = 0; }
  __device__ static __forceinline__ bool reset_pending(const regs& r) { return r.st != 0; }
  __device__ static __forceinline__ void reset_part(const args&, uint64_t, uint64_t, int, unsigned, bsx_reset_pool*) {}
  __host__ __device__ static bool table_fits(const args&) { return false; }
  static size_t table_bytes(const args&) { return 0; }
  __device__ static __forceinline__ bsx_lds_table stage_tables(const args&, float*) { return (bsx_lds_table)0; }
  template <int V = -1>
  __device__ static __forceinline__ void load_info(const args& a, int64_t i, regs& r) { r.inf0 = a.info[i]; }
  template <int V = -1>
  __device__ static __forceinline__ void store_info(const args& a, int64_t i, const regs& r) { a.info[i] = r.inf0; }
  __device__ static __forceinline__ void load(const args& a, int64_t i, regs& r) { r.st = a.state[i]; }
  __device__ static __forceinline__ void store(const args& a, int64_t i, const regs& r) { a.state[i] = r.st; }
  // (the same transitions as step() below; tests/test_gpu_rollout.py holds rollout(T) to T step() calls bit for bit)
  template <int LOG, int MT, bool IREGS = false, bool TAB = false, bool POOL = false, int V = -1, bool NOFORCE = false>
  __device__ static __forceinline__ int core(const args& a, regs& rg, int act, int64_t i, uint64_t, uint64_t,
                                             float* o, double& reward, bsx_lds_table = (bsx_lds_table)0,
                                             const bsx_reset_pool* = nullptr) {
    BSX_NO_CONTRACT
    o[0] = 1.0f;                                                // bandit.py:54 (ones)
    if ((!NOFORCE && a.ctl.force_reset) || rg.st) { rg.st = 0; return BSX_FIRST; }
    if (act < 0 || act >= a.num_actions) {                      // reference: IndexError (bandit.py:61)
      bsx_note_invalid_action(a.ctl, i);
      act = act < 0 ? 0 : a.num_actions - 1;
    }
    reward = a.rewards[act];                                    // :61
    if constexpr (IREGS) rg.inf0 += 1.0 - reward; else a.info[i] += 1.0 - reward;   // :62
    rg.st = 1;
    return BSX_LAST;                                            // :64
  }
  template <int LOG, int MT>
  __device__ static int step(const args& a, int64_t i, int64_t oi, uint64_t, uint64_t step, float* o, double& reward) {
    BSX_NO_CONTRACT
    o[0] = 1.0f;                                                // bandit.py:54 (ones)
    if (a.ctl.force_reset || a.state[i]) { a.state[i] = 0; return BSX_FIRST; }
    int act = bsx_action(a.ctl, a.action, oi, step);
    if (act < 0 || act >= a.num_actions) {                      // reference: IndexError (bandit.py:61)
      bsx_note_invalid_action(a.ctl, i);
      act = act < 0 ? 0 : a.num_actions - 1;                    // never read OOB
    }
    reward = a.rewards[act];                                    // :61
    a.info[i] += 1.0 - reward;                                  // :62 (every second call of every lane: a plain
                                                                //      read-modify-write beats 2^20 atomics, 9.6 vs 11.2 us)
    a.state[i] = 1;
    return BSX_LAST;                                            // :64
  }
};

// the chains' step(): the action loaded beside the state word (1) or where the episode needs it (0)
#ifndef BSX_EARLY_ACTION
#define BSX_EARLY_ACTION 1
#endif
// The time fraction 1 - t / L of the chains' rows (memory_chain.py:64, umbrella_chain.py:64): from the workgroup's LDS
// table where the PACKED path staged one (bsx_bit_sink::tf), else the f64 division itself.  Evaluated ONCE per step(),
// before the reset / step paths part ways.
template <bool PACK, class Sink>
__device__ __forceinline__ float bsx_chain_time_fraction(int t, int L, const Sink* sink) {
  BSX_NO_CONTRACT
  if constexpr (PACK) {
    if (sink->tf != nullptr) return sink->tf[t];               // (t <= L; uniform branch)
  }
  return (float)(1.0 - (double)t / (double)L);
}

// ------------------------------------------------------------------------------ memory_chain
#define MC_RESET_BIT (1 << 28)
struct memory_chain_env {
  // Register-resident in a fused rollout of SHORT rows (num_bits <= 6: the row is stored by the lane's own thread) — every
  // memory_len id has one context bit.  The generic rollout re-read the lane's state word and context from L2 on every
  // step behind a drain of the previous step's stores (three dependent round trips per step): memory_len/10 took 13.6 us
  // per step inside rollout(16) against 12.3 us for an eager step() (profiles/r05/bench_default_call1.json).
  static constexpr bool HAS_REGS = true, PACKED = true, POOLED_RESETS = false, ROWS_VIA_LDS = false;
  static constexpr int EAGER_LPT_MIN_BLOCKS = 2048, EAGER_LPT = 2;
  static constexpr int N_VARIANTS = 1;
  __host__ __device__ static constexpr int numel_of(int) { return 3; }     // variant 0: one context bit, rows of 3 floats
  struct regs { int32_t st; uint64_t ctx; double inf[2]; };                // inf: total_perfect, total_regret in a fused rollout
  struct args {
    bsx_ctl ctl; const int32_t* action; int32_t* state; uint64_t* context; bsx_timestep_t out;
    double* info; int32_t obs_numel; int32_t L; int32_t nb; uint32_t numel_magic;
    uint32_t* rows; int64_t row_plane_words;                   // bsx_call_t.row_scratch (bsx_rows.h) + words per plane, or nullptr
  };
  // Packed rows: HEAD = [time, query]; element 2+b is 0 unless t == 0, then +-1 by context bit b: plane 0 says
  // "non-zero", plane 1 carries the context bit (memory_rows, row_stream.h).
  // (PACKED path: the time fractions in LDS when the chain has at most 1024 steps)
  __host__ __device__ static bool tf_table_fits(const args& a) { return a.L <= 1023; }
  typedef memory_rows rows_t;
  static constexpr int HEAD = rows_t::HEAD, PLANES = rows_t::PLANES;
  __device__ static float decode(uint32_t nonzero, uint32_t bit) { return rows_t::decode(nonzero, bit); }
  template <bool PACK, class Sink>
  __device__ static void observe(const args& a, float* o, int t, int query, uint64_t ctx, const Sink* sink) {
    BSX_NO_CONTRACT
    // (PACK: o[0], the time fraction of :64, is step()'s — bsx_chain_time_fraction)
    if constexpr (!PACK) o[0] = (float)(1.0 - (double)t / (double)a.L);   // memory_chain.py:64
    o[1] = (t == a.L - 1) ? (float)query : 0.0f;                // :66-67
    if constexpr (PACK) {
      if (t == 0) {                                             // :69-70 (the tile is zero-filled before every step)
        const int n0 = a.nb < 32 ? a.nb : 32;
        sink->put(0, 0, 0xFFFFFFFFu, n0);
        sink->put(1, 0, (uint32_t)ctx, n0);
        if (a.nb > 32) {
          sink->put(0, 1, 0xFFFFFFFFu, a.nb - 32);
          sink->put(1, 1, (uint32_t)(ctx >> 32), a.nb - 32);
        }
      }
    } else {
      for (int b = 0; b < a.nb; ++b)                            // :69-70
        o[2 + b] = (t == 0) ? (float)(2 * (int)((ctx >> b) & 1ull) - 1) : 0.0f;
    }
  }
  // ---- the register-resident form (small_obs_regs_rollout): state word + context in registers for the T steps
  static int variant_of(const args& a) { return a.nb == 1 ? 0 : -1; }
  template <bool NOFORCE = false>
  __device__ static __forceinline__ bool wants_reset(const args&, const regs&) { return false; }
  __device__ static __forceinline__ void clear(regs& r) { r.st = 0; r.ctx = 0ull; }
  __device__ static __forceinline__ bool reset_pending(const regs& r) { return (r.st & MC_RESET_BIT) != 0; }
  __device__ static __forceinline__ void reset_part(const args&, uint64_t, uint64_t, int, unsigned, bsx_reset_pool*) {}
  // the time fraction 1 - t / L (an f64 division per step) from a table in LDS that the workgroup fills once per launch with
  // that same division
  static constexpr int TABLE_MAX_BYTES = 16384;
  __host__ __device__ static bool table_fits(const args& a) { return ((int64_t)a.L + 1) * 4 <= TABLE_MAX_BYTES; }
  static size_t table_bytes(const args& a) { return table_fits(a) ? ((size_t)a.L + 1) * 4 : 0; }
  __device__ static __forceinline__ bsx_lds_table stage_tables(const args& a, float* s_dyn) {
    BSX_NO_CONTRACT
    for (int k = threadIdx.x; k <= a.L; k += BSX_BLOCK) s_dyn[k] = (float)(1.0 - (double)k / (double)a.L);
    return (bsx_lds_table)s_dyn;
  }
  template <int V = -1>
  __device__ static __forceinline__ void load_info(const args& a, int64_t i, regs& r) { r.inf[0] = a.info[i]; r.inf[1] = a.info[a.ctl.n_lanes + i]; }
  template <int V = -1>
  __device__ static __forceinline__ void store_info(const args& a, int64_t i, const regs& r) { a.info[i] = r.inf[0]; a.info[a.ctl.n_lanes + i] = r.inf[1]; }
  __device__ static __forceinline__ void load(const args& a, int64_t i, regs& r) { r.st = a.state[i]; r.ctx = a.context[i]; }
  __device__ static __forceinline__ void store(const args& a, int64_t i, const regs& r) { a.state[i] = r.st; a.context[i] = r.ctx; }
  // One reset()/step() of the lane in `rg` (memory_chain.py:60-97; the same transitions, draws and info updates as step()
  // below — tests/test_gpu_rollout.py holds rollout(T) to T step() calls bit for bit).  Short rows only: o[0 .. nb + 2).
  template <int LOG, int MT, bool IREGS = false, bool TAB = false, bool POOL = false, int V = -1, bool NOFORCE = false>
  __device__ static __forceinline__ int core(const args& a, regs& rg, const int act, int64_t i, uint64_t lane, uint64_t step,
                                             float* o, double& reward, bsx_lds_table s_tf = (bsx_lds_table)0,
                                             const bsx_reset_pool* = nullptr) {
    BSX_NO_CONTRACT
    const int nb = V == 0 ? 1 : a.nb;
    const int32_t st = rg.st;
    int t = st & 0xFFFFF, query = (st >> 20) & 0xFF;
    uint64_t ctx = rg.ctx;
    const bool reset = (!NOFORCE && a.ctl.force_reset) || (st & MC_RESET_BIT);
    if (reset) {                                                // :91-97
      bsx_draws d;
      bsx_draws_begin<MT>(&d, a.ctl, i, lane, step);
      ctx = 0;
      if (MT == 0 || d.mt == nullptr) {
        ctx = (uint64_t)bsx_word(&d) & ((1ull << nb) - 1ull);   // BernVec(nb), nb <= 6 here: the low bits of one word
      } else {
        uint32_t w = 0;
        for (int b = 0; b < nb; ++b) ctx |= (uint64_t)bsx_bern_vec_bit(&d, b, &w) << b;
      }
      query = (int)bsx_randint(&d, (uint32_t)nb);
      bsx_draws_end<MT>(&d, a.ctl, i);
      t = 0;
      rg.ctx = ctx;
    }
    // the observation of the state BEFORE the step's increment (:74; after a reset: of the fresh state)
    if constexpr (TAB) o[0] = s_tf[t];                          // (t <= L)
    else o[0] = (float)(1.0 - (double)t / (double)a.L);         // :64
    o[1] = (t == a.L - 1) ? (float)query : 0.0f;                // :66-67
#pragma unroll
    for (int b = 0; b < 6; ++b)
      if (b < nb) o[2 + b] = (t == 0) ? (float)(2 * (int)((ctx >> b) & 1ull) - 1) : 0.0f;   // :69-70
    if (reset) { rg.st = t | (query << 20); return BSX_FIRST; }
    t += 1;                                                     // :75
    if (t - 1 < a.L) { rg.st = t | (query << 20); return BSX_MID; }   // :77-79
    const bool hit = act == (int)((ctx >> query) & 1ull);
    reward = hit ? 1.0 : -1.0;                                  // :83-88
    if constexpr (IREGS) {
      if (hit) rg.inf[0] += 1.0; else rg.inf[1] += 2.0;
    } else {
      const bool quiet = a.L >= 8 && bsx_info_quiet<LOG>(a.ctl);
      if (hit) bsx_info_add(quiet, &a.info[i], 1.0); else bsx_info_add(quiet, &a.info[a.ctl.n_lanes + i], 2.0);
    }
    rg.st = t | (query << 20) | MC_RESET_BIT;
    return BSX_LAST;
  }
  template <int LOG, int MT, bool PACK = false, class Sink = bsx_bit_sink>
  __device__ static int step(const args& a, int64_t i, int64_t oi, uint64_t lane, uint64_t step, float* o, double& reward,
                             const Sink* sink = nullptr) {
    int32_t st = a.state[i];
    const int act = BSX_EARLY_ACTION && !a.ctl.force_reset ? bsx_action(a.ctl, a.action, oi, step) : 0;   // (see umbrella_chain_env::step)
    int t = st & 0xFFFFF, query = (st >> 20) & 0xFF;
    uint64_t ctx = a.context[i];
    const bool resets = a.ctl.force_reset || (st & MC_RESET_BIT);
    // :64 — of the state BEFORE the increment (:74), or of the fresh one (short rows: observe() does it, where it always was)
    if constexpr (PACK) o[0] = bsx_chain_time_fraction<PACK>(resets ? 0 : t, a.L, sink);
    if (resets) {                                               // :91-97
      bsx_draws d;
      bsx_draws_begin<MT>(&d, a.ctl, i, lane, step);
      ctx = 0;
      if (MT == 0 || d.mt == nullptr) {                         // BernVec(nb) = the low nb bits of ceil(nb/32) words
        ctx = (uint64_t)bsx_word(&d);
        if (a.nb > 32) ctx |= (uint64_t)bsx_word(&d) << 32;
        ctx &= (1ull << a.nb) - 1ull;                           // nb <= 62
      } else {
        uint32_t w = 0;
        for (int b = 0; b < a.nb; ++b) ctx |= (uint64_t)bsx_bern_vec_bit(&d, b, &w) << b;   // one legacy double per bit
      }
      query = (int)bsx_randint(&d, (uint32_t)a.nb);
      bsx_draws_end<MT>(&d, a.ctl, i);
      t = 0;
      a.context[i] = ctx;
      a.state[i] = t | (query << 20);
      observe<PACK>(a, o, t, query, ctx, sink);
      return BSX_FIRST;
    }
    observe<PACK>(a, o, t, query, ctx, sink);                   // :74 — before the increment
    t += 1;                                                     // :75
    if (t - 1 < a.L) { a.state[i] = t | (query << 20); return BSX_MID; }   // :77-79
    // (the episode's one bsuite_info update: a no-return atomic when episodes are long, i.e. when only a few lanes of a
    // wave end on a given call — bsx_info_add)
    const bool quiet = a.L >= 8 && bsx_info_quiet<LOG>(a.ctl);
    if ((BSX_EARLY_ACTION ? act : bsx_action(a.ctl, a.action, oi, step)) == (int)((ctx >> query) & 1ull)) { reward = 1.0; bsx_info_add(quiet, &a.info[i], 1.0); }   // :83-85
    else { reward = -1.0; bsx_info_add(quiet, &a.info[a.ctl.n_lanes + i], 2.0); }   // :86-88
    a.state[i] = t | (query << 20) | MC_RESET_BIT;
    return BSX_LAST;
  }
};

// ------------------------------------------------------------------------------ umbrella_chain
#define UC_RESET_BIT (1 << 22)
struct umbrella_chain_env {
  static constexpr bool HAS_REGS = false, PACKED = true;
  struct regs { int unused; };
  struct args {
    bsx_ctl ctl; const int32_t* action; int32_t* state; bsx_timestep_t out; double* info;
    int32_t obs_numel; int32_t L; int32_t nd; uint32_t numel_magic;
    uint32_t* rows; int64_t row_plane_words;                   // bsx_call_t.row_scratch (bsx_rows.h) + words per plane, or nullptr
  };
  // Packed rows: HEAD = [need, has, time]; element 3+b is distractor bit b as 0.0 / 1.0 (one plane; umbrella_rows).
  __host__ __device__ static bool tf_table_fits(const args& a) { return a.L <= 1023; }
  typedef umbrella_rows rows_t;
  static constexpr int HEAD = rows_t::HEAD, PLANES = rows_t::PLANES;
  __device__ static float decode(uint32_t bit, uint32_t) { return rows_t::decode(bit, 0u); }
  template <bool PACK, int MT, class Sink>
  __device__ static void observe(const args& a, float* o, int t, int need, int has, bsx_draws* d, const Sink* sink) {
    BSX_NO_CONTRACT
    o[0] = (float)need;                                         // umbrella_chain.py:62
    o[1] = (float)has;                                          // :63
    // (o[2], the time fraction of :64, is step()'s: bsx_chain_time_fraction)
    uint32_t w = 0;
    if constexpr (PACK) {
      if (MT == 0 || d->mt == nullptr) {
        // :65 BernVec(nd) IS a run of stream words (bit i of the vector = bit i%32 of word i/32): hand the words to
        // the tile as they come — a bit-by-bit loop cost ~30 scalar + ~5 vector instructions per distractor
        // (3400 SALU per wave at nd = 100, profiles/r03/umbrella_distract_before_pmc_sq.json)
        for (int k = 0; 32 * k < a.nd; ++k) {
          const int n = a.nd - 32 * k;
          sink->put(0, k, bsx_word(d), n < 32 ? n : 32);
        }
      } else {
        uint32_t acc = 0;
        for (int b = 0; b < a.nd; ++b) {                        // MT19937-exact mode: one legacy double per bit
          acc |= bsx_bern_vec_bit(d, b, &w) << (b & 31);
          if ((b & 31) == 31 || b == a.nd - 1) { sink->put(0, b >> 5, acc, (b & 31) + 1); acc = 0; }
        }
      }
    } else {
      for (int b = 0; b < a.nd; ++b) o[3 + b] = (float)bsx_bern_vec_bit(d, b, &w);   // :65 BernVec(nd)
    }
  }
  template <int LOG, int MT, bool PACK = false, class Sink = bsx_bit_sink>
  __device__ static int step(const args& a, int64_t i, int64_t oi, uint64_t lane, uint64_t step, float* o, double& reward,
                             const Sink* sink = nullptr) {
    BSX_NO_CONTRACT
    int32_t st = a.state[i];
    // (the action matters on the episode's first step only, but which lanes are there is known when the state word has
    // arrived: loaded now, beside it, not in a second dependent round trip — in any real batch every wave holds such a
    // lane, and the line is fetched for it anyway)
    const int act = BSX_EARLY_ACTION && !a.ctl.force_reset ? bsx_action(a.ctl, a.action, oi, step) : 0;
    int t = st & 0xFFFFF, need = (st >> 20) & 1, has = (st >> 21) & 1;
    bsx_draws d;
    bsx_draws_begin<MT>(&d, a.ctl, i, lane, step);
    // (every path below draws from block 0 of the lane's stream: computed once, before the lanes of a wave — at different
    // episode phases in any real batch — part ways; 659 -> see profiles/r05/ab_umbrella_shared_philox_block.log)
    bsx_draws_prime(&d);
    const bool resets = a.ctl.force_reset || (st & UC_RESET_BIT);
    o[2] = bsx_chain_time_fraction<PACK>(resets ? 0 : t + 1, a.L, sink);   // :64 — of the state AFTER the increment (:69)
    if (resets) {                                               // :87-92
      t = 0;
      need = (int)bsx_bern(&d);
      has = (int)bsx_bern(&d);
      observe<PACK, MT>(a, o, t, need, has, &d, sink);
      bsx_draws_end<MT>(&d, a.ctl, i);
      a.state[i] = t | (need << 20) | (has << 21);
      return BSX_FIRST;
    }
    t += 1;                                                     // :69
    if (t == 1) has = ((BSX_EARLY_ACTION ? act : bsx_action(a.ctl, a.action, oi, step)) == 1);   // :71-72 (action_spec: {0,1})
    int type;
    if (t == a.L) {                                             // :74-81
      if (has == need) reward = 1.0;
      else { reward = -1.0; a.info[i] += 2.0; }
      observe<PACK, MT>(a, o, t, need, has, &d, sink);
      type = BSX_LAST;
    } else {                                                    // :83-85
      reward = 2.0 * (double)bsx_bern(&d) - 1.0;
      observe<PACK, MT>(a, o, t, need, has, &d, sink);
      type = BSX_MID;
    }
    bsx_draws_end<MT>(&d, a.ctl, i);
    a.state[i] = t | (need << 20) | (has << 21) | (type == BSX_LAST ? UC_RESET_BIT : 0);
    return type;
  }
};

// ------------------------------------------------------------------------------ discounting_chain
#define DC_RESET_BIT (1 << 12)
struct discounting_chain_env : small_regs_defaults {
  static constexpr bool HAS_REGS = true, PACKED = false;        // (register-resident in a fused rollout, like the bandit)
  static constexpr int EAGER_LPT_MIN_BLOCKS = 4096;              // (two lanes per thread: equal at 2^19 lanes)
  __host__ __device__ static constexpr int numel_of(int) { return 2; }
  struct regs { int32_t st; };
  struct args {
    bsx_ctl ctl; const int32_t* action; int32_t* state; bsx_timestep_t out;
    int32_t obs_numel; int32_t bonus;
  };
  static int variant_of(const args&) { return 0; }
  template <bool NOFORCE = false>
  __device__ static __forceinline__ bool wants_reset(const args&, const regs&) { return false; }
  // state word: timestep (bits 0-7) | context + 6 (bits 8-11: the context is the episode's first action, -5..4, -1 after a
  // reset — the reference indexes Python lists with it, so -5..-1 are legal and wrap, discounting_chain.py:76-81) | reset_next
  __device__ static __forceinline__ int dc_context(int32_t st) { return ((st >> 8) & 0xF) - 6; }
  __device__ static __forceinline__ int32_t dc_pack(int t, int ctx, bool last) {
    return t | ((ctx + 6) << 8) | (last ? DC_RESET_BIT : 0);
  }
  __device__ static __forceinline__ void clear(regs& r) { r.st = dc_pack(0, -1, false); }
  __device__ static __forceinline__ bool reset_pending(const regs& r) { return (r.st & DC_RESET_BIT) != 0; }
  __device__ static __forceinline__ void reset_part(const args&, uint64_t, uint64_t, int, unsigned, bsx_reset_pool*) {}
  __host__ __device__ static bool table_fits(const args&) { return false; }
  static size_t table_bytes(const args&) { return 0; }
  __device__ static __forceinline__ bsx_lds_table stage_tables(const args&, float*) { return (bsx_lds_table)0; }
  template <int V = -1>
  __device__ static __forceinline__ void load_info(const args&, int64_t, regs&) {}
  template <int V = -1>
  __device__ static __forceinline__ void store_info(const args&, int64_t, const regs&) {}
  __device__ static __forceinline__ void load(const args& a, int64_t i, regs& r) { r.st = a.state[i]; }
  __device__ static __forceinline__ void store(const args& a, int64_t i, const regs& r) { a.state[i] = r.st; }
  template <int LOG, int MT, bool IREGS = false, bool TAB = false, bool POOL = false, int V = -1, bool NOFORCE = false>
  __device__ static __forceinline__ int core(const args& a, regs& rg, const int act, int64_t i, uint64_t, uint64_t,
                                             float* o, double& reward, bsx_lds_table = (bsx_lds_table)0,
                                             const bsx_reset_pool* = nullptr) {
    BSX_NO_CONTRACT
    const int32_t st = rg.st;
    int t = st & 0xFF, ctx = dc_context(st);
    if ((!NOFORCE && a.ctl.force_reset) || (st & DC_RESET_BIT)) {   // discounting_chain.py:69-73
      o[0] = -1.0f; o[1] = 0.0f;
      rg.st = dc_pack(0, -1, false);
      return BSX_FIRST;
    }
    if (t == 0) {                                               // :76-77
      ctx = act;
      if (ctx < -5 || ctx > 4) {                                // reference: IndexError at the lookup of :80
        bsx_note_invalid_action(a.ctl, i);
        ctx = ctx < 0 ? 0 : 4;
      }
    }
    t += 1;
    const int chain = ctx < 0 ? ctx + 5 : ctx;                  // :80-81 index Python lists: -5..-1 wrap, the context stays negative
    const int when = chain == 0 ? 1 : chain == 1 ? 3 : chain == 2 ? 10 : chain == 3 ? 30 : 100;   // :49
    if (t == when) reward = (chain == a.bonus) ? 1.0 + 0.1 : 1.0;                          // :57-58,80-83
    o[0] = (float)ctx;                                          // :65
    o[1] = (float)((double)t / 100.0);                          // :66
    const int type = (t == 100) ? BSX_LAST : BSX_MID;           // :86-88
    rg.st = dc_pack(t, ctx, type == BSX_LAST);
    return type;
  }
  template <int LOG, int MT>
  __device__ static int step(const args& a, int64_t i, int64_t oi, uint64_t, uint64_t step, float* o, double& reward) {
    BSX_NO_CONTRACT
    int32_t st = a.state[i];
    int t = st & 0xFF, ctx = dc_context(st);
    if (a.ctl.force_reset || (st & DC_RESET_BIT)) {             // discounting_chain.py:69-73
      t = 0; ctx = -1;
      o[0] = -1.0f; o[1] = 0.0f;
      a.state[i] = dc_pack(0, -1, false);
      return BSX_FIRST;
    }
    if (t == 0) {                                               // :76-77
      ctx = bsx_action(a.ctl, a.action, oi, step);
      if (ctx < -5 || ctx > 4) {                                // reference: IndexError at the lookup of :80
        bsx_note_invalid_action(a.ctl, i);
        ctx = ctx < 0 ? 0 : 4;                                  // action_spec: 5 values; never OOB
      }
    }
    t += 1;
    const int chain = ctx < 0 ? ctx + 5 : ctx;                  // :80-81 index Python lists: -5..-1 wrap, the context stays negative
    const int when = chain == 0 ? 1 : chain == 1 ? 3 : chain == 2 ? 10 : chain == 3 ? 30 : 100;   // :49
    if (t == when) reward = (chain == a.bonus) ? 1.0 + 0.1 : 1.0;                          // :57-58,80-83
    o[0] = (float)ctx;                                          // :65
    o[1] = (float)((double)t / 100.0);                          // :66
    const int type = (t == 100) ? BSX_LAST : BSX_MID;           // :86-88
    a.state[i] = dc_pack(t, ctx, type == BSX_LAST);
    return type;
  }
};

// ------------------------------------------------------------------------------ cartpole / swingup
#define CP_RESET_BIT (1 << 30)
// Info columns (f64 [4,B]): 0 raw_return, 1 best_episode, 2 episode_return, 3 total_upright.
// Classic cartpole pays r in {0, 1}: an episode of k steps returns (k-1) + [last step rewarded], so
// raw_return / best_episode / episode_return are EXACT integer-valued functions of the step counter and
// are folded into the f64 columns only when the episode ends (column 0 then holds finished episodes;
// the host adds the running episode's k, environments/cartpole.py).  That removes two f64
// read-modify-writes (32 B) per lane per step — a third of the step's HBM traffic.  Swing-up's
// rewards (-0.1*|a-1| + 1) do not sum exactly out of order and the fused Logging rows snapshot the
// columns mid-episode, so swing-up and logging runs keep the reference's per-step accumulation.
struct cartpole_env {
  struct args {
    bsx_ctl ctl; const int32_t* action; float* state; int32_t* steps; bsx_timestep_t out;
    double* info; int32_t obs_numel; bsx_cartpole_t cfg;
    // derived on the host in f64, rounded once (cartpole_make)
    float inv_m_total, pole_ml, pole_ml_over_mt, den_a, den_b, inv_x_threshold;
  };
  // The lane's state in registers: step() = load + core + store; the fused rollout loads once, runs core
  // T times and stores once (small_obs_body), instead of a round trip through L2 every step.
  static constexpr bool HAS_REGS = true, PACKED = false, POOLED_RESETS = true, ROWS_VIA_LDS = true;
  static constexpr int EAGER_LPT_MIN_BLOCKS = 0, EAGER_LPT = 2;      // (equal within noise at 2^20 lanes, 4 % slower at 2^19)
  // compile-time variants of the lean fused rollout (small_obs_regs_rollout, V): 0 = classic, 1 = swing-up
  static constexpr int N_VARIANTS = 2;
  __host__ __device__ static constexpr int numel_of(int v) { return v == 1 ? 8 : 6; }
  static int variant_of(const args& a) { return a.cfg.swingup ? 1 : 0; }
  struct regs { float x, xd, th, thd; int32_t sk; double inf[4]; };     // inf: the info columns in a fused rollout
  // NOFORCE: inside a rollout (n_steps > 1 excludes force_reset: bsx_check_call)
  template <bool NOFORCE = false>
  __device__ static __forceinline__ bool wants_reset(const args& a, const regs& r) { return (!NOFORCE && a.ctl.force_reset) || (r.sk & CP_RESET_BIT); }
  __device__ static __forceinline__ void clear(regs& r) { r.sk = 0; }
  __device__ static __forceinline__ bool reset_pending(const regs& r) { return (r.sk & CP_RESET_BIT) != 0; }
  // Half of a lane's reset (cartpole.py:118-128), counter-based stream only: part 0 = x, x_dot from words 0..3 of
  // the (lane, step) stream, part 1 = theta, theta_dot from words 4..7 and the new angle's sine / cosine — the same
  // words, the same arithmetic as core()'s in-line reset, one Philox block per part.
  // (Tried: these parameters in LDS, read where they are used, instead of ten scalar registers live through the
  // whole step loop — no spill reload left in any variant's loop, but either the Philox key schedule moves to the
  // vector unit (80 VGPRs) or, with the words read back into scalar registers, the allocator still ends at 68-71
  // VGPRs instead of 61-65: one reload per step is the cheaper price.)
  __device__ static __forceinline__ void reset_part(const args& a, uint64_t lane, uint64_t step, int part, unsigned owner,
                                                    bsx_reset_pool* pool) {
    BSX_NO_CONTRACT
    const bsx_cartpole_t& g = a.cfg;
    bsx_draws d;
    bsx_draws_init(&d, a.ctl.seed, lane, step, BSX_STREAM_ENV);
    d.next = 4u * (uint32_t)part;
    const double lo = -g.init_range, hi = g.init_range;
    const double w0 = lo + (hi - lo) * bsx_uniform(&d);
    const double w1 = lo + (hi - lo) * bsx_uniform(&d);
    const float v0 = (float)(part ? g.theta_offset + w0 : w0), v1 = (float)w1;
    float si, co;
    bsx_sincosf(v0, &si, &co);
    pool->vals[2 * part][owner] = v0;
    pool->vals[2 * part + 1][owner] = v1;
    if (part) { pool->vals[4][owner] = si; pool->vals[5][owner] = co; }
  }
  // Fused rollouts keep the time-fraction table in LDS when it is small (the default 1002 entries: 4 KiB)
  static constexpr int TABLE_MAX_BYTES = 16384;
  __host__ __device__ static bool table_fits(const args& a) { return ((int64_t)a.cfg.last_step + 1) * 4 <= TABLE_MAX_BYTES; }
  static size_t table_bytes(const args& a) { return table_fits(a) ? ((size_t)a.cfg.last_step + 1) * 4 : 0; }
  __device__ static __forceinline__ bsx_lds_table stage_tables(const args& a, float* s_dyn) {
    const int n = a.cfg.last_step + 1;
    for (int k = threadIdx.x; k < n; k += BSX_BLOCK) s_dyn[k] = a.cfg.time_frac[k];
    return (bsx_lds_table)s_dyn;
  }
  template <int V = -1>
  __device__ static __forceinline__ void load_info(const args& a, int64_t i, regs& r) {
    const int64_t B = a.ctl.n_lanes;
    r.inf[0] = a.info[i]; r.inf[1] = a.info[B + i];
    if (V >= 0 ? V == 1 : (bool)a.cfg.swingup) { r.inf[2] = a.info[2 * B + i]; r.inf[3] = a.info[3 * B + i]; }
    else { r.inf[2] = 0.0; r.inf[3] = 0.0; }
  }
  template <int V = -1>
  __device__ static __forceinline__ void store_info(const args& a, int64_t i, const regs& r) {
    const int64_t B = a.ctl.n_lanes;
    a.info[i] = r.inf[0]; a.info[B + i] = r.inf[1];
    if (V >= 0 ? V == 1 : (bool)a.cfg.swingup) { a.info[2 * B + i] = r.inf[2]; a.info[3 * B + i] = r.inf[3]; }
  }
  __device__ static __forceinline__ void load(const args& a, int64_t i, regs& r) {
    const int64_t B = a.ctl.n_lanes;
    r.sk = a.steps[i];
    r.x = a.state[i]; r.xd = a.state[B + i]; r.th = a.state[2 * B + i]; r.thd = a.state[3 * B + i];
  }
  __device__ static __forceinline__ void store(const args& a, int64_t i, const regs& r) {
    const int64_t B = a.ctl.n_lanes;
    a.state[i] = r.x; a.state[B + i] = r.xd; a.state[2 * B + i] = r.th; a.state[3 * B + i] = r.thd;
    a.steps[i] = r.sk;
  }
  template <int LOG, int MT>
  __device__ static int step(const args& a, int64_t i, int64_t oi, uint64_t lane, uint64_t step, float* o, double& reward) {
    regs r;
    load(a, i, r);
    const int act = a.ctl.force_reset ? 0 : bsx_action(a.ctl, a.action, oi, step);
    const int type = core<LOG, MT>(a, r, act, i, lane, step, o, reward);
    store(a, i, r);
    return type;
  }
  // IREGS: the info columns are rg.inf[] (fused rollout without Logging), else read-modify-written in HBM.
  // s_tf: the time-fraction table in LDS, or nullptr (-> g.time_frac in device memory).
  // POOL: the reset values were computed by the workgroup's pool (reset_part) and wait in s_pool.
  // V: -1 = swing-up or not is a.cfg.swingup, 0 / 1 = known at compile time.  NOFORCE: see wants_reset.
  template <int LOG, int MT, bool IREGS = false, bool TAB = false, bool POOL = false, int V = -1, bool NOFORCE = false>
  __device__ static __forceinline__ int core(const args& a, regs& rg, const int act, int64_t i, uint64_t lane, uint64_t step,
                                             float* o, double& reward, bsx_lds_table s_tf = (bsx_lds_table)0,
                                             const bsx_reset_pool* s_pool = nullptr) {
    BSX_NO_CONTRACT
    const int64_t B = a.ctl.n_lanes;
    const bsx_cartpole_t& g = a.cfg;
    const bool swingup = V >= 0 ? V == 1 : (bool)g.swingup;
    auto info_get = [&](int col) -> double { if constexpr (IREGS) return rg.inf[col]; else return a.info[(int64_t)col * B + i]; };
    auto info_set = [&](int col, double v) { if constexpr (IREGS) rg.inf[col] = v; else a.info[(int64_t)col * B + i] = v; };
    const int32_t sk = rg.sk;
    const bool per_step_info = swingup || LOG == 1 || (LOG == -1 && a.ctl.log.steps != nullptr);
    int k = sk & 0x3FFFFFFF;
    float x, xd, th, thd, si, co;
    int type;
    if ((!NOFORCE && a.ctl.force_reset) || (sk & CP_RESET_BIT)) {   // cartpole.py:118-128 / swingup:81-91
      // (Tried, not adopted — profiles/r03/ab_regs_rollout_scalar_reset_draws.log: walking the wave's few resetting
      // lanes one at a time with wave-uniform inputs puts the Philox rounds on the scalar unit and cuts the vector
      // instructions by 19 %, but the ~200-instruction dependent scalar chain per resetting lane stalls the wave
      // longer than the divergent branch did: fused rollout 12.5 -> 14.7 us per step.)
      if constexpr (POOL) {
        const unsigned me = threadIdx.x;
        x = s_pool->vals[0][me]; xd = s_pool->vals[1][me]; th = s_pool->vals[2][me]; thd = s_pool->vals[3][me];
        si = s_pool->vals[4][me]; co = s_pool->vals[5][me];
      } else {
        bsx_draws d;
        bsx_draws_begin<MT>(&d, a.ctl, i, lane, step);
        const double lo = -g.init_range, hi = g.init_range;
        x = (float)(lo + (hi - lo) * bsx_uniform(&d));
        xd = (float)(lo + (hi - lo) * bsx_uniform(&d));
        th = (float)(g.theta_offset + (lo + (hi - lo) * bsx_uniform(&d)));
        thd = (float)(lo + (hi - lo) * bsx_uniform(&d));
        bsx_draws_end<MT>(&d, a.ctl, i);
        bsx_sincosf(th, &si, &co);                             // |theta_offset| + init_range <= 32 (cartpole_make)
      }
      // an explicit reset() in mid-episode abandons it: the k rewards of 1 it has paid stay in raw_return
      if (!per_step_info && !(sk & CP_RESET_BIT) && k > 0) info_set(0, info_get(0) + (double)k);
      k = 0;
      if (per_step_info) info_set(2, 0.0);                      // _episode_return = 0
      type = BSX_FIRST;
    } else {
      x = rg.x; xd = rg.xd; th = rg.th; thd = rg.thd;
      // step_cartpole, cartpole.py:37-65, in f32.  One sine/cosine pair per step: that of the OLD
      // angle; the new angle's pair follows from it by the angle-addition formulas below.
      float s0, c0;
      bsx_sincosf(th, &s0, &c0);                                // th is in [0, 2*pi) or a reset value
      const float force = (float)(act - 1) * g.force_mag;
      const float temp = (force + a.pole_ml * (thd * thd) * s0) * a.inv_m_total;
      // theta_acc = (g sin - cos*temp) / (l (4/3 - m_p cos^2 / m_t)); v_rcp_f32 is 1 ulp and the
      // accelerations enter the state scaled by dt = 0.01
      const float theta_acc = (g.gravity * s0 - c0 * temp) * __builtin_amdgcn_rcpf(a.den_a - a.den_b * (c0 * c0));
      const float x_acc = temp - a.pole_ml_over_mt * theta_acc * c0;
      const float dth = g.timescale * thd;
      x = __builtin_fmaf(g.timescale, xd, x);
      xd = __builtin_fmaf(g.timescale, x_acc, xd);
      // np.remainder(theta + dt*theta_dot, 2*pi) in f64 (the period is not the f32 2*pi): one
      // conditional +-2*pi is exact (Sterbenz) whenever the sum is within one period of [0, 2*pi)
      const double raw_ang = (double)th + (double)g.timescale * (double)thd;
      double ang = raw_ang >= 6.283185307179586 ? raw_ang - 6.283185307179586       // selects, not branches
                   : (raw_ang < 0.0 ? raw_ang + 6.283185307179586 : raw_ang);
      if (!(ang >= 0.0 && ang < 6.283185307179586)) {           // |dt*theta_dot| > 2*pi (theta_dot > 600 rad/s:
        ang = (double)th + (double)g.timescale * (double)thd;   // only reachable from a loaded state)
        ang -= 6.283185307179586 * floor(ang / 6.283185307179586);
        if (!(ang >= 0.0 && ang < 6.283185307179586)) ang = 0.0;
      }
      th = (float)ang;
      thd = __builtin_fmaf(g.timescale, theta_acc, thd);
      if (fabsf(dth) <= 0.5f) bsx_sincos_advance(s0, c0, dth, &si, &co);
      else bsx_sincosf(th, &si, &co);                           // th is in [0, 2*pi) here
      k += 1;                                                   // time_elapsed += timescale (:63)
      const bool timeout = k >= g.last_step;                    // time_elapsed > max_time
      bool end;
      double r;
      if (!swingup) {                                           // cartpole.py:142-153
        const bool ok = (co > g.height_threshold) && (fabsf(x) < g.x_threshold);
        r = ok ? 1.0 : 0.0;
        end = timeout || !ok;
      } else {                                                  // swingup:104-123
        const bool up = (co > g.height_threshold) && (fabsf(thd) < g.theta_dot_threshold) &&
                        (fabsf(x) < g.x_reward_threshold);
        r = -1.0 * fabs((double)(act - 1)) * g.move_cost;
        if (up) { r += 1.0; info_set(3, info_get(3) + 1.0); }
        end = timeout || (fabsf(x) > g.x_threshold);
      }
      reward = r;
      type = end ? BSX_LAST : BSX_MID;
      if (per_step_info) {
        info_set(0, info_get(0) + r);                           // _raw_return
        const double ep = info_get(2) + r;                      // _episode_return
        info_set(2, ep);
        if (end) {
          const double best = info_get(1);
          info_set(1, ep > best ? ep : best);                   // max(episode_return, best_episode)
        }
      } else if (end) {
        const double ep = (double)(k - 1) + r;                  // sum of the episode's rewards, exact
        info_set(0, info_get(0) + ep);
        const double best = info_get(1);
        info_set(1, ep > best ? ep : best);
      }
    }
    rg.x = x; rg.xd = xd; rg.th = th; rg.thd = thd;
    rg.sk = k | (type == BSX_LAST ? CP_RESET_BIT : 0);
    o[0] = x * a.inv_x_threshold;                               // cartpole.py:171-176
    o[1] = xd * a.inv_x_threshold;
    o[2] = si;
    o[3] = co;
    o[4] = thd;
    const int kf = k < g.last_step ? k : g.last_step;
    if constexpr (TAB) o[5] = s_tf[kf];                         // the fused rollout's LDS copy
    else o[5] = g.time_frac[kf];
    if (swingup) {                                              // swingup:147-149
      o[6] = (fabsf(x) < g.x_reward_threshold) ? 1.0f : -1.0f;
      o[7] = (fabsf(thd) < g.theta_dot_threshold) ? 1.0f : -1.0f;
    }
    return type;
  }
};

// ------------------------------------------------------------------------------ mountain_car
// Info column 0 = raw_return = -(steps taken): every step pays -1 (mountain_car.py:75-76), so the
// column is folded at episode ends (+= -t, exact) and the host subtracts the running episode's t;
// under the fused Logging wrapper (rows snapshot the column mid-episode) it is kept per step.
struct mountain_car_env {
  struct args {
    bsx_ctl ctl; const int32_t* action; float* state; int32_t* steps; bsx_timestep_t out;
    double* info; int32_t obs_numel; int32_t max_steps;
  };
  // (resets are rare — 1000-step episodes — and one Philox block: not pooled; the 12-byte rows of a wave are one dense
  // 768-byte range already: staging them gained nothing, profiles/r03/ab_rows_via_lds.log)
  static constexpr bool HAS_REGS = true, PACKED = false, POOLED_RESETS = false, ROWS_VIA_LDS = false;
  static constexpr int EAGER_LPT_MIN_BLOCKS = 2048, EAGER_LPT = 2;
  static constexpr int N_VARIANTS = 1;
  __host__ __device__ static constexpr int numel_of(int) { return 3; }
  static int variant_of(const args&) { return 0; }
  struct regs { float pos, vel; int32_t sk; double inf0; };             // inf0: raw_return in a fused rollout
  template <bool NOFORCE = false>
  __device__ static __forceinline__ bool wants_reset(const args&, const regs&) { return false; }
  __device__ static __forceinline__ void clear(regs& r) { r.sk = 0; }
  __device__ static __forceinline__ bool reset_pending(const regs& r) { return (r.sk & CP_RESET_BIT) != 0; }
  __device__ static __forceinline__ void reset_part(const args&, uint64_t, uint64_t, int, unsigned, bsx_reset_pool*) {}
  // Fused rollouts read the time fraction t / max_steps (an f32 division: 11 of the step's ~95 vector instructions)
  // from a table in LDS that the workgroup fills once per launch with that same division.
  static constexpr int TABLE_MAX_BYTES = 16384;
  __host__ __device__ static bool table_fits(const args& a) { return ((int64_t)a.max_steps + 1) * 4 <= TABLE_MAX_BYTES; }
  static size_t table_bytes(const args& a) { return table_fits(a) ? ((size_t)a.max_steps + 1) * 4 : 0; }
  __device__ static __forceinline__ bsx_lds_table stage_tables(const args& a, float* s_dyn) {
    for (int k = threadIdx.x; k <= a.max_steps; k += BSX_BLOCK) s_dyn[k] = (float)k / (float)a.max_steps;
    return (bsx_lds_table)s_dyn;
  }
  template <int V = -1>
  __device__ static __forceinline__ void load_info(const args& a, int64_t i, regs& r) { r.inf0 = a.info[i]; }
  template <int V = -1>
  __device__ static __forceinline__ void store_info(const args& a, int64_t i, const regs& r) { a.info[i] = r.inf0; }
  __device__ static __forceinline__ void load(const args& a, int64_t i, regs& r) {
    r.sk = a.steps[i]; r.pos = a.state[i]; r.vel = a.state[a.ctl.n_lanes + i];
  }
  __device__ static __forceinline__ void store(const args& a, int64_t i, const regs& r) {
    a.state[i] = r.pos; a.state[a.ctl.n_lanes + i] = r.vel; a.steps[i] = r.sk;
  }
  template <int LOG, int MT>
  __device__ static int step(const args& a, int64_t i, int64_t oi, uint64_t lane, uint64_t step, float* o, double& reward) {
    regs r;
    load(a, i, r);
    const int act = a.ctl.force_reset ? 0 : bsx_action(a.ctl, a.action, oi, step);
    const int type = core<LOG, MT>(a, r, act, i, lane, step, o, reward);
    store(a, i, r);
    return type;
  }
  template <int LOG, int MT, bool IREGS = false, bool TAB = false, bool POOL = false, int V = -1, bool NOFORCE = false>
  __device__ static __forceinline__ int core(const args& a, regs& rg, const int act, int64_t i, uint64_t lane, uint64_t step,
                                             float* o, double& reward, bsx_lds_table s_tf = (bsx_lds_table)0,
                                             const bsx_reset_pool* = nullptr) {
    BSX_NO_CONTRACT
    const int32_t sk = rg.sk;
    int t = sk & 0x3FFFFFFF;
    auto info_add = [&](double v) { if constexpr (IREGS) rg.inf0 += v; else a.info[i] += v; };
    float pos, vel;
    int type;
    if ((!NOFORCE && a.ctl.force_reset) || (sk & CP_RESET_BIT)) {   // mountain_car.py:66-71
      bsx_draws d;
      bsx_draws_begin<MT>(&d, a.ctl, i, lane, step);
      // an explicit reset() in mid-episode abandons it: its t rewards of -1 stay in raw_return
      if (!(LOG == 1 || (LOG == -1 && a.ctl.log.steps != nullptr)) && !(sk & CP_RESET_BIT) && t > 0) info_add(-(double)t);
      t = 0;
      pos = (float)(-0.6 + (-0.4 - -0.6) * bsx_uniform(&d));
      bsx_draws_end<MT>(&d, a.ctl, i);
      vel = 0.0f;
      type = BSX_FIRST;
    } else {
      pos = rg.pos; vel = rg.vel;
      t += 1;                                                   // :74
      reward = -1.0;
      float sn, cs;
      bsx_sincosf(3.0f * pos, &sn, &cs);                        // position is clipped to [-1.2, 0.6]
      vel += (float)(act - 1) * 0.001f + cs * -0.0025f;            // :79-80
      vel = fminf(fmaxf(vel, -0.07f), 0.07f);                   // :81
      pos += vel;                                               // :82
      pos = fminf(fmaxf(pos, -1.2f), 0.6f);                     // :83
      if (pos == -1.2f) vel = fminf(fmaxf(vel, 0.0f), 0.07f);   // :84-85
      type = (pos >= 0.5f || t >= a.max_steps) ? BSX_LAST : BSX_MID;   // :88-90
      if (LOG == 1 || (LOG == -1 && a.ctl.log.steps != nullptr)) info_add(reward);   // :76, per step under Logging
      else if (type == BSX_LAST) info_add(-(double)t);          // the episode's t rewards of -1, exact
    }
    rg.pos = pos; rg.vel = vel;
    rg.sk = t | (type == BSX_LAST ? CP_RESET_BIT : 0);
    o[0] = pos;                                                 // :62-64
    o[1] = vel;
    if constexpr (TAB) o[2] = s_tf[t];                          // (t <= max_steps)
    else o[2] = (float)t / (float)a.max_steps;                  // both exact in f32; correctly rounded quotient
    return type;
  }
};
// (with its 12-byte rows non-temporal, mountain_car's rollout is faster with ORDINARY scalar stores: 4.45 against 4.7 us per step)
template <> struct small_rollout_nt_scalars<mountain_car_env> { static constexpr bool value = false; };

#endif  // BSX_SMALL_OBS_H_
