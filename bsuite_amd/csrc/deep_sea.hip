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

// ---------------------------------------------------------------------------------------------
// ONE launch per step (ABI v11, bsx_call_t.flags & BSX_CALL_STATE_TAGGED; deterministic, no wrapper, no Logging).
// The observation store stream, unchanged in shape — workgroup b writes floats [b*K*1024, (b+1)*K*1024) as K
// lane-interleaved 16-byte stores per thread — but nobody has advanced the lanes beforehand: every thread loads the
// packed state AND the action of the lane whose row its chunk belongs to and recomputes that lane's transition itself
// (deterministic deep_sea: ~25 vector instructions and one look-up in the action mapping, which every WAVE keeps as its
// own copy in LDS — no workgroup barrier anywhere).  The thread whose chunk holds the first float of a lane's row is
// that lane's WRITER: it alone stores the new state word, reward / discount / step_type and the bsuite_info updates.
// A reader may find the word already advanced by its writer; bit 18 (the parity of the next call's index, written by
// every advance: deep_sea_fam.h) says which, and both cases end in the same new state.  No second launch, no launch gap,
// no state column round trip: the 12.5 us lane advance of the two-launch step and the gap behind it are gone.
// (The same for catch would put a Philox block — its reset draw — into most waves of the stream: r01 measured 100 vs
// 42 us.  Stochastic deep_sea draws per step and keeps the two launches.)
// (cells: a multiple of 4 — a 16-byte chunk then never straddles two lanes — and at least 768; other boards keep the two launches)
template <int K>
__global__ void __launch_bounds__(BSX_BLOCK) deep_sea_step1_kernel(const deep_sea_fam::args a, const uint32_t cells,
                                                                   const uint32_t cells_magic, const bsx_div64 dv) {
  __shared__ deep_sea_fam::shared s_map[BSX_BLOCK / BSX_WAVE];           // one copy per wave: in-order LDS, no s_barrier
  const int wl = (int)(threadIdx.x & 63u);
  deep_sea_fam::shared& map = s_map[threadIdx.x >> 6];
  const int64_t n_lanes = a.ctl.n_lanes;
  const uint64_t step = bsx_step_of(a.ctl);
  const int32_t tag_new = (int32_t)(((uint32_t)(step + 1) & 1u) << DS_TAG_SHIFT);
  const bool forced = a.ctl.force_reset != 0;
  const deep_sea_hot fn{a.size};
  const uint64_t total = (uint64_t)n_lanes * cells;
  const uint64_t F0 = (uint64_t)blockIdx.x * (uint64_t)(K * 4 * BSX_BLOCK);
  const uint64_t lane_b = __umul64hi(F0, dv.m) >> dv.s;                  // uniform: the lane of the workgroup's first float
  const uint32_t r_b = (uint32_t)(F0 - lane_b * cells);
  bsx_f4* __restrict__ o4 = reinterpret_cast<bsx_f4*>(a.out.observation + F0);
  // everything per lane is addressed {uniform pointer at lane_b} + {32-bit lane offset}
  const int32_t* __restrict__ st_b = a.state + lane_b;
  const int32_t* __restrict__ act_b = a.action + (forced ? 0 : lane_b);

  // The K chunks of a thread lie 256 floats apart (each wave owns K consecutive KiB), the first and the last one
  // (K - 1) * 256 = 768 floats: with rows of at least that many floats (the launcher's condition: N >= 28) they touch
  // at most TWO lanes, the first chunk's and the last chunk's — two transitions per thread, not one per chunk.
  uint32_t dl[K];
  int r0[K];
  bool live[K];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const uint32_t c = (threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63);
    const uint32_t f = r_b + (c << 2);
    dl[u] = __umulhi(f, cells_magic);
    r0[u] = (int)(f - dl[u] * cells);
    live[u] = F0 + ((uint64_t)c << 2) + 3 < total;
  }
  // lanes A = the first chunk's, B = the last live chunk's (B == A or A + 1 when cells >= (K - 1) * 256)
  const uint32_t dA = dl[0];
  uint32_t dB = dA;
  bool wrA = false, wrB = false;                                         // is some chunk of mine the first of lane A's / B's row?
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (live[u]) dB = dl[u];
  }
#pragma unroll
  for (int u = 0; u < K; ++u) {
    wrA |= live[u] && r0[u] == 0 && dl[u] == dA;
    wrB |= live[u] && r0[u] == 0 && dl[u] == dB && dB != dA;
  }
  const bool any = live[0];
  const int32_t wA = any ? st_b[dA] : 0, wB = any ? st_b[dB] : 0;
  const int aA = (any && !forced) ? act_b[dA] : 0, aB = (any && !forced) ? act_b[dB] : 0;

  // (the wave's copy of the action mapping: its loads go out BEHIND the state / action loads above — one round trip for all)
  const int map_words = (a.size * a.size + 31) >> 5;                    // <= DS_MAP_WORDS = 128: two words per lane at most
  static_assert(DS_MAP_WORDS <= 2 * BSX_WAVE, "the per-wave copy of the action mapping is two passes at most");
  if (wl < map_words) map.map[wl] = a.mapping_bits[wl];
  if (wl + BSX_WAVE < map_words) map.map[wl + BSX_WAVE] = a.mapping_bits[wl + BSX_WAVE];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");


  // The new packed state of lane lane_b + d given its state word w and action; `writer` publishes it.
  auto resolve = [&](uint32_t d, int32_t w, int act, bool writer, int& type) -> int32_t {
    type = -1;
    if (!writer && (w & DS_TAG_BIT) == tag_new) return w;                // the lane's writer has been here already
    const int64_t L = (int64_t)lane_b + d;
    int32_t nst; double reward;
    const int ty = deep_sea_fam::advance<true, 1>(a, map, L, a.ctl.lane_offset + (uint64_t)L, step, w, act, nst, reward, writer);
    if (writer) {
      type = ty;
      a.state[L] = nst;
      bsx_emit_at<0, 0, false>(a.ctl, a.out, L, L, a.ctl.lane_offset + (uint64_t)L, step, ty, reward);
    }
    return nst;
  };
  int tA = -1, tB = -1;
  int32_t nA = 0, nB = 0;
  if (any) {
    nA = resolve(dA, wA, aA, wrA, tA);
    nB = dB != dA ? resolve(dB, wB, aB, wrB, tB) : nA;
  }
  int hA, hB, unused;
  fn(nA, hA, unused);
  fn(nB, hB, unused);
#pragma unroll
  for (int u = 0; u < K; ++u) {
    if (!live[u]) continue;
    int hot = dl[u] == dA ? hA : hB;
    const int p0 = hot < 0 ? -1 : hot - r0[u];
    bsx_f4 v;
    v.x = p0 == 0 ? 1.0f : 0.0f;
    v.y = p0 == 1 ? 1.0f : 0.0f;
    v.z = p0 == 2 ? 1.0f : 0.0f;
    v.w = p0 == 3 ? 1.0f : 0.0f;
    // (non-temporal here: 79 -> 83.5 us per step at 2^17 lanes, 157 -> 166 at 2^18: profiles/r06/ab_deep_sea_step1_nt.log)
    o4[(threadIdx.x >> 6) * (K * 64) + u * 64 + (threadIdx.x & 63)] = v;
  }
  const unsigned int n_last = (unsigned int)__popcll(__ballot(tA == BSX_LAST)) + (unsigned int)__popcll(__ballot(tB == BSX_LAST));
  const unsigned int n_first = (unsigned int)__popcll(__ballot(tA == BSX_FIRST)) + (unsigned int)__popcll(__ballot(tB == BSX_FIRST));
  // one pair of sharded atomics per WAVE that saw an episode end or begin (a 1/31 of the lanes each on a steady-state call)
  if (a.ctl.counters != nullptr && wl == 0 && (n_last | n_first) != 0u) {
    unsigned long long* shard = (unsigned long long*)a.ctl.counters +
                                (size_t)((blockIdx.x * (BSX_BLOCK / BSX_WAVE) + (threadIdx.x >> 6)) & (BSX_COUNTER_SHARDS - 1)) * BSX_COUNTER_STRIDE;
    if (n_last) atomicAdd(&shard[0], (unsigned long long)n_last);
    if (n_first) atomicAdd(&shard[1], (unsigned long long)n_first);
  }
}

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
  // ONE launch (deep_sea_step1_kernel) where the caller vouches for the state words' parity tags and nothing but the
  // deterministic transition itself has to be recomputed per thread; small boards in small batches keep their fused
  // tile step (bsx_pair_call), which is one launch already.
  // ... up to 2^18 lanes of N = 30 (BSX_DEEP_SEA_STEP1_MAX_MIB of observations per step): there the 12.5 us lane advance and
  // the launch gap it removes outweigh the transitions recomputed in every wave of the stream; at 2^20 lanes the
  // recomputation costs the stream 25-50 us (597 -> 622-648 us, profiles/r04/ab_deep_sea_single_launch.log) and the
  // two launches stay.  (The kernel reads `action` as a plain [B] column: a call that walks an action ring takes the
  // two-launch path, whose lane advance goes through bsx_action().)
  static const int step1_env = bsx_env_int("BSX_DEEP_SEA_STEP1", 1);
  static const int step1_max_mib = bsx_env_int("BSX_DEEP_SEA_STEP1_MAX_MIB", 1024);
  if (step1_env != 0 && (int64_t)call->n_lanes * cells * 4 <= ((int64_t)step1_max_mib << 20) && (call->flags & BSX_CALL_STATE_TAGGED) && call->n_steps <= 1 && call->action_ring <= 1 && cfg->deterministic && bsx_ctl_lean(a.ctl) &&
      call->obs_paint == nullptr && (cells & 3u) == 0 && cells >= 3u * 256u) {
    constexpr int K = 4;
    const uint64_t total = (uint64_t)call->n_lanes * cells;
    const uint64_t blocks = (total + (uint64_t)K * 4 * BSX_BLOCK - 1) / ((uint64_t)K * 4 * BSX_BLOCK);
    if (blocks > 0x7FFFFFFFull) return BSX_EINVAL;
    deep_sea_step1_kernel<K><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, (hipStream_t)call->hip_stream>>>(
        a, cells, bsx_div_magic(cells), bsx_make_div64(cells));
    return bsx_launch_status();
  }
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
                              bsx_flat_blocks((uint64_t)a.ctl.n_lanes * cells, PAIR_DEEP_SEA_K), 0);
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
