// deep_sea.hip — batched DeepSea for gfx950.  Replaces bsuite/environments/deep_sea.py:103-144
// (`_get_observation`, `_reset`, `_step`) plus the auto-reset of bsuite/environments/base.py:54-65.
//
// Shape of the work at the headline config (N=30, B=2^20): 21 B of scalar traffic and 3600 B of
// observation stores per lane per call — a pure store stream.  Two launches per call (default):
//   advance  deep_sea_step_kernel<256,1,false>: the N*N-bit action mapping goes kernarg -> LDS once
//            per block (lane-divergent lookup); every thread advances one lane (coalesced column
//            loads/stores of action / packed state / reward / discount / step_type), LAST/FIRST
//            masks by wavefront ballot;
//   observe  bsx_hot_stream_kernel<deep_sea_hot,4,256>: a pure store stream over [B x N*N] f32 —
//            block b writes floats [b*4096,(b+1)*4096) as 4 lane-interleaved 16-byte stores per
//            thread, hot cells recomputed from the packed state column (4 B/lane, L2-resident).
//            Measured 6.2-6.3 TB/s at N=30, B=2^20 against 5.3-5.5 TB/s for the fused per-block
//            tile writer (profiles/r01/sweep_stream_k.log, ab_fused_vs_split*.log).
// BSX_DS_SPLIT=0 selects the fused single-kernel variant (phase 1 on LPB lanes, then all 256
// threads stream the block's contiguous [LPB x N*N] tile, bsx_write_hot_tile), kept for A/B.
#include "bsx_host.h"

struct deep_sea_args {
  bsx_ctl ctl;
  const int32_t* action;
  int32_t* state;
  bsx_timestep_t out;
  double* info;        // [2,B]: total_bad_episodes, denoised_return
  double move_cost;
  double inv_size;
  int32_t size;
  int32_t deterministic;
  uint32_t cells;
  uint32_t cells_magic;
  uint32_t mapping_bits[BSX_DEEP_SEA_MAX_SIZE * BSX_DEEP_SEA_MAX_SIZE / 32];
};

#define DS_RESET_BIT (1 << 17)
#define DS_BAD_BIT (1 << 16)

// hot cell of a lane from its packed state (split-phase writer)
struct deep_sea_hot {
  int N;
  __device__ __forceinline__ void operator()(int32_t st, int& a, int& b) const {
    const int row = st & 0xFF, col = (st >> 8) & 0xFF;
    a = row < N ? row * N + col : -1;     // deep_sea.py:105-107
    b = -1;
  }
};

template <int LPB, int UNROLL, int FUSED>
__global__ void __launch_bounds__(BSX_BLOCK) deep_sea_step_kernel(const deep_sea_args a) {
  __shared__ uint32_t s_map[BSX_DEEP_SEA_MAX_SIZE * BSX_DEEP_SEA_MAX_SIZE / 32];
  __shared__ int s_hot[LPB];
  __shared__ unsigned int s_cnt[2];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;

  const int N = a.size;
  const int map_words = (N * N + 31) >> 5;
  for (int w = threadIdx.x; w < map_words; w += BSX_BLOCK) s_map[w] = a.mapping_bits[w];
  __syncthreads();

  const int64_t lane0 = (int64_t)blockIdx.x * LPB;
  int64_t remaining = a.ctl.n_lanes - lane0;
  const int lanes_here = remaining < LPB ? (int)remaining : LPB;

  if (threadIdx.x < LPB) {
    const int64_t i = lane0 + threadIdx.x;
    int type = -1;
    if (threadIdx.x < lanes_here) {
      BSX_NO_CONTRACT
      const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;
      const uint64_t step = bsx_step_of(a.ctl);
      int32_t st = a.state[i];
      int row = st & 0xFF, col = (st >> 8) & 0xFF, bad = (st >> 16) & 1;
      double reward = 0.0;
      if (a.ctl.force_reset || (st & DS_RESET_BIT)) {          // base.py:61-62 -> deep_sea.py:110-114
        row = 0; col = 0; bad = 0;
        type = BSX_FIRST;
      } else {
        const int act = a.action[i];
        const int cell = row * N + col;
        const int mapped = (int)((s_map[cell >> 5] >> (cell & 31)) & 1u);
        const bool right = (act == mapped);                     // deep_sea.py:118
        bsx_draws d;
        bsx_draws_init(&d, a.ctl.seed, lane, step, BSX_STREAM_ENV);
        if (col == N - 1 && right) {                            // :121-123
          reward += 1.0;
          a.info[a.ctl.n_lanes + i] += 1.0;
        }
        if (!a.deterministic && row == N - 1 && (col == 0 || col == N - 1))   // :124-126
          reward += bsx_normal(&d);
        if (right) {                                            // :129-132
          const double u = bsx_uniform(&d);                     // drawn even when deterministic
          if (u > a.inv_size || a.deterministic) col = col + 1 > N - 1 ? N - 1 : col + 1;
          reward -= a.move_cost;
        } else {                                                // :133-136
          if (row == col) bad = 1;
          col = col - 1 < 0 ? 0 : col - 1;
        }
        row += 1;                                               // :137
        if (row == N) {                                         // :140-143
          if (bad) a.info[i] += 1.0;
          type = BSX_LAST;
        } else {
          type = BSX_MID;
        }
      }
      a.state[i] = row | (col << 8) | (bad << 16) | (type == BSX_LAST ? DS_RESET_BIT : 0);
      bsx_emit(a.ctl, a.out, i, lane, step, type, reward);
      s_hot[threadIdx.x] = (row < N) ? row * N + col : -1;      // :105-107 (terminal obs all-zero)
    }
    bsx_count_types(a.ctl, type, s_cnt);
  }
  __syncthreads();
  bsx_flush_counts(a.ctl, s_cnt);
  if (FUSED != 1) return;

  bsx_write_hot_tile<false, UNROLL>(a.out.observation + lane0 * (int64_t)a.cells, lanes_here, a.cells,
                            a.cells_magic, s_hot, nullptr);
}

extern "C" int bsx_deep_sea_step(const bsx_deep_sea_t* cfg, const bsx_call_t* call,
                                 const int32_t* action, int32_t* state, bsx_timestep_t out,
                                 double* info) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->size < 1 || cfg->size > BSX_DEEP_SEA_MAX_SIZE) return BSX_ERANGE;
  if (call->n_lanes == 0) return 0;
  if (state == nullptr || info == nullptr) return BSX_ENULL;

  deep_sea_args a;
  a.ctl = bsx_make_ctl(call);
  a.action = action; a.state = state; a.out = out; a.info = info;
  a.move_cost = cfg->move_cost; a.inv_size = cfg->inv_size;
  a.size = cfg->size; a.deterministic = cfg->deterministic;
  a.cells = (uint32_t)(cfg->size * cfg->size);
  a.cells_magic = bsx_div_magic(a.cells);
  for (int w = 0; w < BSX_DEEP_SEA_MAX_SIZE * BSX_DEEP_SEA_MAX_SIZE / 32; ++w) a.mapping_bits[w] = cfg->mapping_bits[w];

  hipStream_t st = (hipStream_t)call->hip_stream;
  // Tile = LPB lanes: 64 lanes x 3600 B = 230 KB per block at N=30 -> 16384 blocks at B=2^20,
  // enough to keep 256 CUs x 8 resident blocks busy with a short tail.  BSX_DS_LPB / BSX_DS_UNROLL
  // are tuning knobs for the A/B sweeps recorded under profiles/ (defaults = measured best).
  static const int split = bsx_env_int("BSX_DS_SPLIT", 1);
  if (split) {
    // advance kernel (all 256 threads own a lane) + pure streaming observation writer
    const int64_t blocks_a = (call->n_lanes + 255) / 256;
    if (blocks_a > 0x7FFFFFFF) return BSX_EINVAL;
    deep_sea_step_kernel<256, 1, 0><<<dim3((unsigned)blocks_a), dim3(BSX_BLOCK), 0, st>>>(a);
    deep_sea_hot fn{cfg->size};
    rc = bsx_launch_hot_stream(out.observation, state, call->n_lanes, a.cells, a.cells_magic, fn, st, 4);
    if (rc != 0) return rc;
    return bsx_launch_status();
  }
  static const int lpb = bsx_env_int("BSX_DS_LPB", 64);
  static const int unroll = bsx_env_int("BSX_DS_UNROLL", 4);
#define DS_LAUNCH(L, U)                                                                     \
  do {                                                                                      \
    const int64_t blocks = (call->n_lanes + (L) - 1) / (L);                                 \
    if (blocks > 0x7FFFFFFF) return BSX_EINVAL;                                             \
    deep_sea_step_kernel<L, U, 1><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(a);      \
  } while (0)
#define DS_UNROLLS(L)                                              \
  do {                                                             \
    if (unroll >= 8) DS_LAUNCH(L, 8);                              \
    else if (unroll >= 4) DS_LAUNCH(L, 4);                         \
    else if (unroll >= 2) DS_LAUNCH(L, 2);                         \
    else DS_LAUNCH(L, 1);                                          \
  } while (0)
  if (lpb >= 256) DS_UNROLLS(256);
  else if (lpb >= 128) DS_UNROLLS(128);
  else if (lpb >= 64) DS_UNROLLS(64);
  else if (lpb >= 32) DS_UNROLLS(32);
  else DS_UNROLLS(16);
  return bsx_launch_status();
}
