// catch.hip — batched Catch for gfx950.  Replaces bsuite/environments/catch.py:68-114
// (`_reset`, `_step`, `_observation`) including its private auto-reset (:80-81).
//
// 221 B per lane per call at 10x5: 21 B of scalar columns + a 200 B board with at most two ones.
// All 256 threads advance one lane each (coalesced column traffic), publish ball / paddle cell
// indices to LDS, then the block streams its contiguous [256 x rows*columns] f32 tile with 16-byte
// stores (two-hot variant of bsx_write_hot_tile).  The board itself is never materialised in LDS:
// only the two hot indices per lane are.
#include "bsx_host.h"

struct catch_args {
  bsx_ctl ctl;
  const int32_t* action;
  int32_t* state;
  bsx_timestep_t out;
  double* info;        // [1,B]: total_regret
  int32_t rows, columns;
  uint32_t cells, cells_magic;
};

#define CATCH_RESET_BIT (1 << 24)

struct catch_hot {
  int rows, cols;
  __device__ __forceinline__ void operator()(int32_t st, int& a, int& b) const {
    a = ((st >> 8) & 0xFF) * cols + (st & 0xFF);          // ball   (catch.py:111)
    b = (rows - 1) * cols + ((st >> 16) & 0xFF);          // paddle (catch.py:112)
  }
};

template <int LPB, bool FUSED>
__global__ void __launch_bounds__(BSX_BLOCK) catch_step_kernel(const catch_args a) {
  __shared__ int s_ball[LPB];
  __shared__ int s_paddle[LPB];
  __shared__ unsigned int s_cnt[2];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t lane0 = (int64_t)blockIdx.x * LPB;
  const int64_t remaining = a.ctl.n_lanes - lane0;
  const int lanes_here = remaining < LPB ? (int)remaining : LPB;
  const int rows = a.rows, cols = a.columns;

  if (threadIdx.x < LPB) {
    const int64_t i = lane0 + threadIdx.x;
    int type = -1;
    if (threadIdx.x < lanes_here) {
      const uint64_t lane = a.ctl.lane_offset + (uint64_t)i;
      const uint64_t step = bsx_step_of(a.ctl);
      int32_t st = a.state[i];
      int ball_x = st & 0xFF, ball_y = (st >> 8) & 0xFF, paddle_x = (st >> 16) & 0xFF;
      double reward = 0.0;
      if (a.ctl.force_reset || (st & CATCH_RESET_BIT)) {       // catch.py:80-81 -> :68-76
        bsx_draws d;
        bsx_draws_init(&d, a.ctl.seed, lane, step, BSX_STREAM_ENV);
        ball_x = (int)bsx_randint(&d, (uint32_t)cols);          // :71
        ball_y = 0;
        paddle_x = cols / 2;
        type = BSX_FIRST;
      } else {
        const int dx = a.action[i] - 1;                         // _ACTIONS :27
        paddle_x = paddle_x + dx;                               // :85 np.clip
        paddle_x = paddle_x < 0 ? 0 : (paddle_x > cols - 1 ? cols - 1 : paddle_x);
        ball_y += 1;                                            // :88
        if (ball_y == rows - 1) {                               // :91-95
          reward = (paddle_x == ball_x) ? 1.0 : -1.0;
          a.info[i] += (1.0 - reward);
          type = BSX_LAST;
        } else {
          type = BSX_MID;                                       // :97
        }
      }
      a.state[i] = ball_x | (ball_y << 8) | (paddle_x << 16) | (type == BSX_LAST ? CATCH_RESET_BIT : 0);
      bsx_emit(a.ctl, a.out, i, lane, step, type, reward);
      s_ball[threadIdx.x] = ball_y * cols + ball_x;             // :111
      s_paddle[threadIdx.x] = (rows - 1) * cols + paddle_x;     // :112
    }
    bsx_count_types(a.ctl, type, s_cnt);
  }
  __syncthreads();
  bsx_flush_counts(a.ctl, s_cnt);
  if (!FUSED) return;
  bsx_write_hot_tile<true, 4>(a.out.observation + lane0 * (int64_t)a.cells, lanes_here, a.cells,
                           a.cells_magic, s_ball, s_paddle);
}

extern "C" int bsx_catch_step(const bsx_catch_t* cfg, const bsx_call_t* call, const int32_t* action,
                              int32_t* state, bsx_timestep_t out, double* info) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->rows < 2 || cfg->rows > 64 || cfg->columns < 1 || cfg->columns > 64) return BSX_ERANGE;
  if (call->n_lanes == 0) return 0;
  if (state == nullptr || info == nullptr) return BSX_ENULL;
  catch_args a;
  a.ctl = bsx_make_ctl(call);
  a.action = action; a.state = state; a.out = out; a.info = info;
  a.rows = cfg->rows; a.columns = cfg->columns;
  a.cells = (uint32_t)(cfg->rows * cfg->columns);
  a.cells_magic = bsx_div_magic(a.cells);
  constexpr int LPB = 256;
  const int64_t blocks = (call->n_lanes + LPB - 1) / LPB;
  if (blocks > 0x7FFFFFFF) return BSX_EINVAL;
  hipStream_t st = (hipStream_t)call->hip_stream;
  static const int split = bsx_env_int("BSX_CATCH_SPLIT", 1);
  if (split) {
    catch_step_kernel<LPB, false><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(a);
    catch_hot fn{cfg->rows, cfg->columns};
    rc = bsx_launch_hot_stream(out.observation, state, call->n_lanes, a.cells, a.cells_magic, fn, st, 2);
    if (rc != 0) return rc;
    return bsx_launch_status();
  }
  catch_step_kernel<LPB, true><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>(a);
  return bsx_launch_status();
}
