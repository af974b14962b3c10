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

#define CATCH_RESET_BIT (1 << 24)

struct catch_fam {
  struct args {
    bsx_ctl ctl;
    const int32_t* action;
    int32_t* state;
    bsx_timestep_t out;
    double* info;        // [1,B]: total_regret
    int32_t rows, columns;
  };
  struct shared { int unused; };
  __device__ static __forceinline__ void stage(const args&, shared&) {}

  __device__ static __forceinline__ int advance(const args& a, const shared&, int64_t i, uint64_t lane,
                                                uint64_t step, int32_t st, int act, int32_t& nst,
                                                double& reward) {
    const int rows = a.rows, cols = a.columns;
    int ball_x = st & 0xFF, ball_y = (st >> 8) & 0xFF, paddle_x = (st >> 16) & 0xFF;
    int type;
    reward = 0.0;
    if (a.ctl.force_reset || (st & CATCH_RESET_BIT)) {         // catch.py:80-81 -> :68-76
      bsx_draws d;
      bsx_draws_begin(&d, a.ctl, i, lane, step);
      ball_x = (int)bsx_randint(&d, (uint32_t)cols);            // :71
      bsx_draws_end(&d, a.ctl, i);
      ball_y = 0;
      paddle_x = cols / 2;
      type = BSX_FIRST;
    } else {
      if (act < 0 || act > 2) bsx_note_invalid_action(a.ctl, i);   // reference: IndexError (catch.py:84)
      const int dx = act - 1;                                   // _ACTIONS :27
      paddle_x = paddle_x + dx;                                 // :85 np.clip
      paddle_x = paddle_x < 0 ? 0 : (paddle_x > cols - 1 ? cols - 1 : paddle_x);
      ball_y += 1;                                              // :88
      if (ball_y == rows - 1) {                                 // :91-95
        reward = (paddle_x == ball_x) ? 1.0 : -1.0;
        a.info[i] += (1.0 - reward);
        type = BSX_LAST;
      } else {
        type = BSX_MID;                                         // :97
      }
    }
    nst = ball_x | (ball_y << 8) | (paddle_x << 16) | (type == BSX_LAST ? CATCH_RESET_BIT : 0);
    return type;
  }
};

struct catch_hot {
  int rows, cols;
  __device__ __forceinline__ void operator()(int32_t st, int& a, int& b) const {
    a = ((st >> 8) & 0xFF) * cols + (st & 0xFF);          // ball   (catch.py:111)
    b = (rows - 1) * cols + ((st >> 16) & 0xFF);          // paddle (catch.py:112)
  }
};

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
  return 0;
}

extern "C" int bsx_catch_step(const bsx_catch_t* cfg, const bsx_call_t* call, const int32_t* action,
                              int32_t* state, bsx_timestep_t out, double* info) {
  catch_fam::args a;
  int rc = catch_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  const uint32_t cells = (uint32_t)(cfg->rows * cfg->columns);
  hipStream_t st = (hipStream_t)call->hip_stream;
  catch_hot fn{cfg->rows, cfg->columns};
  const int n_steps = bsx_n_steps(call);
  for (int t = 0; t < n_steps; ++t) {       // rollout: the kernel pair once per step, outputs [T,B,...]
    const int64_t off = (int64_t)t * call->n_lanes;
    a.ctl.step_index = call->stream.step_index + (uint64_t)t;
    a.ctl.reward_f64 = call->reward_f64 ? call->reward_f64 + off : nullptr;
    a.action = action ? action + off : action;
    a.out.reward = out.reward + off; a.out.discount = out.discount + off; a.out.step_type = out.step_type + off;
    a.out.observation = out.observation + off * (int64_t)cells;
    if (call->obs_paint != nullptr) {           // delta mode: advance + in-place patch in one launch
      rc = bsx_launch_advance_delta<catch_fam, catch_hot>(a, fn, call->obs_paint, cells, st);
    } else {
      rc = bsx_launch_advance<catch_fam>(a, st);
      if (rc != 0) return rc;
      // stores/thread x 256 threads: a sharp optimum per family (profiles/r01/sweep_stream_*.log)
      rc = bsx_launch_hot_stream(a.out.observation, state, call->n_lanes, cells, bsx_div_magic(cells), fn, st, 2);
    }
    if (rc != 0) return rc;
  }
  return bsx_launch_status();
}

extern "C" int bsx_group_set_catch(bsx_group_t* g, int32_t index, const bsx_catch_t* cfg, const bsx_call_t* call,
                                   const int32_t* action, int32_t* state, bsx_timestep_t out, double* info) {
  int rc = bsx_group_check_set(g, BSX_FAM_CATCH, index, call, sizeof(catch_fam::args),
                               sizeof(bsx_stream_seg<catch_hot>), 0);
  if (rc != 0) return rc;
  catch_fam::args a;
  rc = catch_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  g->launch = bsx_group_launch_pair<catch_fam, catch_hot, 2>;
  g->n_phases = 2;
  return bsx_group_put_pair<catch_fam, catch_hot>(g, index, a, out.observation, state,
                                                  (uint32_t)(cfg->rows * cfg->columns),
                                                  catch_hot{cfg->rows, cfg->columns}, 2);
}
