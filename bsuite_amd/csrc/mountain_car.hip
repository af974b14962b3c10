// mountain_car.hip — C-ABI entry points of mountain_car (bsuite/environments/mountain_car.py:62-90; auto-reset of bsuite/environments/base.py:54-65).
// Device code: small_obs.h (mountain_car_env).  One translation unit per small-observation family: the families' kernels are independent
// template instantiations, and compiling them side by side is what keeps a clean build() under a minute (round 6; as ONE
// file they were a 56 s single-threaded compile, the long pole of every build).
#include "small_obs.h"

static int mountain_car_make(const bsx_mountain_car_t* cfg, const bsx_call_t* call, const int32_t* action, float* state, int32_t* steps, bsx_timestep_t out, double* info, mountain_car_env::args* a) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->max_steps < 1 || cfg->max_steps >= (1 << 30)) return BSX_ERANGE;
  if (call->n_lanes > 0 && (state == nullptr || steps == nullptr || info == nullptr)) return BSX_ENULL;
  a->ctl = bsx_make_ctl(call); a->action = action; a->state = state; a->steps = steps; a->out = out;
  a->info = info; a->obs_numel = 3; a->max_steps = cfg->max_steps;
  return 0;
}

extern "C" int bsx_mountain_car_step(const bsx_mountain_car_t* cfg, const bsx_call_t* call, const int32_t* action, float* state, int32_t* steps, bsx_timestep_t out, double* info) {
  mountain_car_env::args a;
  int rc = mountain_car_make(cfg, call, action, state, steps, out, info, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  return launch_small_obs<mountain_car_env>(a, bsx_n_steps(call), call->hip_stream);
}

extern "C" int bsx_group_set_mountain_car(bsx_group_t* g, int32_t index, const bsx_mountain_car_t* cfg, const bsx_call_t* call,
                                     const int32_t* action, float* state, int32_t* steps, bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  mountain_car_env::args a;
  int rc = mountain_car_make(cfg, call, action, state, steps, out, info, &a);
  if (rc != 0) return rc;
  return small_obs_group_put<mountain_car_env>(g, BSX_FAM_MOUNTAIN_CAR, index, call, a);
}
