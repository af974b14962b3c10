// bandit.hip — C-ABI entry points of bandit (bsuite/environments/bandit.py:54-64; auto-reset of bsuite/environments/base.py:54-65).
// Device code: small_obs.h (bandit_env).  One translation unit per small-observation family: the families' kernels are independent
// template instantiations, and compiling them side by side is what keeps a clean build() under a minute (round 6; as ONE
// file they were a 56 s single-threaded compile, the long pole of every build).
#include "small_obs.h"

static int bandit_make(const bsx_bandit_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state, bsx_timestep_t out, double* info, bandit_env::args* a) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->num_actions < 1 || cfg->num_actions > BSX_BANDIT_MAX_ACTIONS) return BSX_ERANGE;
  if (call->n_lanes > 0 && (state == nullptr || info == nullptr)) return BSX_ENULL;
  a->ctl = bsx_make_ctl(call); a->action = action; a->state = state; a->out = out; a->info = info;
  a->obs_numel = 1; a->num_actions = cfg->num_actions;
  for (int k = 0; k < BSX_BANDIT_MAX_ACTIONS; ++k) a->rewards[k] = cfg->rewards[k];
  return 0;
}

extern "C" int bsx_bandit_step(const bsx_bandit_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state, bsx_timestep_t out, double* info) {
  bandit_env::args a;
  int rc = bandit_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  return launch_small_obs<bandit_env>(a, bsx_n_steps(call), call->hip_stream);
}

extern "C" int bsx_group_set_bandit(bsx_group_t* g, int32_t index, const bsx_bandit_t* cfg, const bsx_call_t* call,
                                     const int32_t* action, int32_t* state, bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  bandit_env::args a;
  int rc = bandit_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  return small_obs_group_put<bandit_env>(g, BSX_FAM_BANDIT, index, call, a);
}
