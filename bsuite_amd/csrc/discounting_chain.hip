// discounting_chain.hip — C-ABI entry points of discounting_chain (bsuite/environments/discounting_chain.py:63-88; auto-reset of bsuite/environments/base.py:54-65).
// Device code: small_obs.h (discounting_chain_env).  One translation unit per small-observation family: the families' kernels are independent
// template instantiations, and compiling them side by side is what keeps a clean build() under a minute (round 6; as ONE
// file they were a 56 s single-threaded compile, the long pole of every build).
#include "small_obs.h"

static int discounting_chain_make(const bsx_discounting_chain_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state, bsx_timestep_t out, discounting_chain_env::args* a) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->bonus_chain < 0 || cfg->bonus_chain > 4) return BSX_ERANGE;
  if (call->n_lanes > 0 && state == nullptr) return BSX_ENULL;
  a->ctl = bsx_make_ctl(call); a->action = action; a->state = state; a->out = out;
  a->obs_numel = 2; a->bonus = cfg->bonus_chain;
  return 0;
}

extern "C" int bsx_discounting_chain_step(const bsx_discounting_chain_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state, bsx_timestep_t out) {
  discounting_chain_env::args a;
  int rc = discounting_chain_make(cfg, call, action, state, out, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  return launch_small_obs<discounting_chain_env>(a, bsx_n_steps(call), call->hip_stream);
}

extern "C" int bsx_group_set_discounting_chain(bsx_group_t* g, int32_t index, const bsx_discounting_chain_t* cfg, const bsx_call_t* call,
                                     const int32_t* action, int32_t* state, bsx_timestep_t out) {
  if (g == nullptr) return BSX_ENULL;
  discounting_chain_env::args a;
  int rc = discounting_chain_make(cfg, call, action, state, out, &a);
  if (rc != 0) return rc;
  return small_obs_group_put<discounting_chain_env>(g, BSX_FAM_DISCOUNTING_CHAIN, index, call, a);
}
