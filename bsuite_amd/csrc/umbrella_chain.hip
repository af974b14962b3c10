// umbrella_chain.hip — C-ABI entry points of umbrella_chain (bsuite/environments/umbrella_chain.py:60-92; auto-reset of bsuite/environments/base.py:54-65).
// Device code: small_obs.h (umbrella_chain_env).  One translation unit per small-observation family: the families' kernels are independent
// template instantiations, and compiling them side by side is what keeps a clean build() under a minute (round 6; as ONE
// file they were a 56 s single-threaded compile, the long pole of every build).
#include "small_obs.h"

#include "chain_rows.h"

static int umbrella_chain_make(const bsx_umbrella_chain_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state, bsx_timestep_t out, double* info, umbrella_chain_env::args* a) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->chain_length < 1 || cfg->chain_length > 1000000 || cfg->n_distractor < 0 || cfg->n_distractor > 253)
    return BSX_ERANGE;
  if (call->n_lanes > 0 && (state == nullptr || info == nullptr)) return BSX_ENULL;
  a->ctl = bsx_make_ctl(call); a->action = action; a->state = state; a->out = out; a->info = info;
  a->obs_numel = 3 + cfg->n_distractor; a->L = cfg->chain_length; a->nd = cfg->n_distractor;
  a->numel_magic = bsx_div_magic((uint32_t)a->obs_numel);
  return chain_rows<umbrella_chain_env>(call, BSX_FAM_UMBRELLA_CHAIN, a);
}

extern "C" int bsx_umbrella_chain_step(const bsx_umbrella_chain_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state, bsx_timestep_t out, double* info) {
  umbrella_chain_env::args a;
  int rc = umbrella_chain_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  return launch_small_obs<umbrella_chain_env>(a, bsx_n_steps(call), call->hip_stream);
}

extern "C" int bsx_group_set_umbrella_chain(bsx_group_t* g, int32_t index, const bsx_umbrella_chain_t* cfg, const bsx_call_t* call,
                                     const int32_t* action, int32_t* state, bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  umbrella_chain_env::args a;
  int rc = umbrella_chain_make(cfg, call, action, state, out, info, &a);
  if (rc != 0) return rc;
  return small_obs_group_put<umbrella_chain_env>(g, BSX_FAM_UMBRELLA_CHAIN, index, call, a);
}
