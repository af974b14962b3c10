// memory_chain.hip — C-ABI entry points of memory_chain (bsuite/environments/memory_chain.py:60-97; auto-reset of bsuite/environments/base.py:54-65).
// Device code: small_obs.h (memory_chain_env).  One translation unit per small-observation family: the families' kernels are independent
// template instantiations, and compiling them side by side is what keeps a clean build() under a minute (round 6; as ONE
// file they were a 56 s single-threaded compile, the long pole of every build).
#include "small_obs.h"

#include "chain_rows.h"

static int memory_chain_make(const bsx_memory_chain_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state, uint64_t* context, bsx_timestep_t out, double* info, memory_chain_env::args* a) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->memory_length < 1 || cfg->memory_length > 1000000 || cfg->num_bits < 1 || cfg->num_bits > 62)
    return BSX_ERANGE;
  if (call->n_lanes > 0 && (state == nullptr || context == nullptr || info == nullptr)) return BSX_ENULL;
  a->ctl = bsx_make_ctl(call); a->action = action; a->state = state; a->context = context; a->out = out;
  a->info = info; a->obs_numel = cfg->num_bits + 2; a->L = cfg->memory_length; a->nb = cfg->num_bits;
  a->numel_magic = bsx_div_magic((uint32_t)a->obs_numel);
  return chain_rows<memory_chain_env>(call, BSX_FAM_MEMORY_CHAIN, a);
}

extern "C" int bsx_memory_chain_step(const bsx_memory_chain_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state, uint64_t* context, bsx_timestep_t out, double* info) {
  memory_chain_env::args a;
  int rc = memory_chain_make(cfg, call, action, state, context, out, info, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  return launch_small_obs<memory_chain_env>(a, bsx_n_steps(call), call->hip_stream);
}

extern "C" int bsx_group_set_memory_chain(bsx_group_t* g, int32_t index, const bsx_memory_chain_t* cfg, const bsx_call_t* call,
                                     const int32_t* action, int32_t* state, uint64_t* context, bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  memory_chain_env::args a;
  int rc = memory_chain_make(cfg, call, action, state, context, out, info, &a);
  if (rc != 0) return rc;
  return small_obs_group_put<memory_chain_env>(g, BSX_FAM_MEMORY_CHAIN, index, call, a);
}
