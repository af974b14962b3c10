// cartpole.hip — C-ABI entry points of cartpole / swingup (bsuite/environments/cartpole.py:37-177, bsuite/experiments/cartpole_swingup/cartpole_swingup.py:81-150; auto-reset of bsuite/environments/base.py:54-65).
// Device code: small_obs.h (cartpole_env).  One translation unit per small-observation family: the families' kernels are independent
// template instantiations, and compiling them side by side is what keeps a clean build() under a minute (round 6; as ONE
// file they were a 56 s single-threaded compile, the long pole of every build).
#include "small_obs.h"

static int cartpole_make(const bsx_cartpole_t* cfg, const bsx_call_t* call, const int32_t* action, float* state, int32_t* steps, bsx_timestep_t out, double* info, cartpole_env::args* a) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->last_step < 1 || cfg->last_step >= (1 << 30)) return BSX_ERANGE;
  if (call->n_lanes > 0 && (state == nullptr || steps == nullptr || info == nullptr || cfg->time_frac == nullptr))
    return BSX_ENULL;
  a->ctl = bsx_make_ctl(call); a->action = action; a->state = state; a->steps = steps; a->out = out;
  a->info = info; a->obs_numel = cfg->swingup ? 8 : 6; a->cfg = *cfg;
  const double m_total = (double)cfg->mass_cart + (double)cfg->mass_pole;
  const double pole_ml = (double)cfg->mass_pole * (double)cfg->length;
  if (!(m_total > 0.0) || !(cfg->x_threshold > 0.0f) || !(cfg->length > 0.0f)) return BSX_ERANGE;
  // the kernel's sine/cosine is specified for |angle| <= BSX_SINCOS_MAX_ARG; angles live in [0, 2*pi) after
  // the first step, so only the reset value theta_offset + U(-init_range, init_range) needs the bound
  if (!(fabs(cfg->theta_offset) + fabs(cfg->init_range) <= 32.0)) return BSX_ERANGE;
  a->inv_m_total = (float)(1.0 / m_total);
  a->pole_ml = (float)pole_ml;
  a->pole_ml_over_mt = (float)(pole_ml / m_total);
  a->den_a = (float)((double)cfg->length * 4.0 / 3.0);                       // l * 4/3
  a->den_b = (float)((double)cfg->length * (double)cfg->mass_pole / m_total);  // l * m_p / m_t
  a->inv_x_threshold = (float)(1.0 / (double)cfg->x_threshold);
  return 0;
}

extern "C" int bsx_cartpole_step(const bsx_cartpole_t* cfg, const bsx_call_t* call, const int32_t* action, float* state, int32_t* steps, bsx_timestep_t out, double* info) {
  cartpole_env::args a;
  int rc = cartpole_make(cfg, call, action, state, steps, out, info, &a);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  return launch_small_obs<cartpole_env>(a, bsx_n_steps(call), call->hip_stream);
}

extern "C" int bsx_group_set_cartpole(bsx_group_t* g, int32_t index, const bsx_cartpole_t* cfg, const bsx_call_t* call,
                                     const int32_t* action, float* state, int32_t* steps, bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  cartpole_env::args a;
  int rc = cartpole_make(cfg, call, action, state, steps, out, info, &a);
  if (rc != 0) return rc;
  return small_obs_group_put<cartpole_env>(g, BSX_FAM_CARTPOLE, index, call, a);
}
