// small_obs.hip — C-ABI entry points of the small-observation families (device code: small_obs.h):
//   bandit            bsuite/environments/bandit.py:54-64
//   memory_chain      bsuite/environments/memory_chain.py:60-97
//   umbrella_chain    bsuite/environments/umbrella_chain.py:60-92
//   discounting_chain bsuite/environments/discounting_chain.py:63-88
//   cartpole/swingup  bsuite/environments/cartpole.py:37-177,
//                     bsuite/experiments/cartpole_swingup/cartpole_swingup.py:81-150
//   mountain_car      bsuite/environments/mountain_car.py:62-90
// each with the auto-reset of bsuite/environments/base.py:54-65.
#include "small_obs.h"

// Tile class of a segment inside a grouped launch (segments of one group must share it).  Always 256 since
// the bit-plane tiles: the 64-lane class for rows wider than 32 floats is gone (kept in the ABI so that a
// caller that buckets segments by class keeps working).
extern "C" int bsx_group_small_class(int32_t numel) { (void)numel; return BSX_BLOCK; }

// Bytes of bsx_call_t.row_scratch for n_lanes lanes (bsx_rows.h): 0 = no row path for this family / row length.
extern "C" int64_t bsx_row_scratch_bytes(int32_t family, int32_t obs_numel, int64_t n_lanes) {
  if (obs_numel < 1 || obs_numel > 256 || n_lanes < 1 || n_lanes > ((int64_t)1 << 40) || bsx_small_direct_shape(obs_numel)) return 0;
  if (family == BSX_FAM_MEMORY_CHAIN) return (int64_t)(4 * bsx_rows_scratch_words(BSX_ROWS_MEMORY, n_lanes, obs_numel));
  if (family == BSX_FAM_UMBRELLA_CHAIN) return (int64_t)(4 * bsx_rows_scratch_words(BSX_ROWS_UMBRELLA, n_lanes, obs_numel));
  return 0;
}

// The row path of a chain segment: the call's scratch, if it brings one and the row is wide.
template <class Env>
static int chain_rows(const bsx_call_t* call, int32_t family, typename Env::args* a) {
  a->rows = nullptr; a->row_plane_words = 0;
  if (call->row_scratch == nullptr || call->n_lanes < 1 || bsx_row_scratch_bytes(family, a->obs_numel, call->n_lanes) == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(call->row_scratch) & 15u) != 0) return BSX_EALIGN;
  a->rows = (uint32_t*)call->row_scratch;
  a->row_plane_words = (int64_t)bsx_rows_plane_words(call->n_lanes, a->obs_numel);
  return 0;
}

// ------------------------------------------------------------------------------ bandit
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

// ------------------------------------------------------------------------------ memory_chain
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

// ------------------------------------------------------------------------------ umbrella_chain
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

// ------------------------------------------------------------------------------ discounting_chain
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

// ------------------------------------------------------------------------------ cartpole / swingup
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

// ------------------------------------------------------------------------------ mountain_car
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

