/* ORACLE — test infrastructure only.  NOT part of the product: nothing in bsuite_amd/ links,
 * loads or calls this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and only as the checker.
 *
 * Plain-C, scalar, batched restatement of the reference's per-environment step()/reset()
 * dynamics (google-deepmind/bsuite v0.3.6, paths relative to /root/reference):
 *   bsuite/environments/base.py:54-65            auto-reset protocol
 *   bsuite/environments/deep_sea.py:103-144      DeepSea
 *   bsuite/environments/catch.py:68-114          Catch
 *   bsuite/environments/bandit.py:54-64          SimpleBandit
 *   bsuite/environments/memory_chain.py:60-97    MemoryChain
 *   bsuite/environments/umbrella_chain.py:60-92  UmbrellaChain
 *   bsuite/environments/discounting_chain.py:63-88 DiscountingChain
 *   bsuite/environments/cartpole.py:37-177       step_cartpole + Cartpole
 *   bsuite/experiments/cartpole_swingup/cartpole_swingup.py:81-150  CartpoleSwingup
 *   bsuite/environments/mountain_car.py:62-90    MountainCar
 *   bsuite/environments/mnist.py:61-75           MNISTBandit
 *   bsuite/utils/wrappers.py:275-283,338-346     RewardNoise / RewardScale
 * All state and rewards are f64 as in the reference (Python floats); observations are cast to
 * f32 exactly where the reference casts (np.zeros(dtype=float32) assignment).
 *
 * Pin: every function here is checked against tests/golden/*.npz, which were produced by running
 * the unmodified reference with its RandomState replaced by a replay of the draw stream
 * (oracle/make_golden.py) — i.e. parity is pinned on outputs of the reference itself.  The
 * reference's own tests hold no value-level golden vectors for this path.
 *
 * The draw stream ("bsx stream v1", specified in include/bsx_stream.h) is restated here
 * independently of that header.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ------------------------------------------------------------------ draw stream restatement */
typedef struct {
  uint32_t ctr[4];
  uint32_t key[2];
  uint32_t out[4];
  int have_block;   /* block index cached in out[], -1 if none */
  uint32_t cursor;  /* next word */
} draws_t;

static void philox_block(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t x0 = ctr_in[0], x1 = ctr_in[1], x2 = ctr_in[2], x3 = ctr_in[3];
  uint32_t ka = key_in[0], kb = key_in[1];
  for (int round = 0; round < 10; round++) {
    uint64_t pa = (uint64_t)0xD2511F53u * x0;
    uint64_t pb = (uint64_t)0xCD9E8D57u * x2;
    uint32_t y0 = (uint32_t)(pb >> 32) ^ x1 ^ ka;
    uint32_t y1 = (uint32_t)pb;
    uint32_t y2 = (uint32_t)(pa >> 32) ^ x3 ^ kb;
    uint32_t y3 = (uint32_t)pa;
    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    ka += 0x9E3779B9u; kb += 0xBB67AE85u;
  }
  out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
}

static void draws_begin(draws_t* d, uint64_t seed, uint64_t lane, uint64_t step, uint32_t stream_id) {
  d->key[0] = (uint32_t)(seed & 0xFFFFFFFFu);
  d->key[1] = (uint32_t)(seed >> 32);
  d->ctr[0] = (uint32_t)(lane & 0xFFFFFFFFu);
  d->ctr[1] = (uint32_t)(lane >> 32);
  d->ctr[2] = (uint32_t)(step & 0xFFFFFFFFu);
  d->ctr[3] = (uint32_t)(((step >> 32) & 0xFFFFu) << 16) | ((stream_id & 0xFFu) << 8);
  d->have_block = -1;
  d->cursor = 0;
}

static uint32_t next_word(draws_t* d) {
  uint32_t w = d->cursor++;
  int blk = (int)(w / 4);
  if (blk != d->have_block) {
    uint32_t c[4] = {d->ctr[0], d->ctr[1], d->ctr[2], d->ctr[3] | (uint32_t)blk};
    philox_block(c, d->key, d->out);
    d->have_block = blk;
  }
  return d->out[w % 4];
}

static uint64_t next_k53(draws_t* d) {
  uint64_t a = next_word(d);
  uint64_t b = next_word(d);
  return ((a >> 5) << 26) | (b >> 6);
}
static double draw_uniform(draws_t* d) { return ldexp((double)next_k53(d), -53); }
static int draw_bern(draws_t* d) { return (int)(next_word(d) >> 31); }
static uint32_t draw_randint(draws_t* d, uint32_t n) { return (uint32_t)(((uint64_t)next_word(d) * n) >> 32); }

static double spec_log(double x) {
  int e;
  double m = frexp(x, &e); /* [0.5,1) */
  m *= 2.0; e -= 1;         /* [1,2)   */
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  double s = (m - 1.0) / (m + 1.0);
  double s2 = s * s;
  static const double odd[] = {25, 23, 21, 19, 17, 15, 13, 11, 9, 7, 5, 3};
  double p = 1.0 / odd[0];
  for (int i = 1; i < 12; i++) p = p * s2 + 1.0 / odd[i];
  p = p * s2 + 1.0;
  return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

static double horner8(const double* c, double r) {
  double acc = c[7];
  for (int i = 6; i >= 0; i--) acc = acc * r + c[i];
  return acc;
}

static double normal_from_k53(uint64_t k) {
  static const double A[8] = {3.3871328727963666080e+0, 1.3314166789178437745e+2, 1.9715909503065514427e+3, 1.3731693765509461125e+4, 4.5921953931549871457e+4, 6.7265770927008700853e+4, 3.3430575583588128105e+4, 2.5090809287301226727e+3};
  static const double B[8] = {1.0, 4.2313330701600911252e+1, 6.8718700749205790830e+2, 5.3941960214247511077e+3, 2.1213794301586595867e+4, 3.9307895800092710610e+4, 2.8729085735721942674e+4, 5.2264952788528545610e+3};
  static const double C[8] = {1.42343711074968357734e+0, 4.63033784615654529590e+0, 5.76949722146069140550e+0, 3.64784832476320460504e+0, 1.27045825245236838258e+0, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4};
  static const double D[8] = {1.0, 2.05319162663775882187e+0, 1.67638483018380384940e+0, 6.89767334985100004550e-1, 1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9};
  static const double E[8] = {6.65790464350110377720e+0, 5.46378491116411436990e+0, 1.78482653991729133580e+0, 2.96560571828504891230e-1, 2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7};
  static const double F[8] = {1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2, 7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15};
  int64_t j = 2 * ((int64_t)k - ((int64_t)1 << 52)) + 1;
  double q = ldexp((double)j, -54);
  double aq = fabs(q);
  if (aq <= 0.425) {
    double r = 0.180625 - q * q;
    return q * horner8(A, r) / horner8(B, r);
  }
  double r = sqrt(-spec_log(0.5 - aq));
  double v = (r <= 5.0) ? horner8(C, r - 1.6) / horner8(D, r - 1.6) : horner8(E, r - 5.0) / horner8(F, r - 5.0);
  return q < 0 ? -v : v;
}
static double draw_normal(draws_t* d) { return normal_from_k53(next_k53(d)); }

/* test hooks for the stream itself */
void orc_stream_words(uint64_t seed, uint64_t lane, uint64_t step, uint32_t stream_id, int n, uint32_t* out) {
  draws_t d; draws_begin(&d, seed, lane, step, stream_id);
  for (int i = 0; i < n; i++) out[i] = next_word(&d);
}
void orc_normals(const uint64_t* k, int64_t n, double* out) {
  for (int64_t i = 0; i < n; i++) out[i] = normal_from_k53(k[i]);
}

/* ------------------------------------------------------------------ call plumbing */
enum { FIRST = 0, MID = 1, LAST = 2 };
enum { WRAP_NONE = 0, WRAP_SCALE = 1, WRAP_NOISE = 2, WRAP_SCALE_NOISE = 3, WRAP_NOISE_SCALE = 4 };

typedef struct {
  int64_t n_lanes;
  const uint64_t* lane_ids;   /* global lane id per lane */
  uint64_t seed;
  uint64_t step;              /* index of this call */
  int32_t force_reset;        /* 1 => env.reset() on every lane */
  int32_t wrap_kind;
  double wrap_param;
  uint64_t wrap_seed;
  const int32_t* action;
  int8_t* step_type;
  double* reward;             /* NaN where the reference returns None */
  double* discount;           /* NaN where the reference returns None */
  float* obs;                 /* [n_lanes, obs_numel] */
  double wrap_param2;         /* stacked wrappers: the outer wrapper's parameter */
} orc_call;

static void emit(const orc_call* c, int64_t i, int type, double reward) {
  c->step_type[i] = (int8_t)type;
  if (type == FIRST) {                       /* dm_env.restart: reward=None, discount=None */
    c->reward[i] = NAN; c->discount[i] = NAN;
    return;
  }
  /* utils/wrappers.py:275-283 / :338-346 — applied by the outer wrapper to non-FIRST steps */
  if (c->wrap_kind == WRAP_NOISE) {
    draws_t w; draws_begin(&w, c->wrap_seed, c->lane_ids[i], c->step, 1);
    reward = reward + c->wrap_param * draw_normal(&w);
  } else if (c->wrap_kind == WRAP_SCALE) {
    reward = reward * c->wrap_param;
  } else if (c->wrap_kind == WRAP_SCALE_NOISE) {          /* RewardNoise(RewardScale(env)): inner first */
    draws_t w; draws_begin(&w, c->wrap_seed, c->lane_ids[i], c->step, 1);
    reward = reward * c->wrap_param;
    reward = reward + c->wrap_param2 * draw_normal(&w);
  } else if (c->wrap_kind == WRAP_NOISE_SCALE) {          /* RewardScale(RewardNoise(env)) */
    draws_t w; draws_begin(&w, c->wrap_seed, c->lane_ids[i], c->step, 1);
    reward = reward + c->wrap_param * draw_normal(&w);
    reward = reward * c->wrap_param2;
  }
  c->reward[i] = reward;
  c->discount[i] = (type == LAST) ? 0.0 : 1.0;   /* dm_env.termination / transition */
}

/* ------------------------------------------------------------------ deep_sea.py */
void orc_deep_sea(const orc_call* c, int N, int deterministic, double unscaled_move_cost,
                  const double* action_mapping /* [N*N], values as the reference holds them */,
                  int32_t* row, int32_t* col, int32_t* bad_episode, int32_t* reset_next,
                  double* total_bad_episodes, double* denoised_return) {
  for (int64_t i = 0; i < c->n_lanes; i++) {
    float* o = c->obs + i * (int64_t)N * N;
    memset(o, 0, sizeof(float) * N * N);                       /* :104 */
    if (c->force_reset || reset_next[i]) {                      /* base.py:61-62 / :54-57 */
      reset_next[i] = 0;
      row[i] = 0; col[i] = 0; bad_episode[i] = 0;               /* :111-113 */
      o[0] = 1.0f;
      emit(c, i, FIRST, 0.0);
      continue;
    }
    draws_t d; draws_begin(&d, c->seed, c->lane_ids[i], c->step, 0);
    double reward = 0.0;
    int a = c->action[i];
    int right = ((double)a == action_mapping[row[i] * N + col[i]]);   /* :118 */
    if (col[i] == N - 1 && right) { reward += 1.0; denoised_return[i] += 1.0; }   /* :121-123 */
    if (!deterministic) {                                             /* :124-126 */
      if (row[i] == N - 1 && (col[i] == 0 || col[i] == N - 1)) reward += draw_normal(&d);
    }
    if (right) {                                                      /* :129-132 */
      double u = draw_uniform(&d);          /* drawn even when deterministic (`or` order) */
      if (u > 1.0 / (double)N || deterministic) {
        int nc = col[i] + 1; if (nc > N - 1) nc = N - 1; col[i] = nc;
      }
      reward -= unscaled_move_cost / (double)N;
    } else {                                                          /* :133-136 */
      if (row[i] == col[i]) bad_episode[i] = 1;
      int nc = col[i] - 1; if (nc < 0) nc = 0; col[i] = nc;
    }
    row[i] += 1;                                                      /* :137 */
    if (row[i] < N) o[row[i] * N + col[i]] = 1.0f;                    /* :105-107 */
    if (row[i] == N) {                                                /* :140-143 */
      if (bad_episode[i]) total_bad_episodes[i] += 1.0;
      reset_next[i] = 1;
      emit(c, i, LAST, reward);
    } else {
      emit(c, i, MID, reward);
    }
  }
}

/* ------------------------------------------------------------------ catch.py */
void orc_catch(const orc_call* c, int rows, int columns, int32_t* ball_x, int32_t* ball_y,
               int32_t* paddle_x, int32_t* reset_next, double* total_regret) {
  const int paddle_y = rows - 1;                                       /* :74 */
  for (int64_t i = 0; i < c->n_lanes; i++) {
    float* o = c->obs + i * (int64_t)rows * columns;
    memset(o, 0, sizeof(float) * rows * columns);                      /* :110 */
    int type; double reward = 0.0;
    if (c->force_reset || reset_next[i]) {                             /* :80-81, :68-76 */
      draws_t d; draws_begin(&d, c->seed, c->lane_ids[i], c->step, 0);
      reset_next[i] = 0;
      ball_x[i] = (int32_t)draw_randint(&d, (uint32_t)columns);        /* :71 */
      ball_y[i] = 0;
      paddle_x[i] = columns / 2;
      type = FIRST;
    } else {
      int dx = c->action[i] - 1;                                       /* _ACTIONS :27 */
      int px = paddle_x[i] + dx;                                       /* :85 */
      if (px < 0) px = 0; if (px > columns - 1) px = columns - 1;
      paddle_x[i] = px;
      ball_y[i] += 1;                                                  /* :88 */
      if (ball_y[i] == paddle_y) {                                     /* :91-95 */
        reward = (paddle_x[i] == ball_x[i]) ? 1.0 : -1.0;
        reset_next[i] = 1;
        total_regret[i] += (1.0 - reward);
        type = LAST;
      } else {
        type = MID;                                                    /* :97 */
      }
    }
    o[ball_y[i] * columns + ball_x[i]] = 1.0f;                         /* :111 */
    o[paddle_y * columns + paddle_x[i]] = 1.0f;                        /* :112 */
    emit(c, i, type, reward);
  }
}

/* ------------------------------------------------------------------ bandit.py */
void orc_bandit(const orc_call* c, int num_actions, const double* rewards, int32_t* reset_next,
                double* total_regret) {
  (void)num_actions;
  for (int64_t i = 0; i < c->n_lanes; i++) {
    c->obs[i] = 1.0f;                                                  /* :54 (ones, not zeros) */
    if (c->force_reset || reset_next[i]) {
      reset_next[i] = 0;
      emit(c, i, FIRST, 0.0);
      continue;
    }
    double reward = rewards[c->action[i]];                             /* :61 */
    total_regret[i] += 1.0 - reward;                                   /* :62, optimal_return=1. */
    reset_next[i] = 1;
    emit(c, i, LAST, reward);                                          /* :64 */
  }
}

/* ------------------------------------------------------------------ memory_chain.py */
static void memory_obs(float* o, int nb, int L, int t, int query, const int32_t* ctx) {
  memset(o, 0, sizeof(float) * (nb + 2));                              /* :62 */
  o[0] = (float)(1.0 - (double)t / (double)L);                         /* :64 */
  if (t == L - 1) o[1] = (float)query;                                 /* :66-67 */
  if (t == 0) for (int b = 0; b < nb; b++) o[2 + b] = (float)(2 * ctx[b] - 1);   /* :69-70 */
}

void orc_memory_chain(const orc_call* c, int L, int nb, int32_t* timestep, int32_t* query,
                      int32_t* context /* [n_lanes, nb] */, int32_t* reset_next,
                      double* total_perfect, double* total_regret) {
  for (int64_t i = 0; i < c->n_lanes; i++) {
    float* o = c->obs + i * (int64_t)(nb + 2);
    int32_t* ctx = context + i * (int64_t)nb;
    if (c->force_reset || reset_next[i]) {                             /* :91-97 */
      draws_t d; draws_begin(&d, c->seed, c->lane_ids[i], c->step, 0);
      reset_next[i] = 0;
      timestep[i] = 0;
      uint32_t w = 0;
      for (int b = 0; b < nb; b++) {                                   /* BernVec(nb) */
        if ((b & 31) == 0) w = next_word(&d);
        ctx[b] = (int32_t)((w >> (b & 31)) & 1u);
      }
      query[i] = (int32_t)draw_randint(&d, (uint32_t)nb);
      memory_obs(o, nb, L, timestep[i], query[i], ctx);
      emit(c, i, FIRST, 0.0);
      continue;
    }
    memory_obs(o, nb, L, timestep[i], query[i], ctx);                  /* :74 — BEFORE t += 1 */
    timestep[i] += 1;                                                  /* :75 */
    if (timestep[i] - 1 < L) { emit(c, i, MID, 0.0); continue; }       /* :77-79 */
    double reward;
    if (c->action[i] == ctx[query[i]]) { reward = 1.0; total_perfect[i] += 1.0; }   /* :83-85 */
    else { reward = -1.0; total_regret[i] += 2.0; }                    /* :86-88 */
    reset_next[i] = 1;
    emit(c, i, LAST, reward);
  }
}

/* ------------------------------------------------------------------ umbrella_chain.py */
static void umbrella_obs(float* o, int nd, int L, int t, int need, int has, draws_t* d) {
  o[0] = (float)need;                                                  /* :62 */
  o[1] = (float)has;                                                   /* :63 */
  o[2] = (float)(1.0 - (double)t / (double)L);                         /* :64 */
  uint32_t w = 0;
  for (int b = 0; b < nd; b++) {                                       /* :65 BernVec(nd) */
    if ((b & 31) == 0) w = next_word(d);
    o[3 + b] = (float)((w >> (b & 31)) & 1u);
  }
}

void orc_umbrella_chain(const orc_call* c, int L, int nd, int32_t* timestep, int32_t* need,
                        int32_t* has, int32_t* reset_next, double* total_regret) {
  for (int64_t i = 0; i < c->n_lanes; i++) {
    float* o = c->obs + i * (int64_t)(3 + nd);
    draws_t d; draws_begin(&d, c->seed, c->lane_ids[i], c->step, 0);
    if (c->force_reset || reset_next[i]) {                             /* :87-92 */
      reset_next[i] = 0;
      timestep[i] = 0;
      need[i] = draw_bern(&d);
      has[i] = draw_bern(&d);
      umbrella_obs(o, nd, L, 0, need[i], has[i], &d);
      emit(c, i, FIRST, 0.0);
      continue;
    }
    timestep[i] += 1;                                                  /* :69 */
    if (timestep[i] == 1) has[i] = c->action[i];                       /* :71-72 */
    if (timestep[i] == L) {                                            /* :74-81 */
      double reward;
      if (has[i] == need[i]) reward = 1.0;
      else { reward = -1.0; total_regret[i] += 2.0; }
      umbrella_obs(o, nd, L, timestep[i], need[i], has[i], &d);
      reset_next[i] = 1;
      emit(c, i, LAST, reward);
    } else {                                                           /* :83-85 */
      double reward = 2.0 * (double)draw_bern(&d) - 1.0;
      umbrella_obs(o, nd, L, timestep[i], need[i], has[i], &d);
      emit(c, i, MID, reward);
    }
  }
}

/* ------------------------------------------------------------------ discounting_chain.py */
void orc_discounting_chain(const orc_call* c, int mapping_seed, int32_t* timestep, int32_t* context,
                           int32_t* reset_next) {
  static const int reward_timestep[5] = {1, 3, 10, 30, 100};           /* :49 */
  double rewards[5] = {1.0, 1.0, 1.0, 1.0, 1.0};                       /* :57 */
  rewards[((mapping_seed % 5) + 5) % 5] += 0.1;                        /* :54-58 */
  for (int64_t i = 0; i < c->n_lanes; i++) {
    float* o = c->obs + i * 2;
    if (c->force_reset || reset_next[i]) {                             /* :69-73 */
      reset_next[i] = 0;
      timestep[i] = 0; context[i] = -1;
      o[0] = -1.0f; o[1] = 0.0f;
      emit(c, i, FIRST, 0.0);
      continue;
    }
    if (timestep[i] == 0) context[i] = c->action[i];                   /* :76-77 */
    timestep[i] += 1;
    int chain = context[i] < 0 ? context[i] + 5 : context[i];          /* :80-81 Python list indexing: -5..-1 wrap (outside
                                                                          -5..4 the reference raises IndexError: not a case here) */
    double reward = (timestep[i] == reward_timestep[chain]) ? rewards[chain] : 0.0;
    o[0] = (float)context[i];                                          /* :65 */
    o[1] = (float)((double)timestep[i] / 100.0);                       /* :66 */
    if (timestep[i] == 100) { reset_next[i] = 1; emit(c, i, LAST, reward); }
    else emit(c, i, MID, reward);
  }
}

/* ------------------------------------------------------------------ cartpole.py / cartpole_swingup.py */
typedef struct {
  int32_t swingup;
  double height_threshold, x_threshold, theta_dot_threshold, x_reward_threshold, move_cost;
  double timescale, max_time, init_range;
  double mass_cart, mass_pole, length, force_mag, gravity;
} orc_cartpole_cfg;

static double py_remainder(double a, double b) {           /* np.remainder for floats */
  double m = fmod(a, b);
  if (m != 0.0) { if ((b < 0) != (m < 0)) m += b; }
  else m = copysign(0.0, b);
  return m;
}

void orc_cartpole(const orc_call* c, const orc_cartpole_cfg* g,
                  double* st /* [n_lanes,5] x, x_dot, theta, theta_dot, time_elapsed */,
                  int32_t* reset_next, double* raw_return, double* best_episode,
                  double* episode_return, double* total_upright) {
  const int nobs = g->swingup ? 8 : 6;
  for (int64_t i = 0; i < c->n_lanes; i++) {
    double* s = st + i * 5;
    float* o = c->obs + i * (int64_t)nobs;
    int type; double reward = 0.0;
    if (c->force_reset || reset_next[i]) {              /* cartpole.py:118-128 / swingup:81-91 */
      draws_t d; draws_begin(&d, c->seed, c->lane_ids[i], c->step, 0);
      reset_next[i] = 0;
      double lo = -g->init_range, hi = g->init_range;
      s[0] = lo + (hi - lo) * draw_uniform(&d);
      s[1] = lo + (hi - lo) * draw_uniform(&d);
      double th = lo + (hi - lo) * draw_uniform(&d);
      s[2] = g->swingup ? M_PI + th : th;
      s[3] = lo + (hi - lo) * draw_uniform(&d);
      s[4] = 0.0;
      episode_return[i] = 0.0;
      type = FIRST;
    } else {
      int a = c->action[i];
      /* step_cartpole, cartpole.py:37-65 */
      double force = (double)(a - 1) * g->force_mag;
      double co = cos(s[2]), si = sin(s[2]);
      double pl = g->mass_pole * g->length;
      double l = g->length, m_pole = g->mass_pole, m_total = g->mass_cart + g->mass_pole;
      double temp = (force + pl * (s[3] * s[3]) * si) / m_total;
      double theta_acc = (g->gravity * si - co * temp) / (l * (4.0 / 3.0 - m_pole * (co * co) / m_total));
      double x_acc = temp - pl * theta_acc * co / m_total;
      double x = s[0] + g->timescale * s[1];
      double x_dot = s[1] + g->timescale * x_acc;
      double theta = py_remainder(s[2] + g->timescale * s[3], 2.0 * M_PI);
      double theta_dot = s[3] + g->timescale * theta_acc;
      double t_el = s[4] + g->timescale;
      s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot; s[4] = t_el;
      int end;
      if (!g->swingup) {                                /* cartpole.py:142-153 */
        int is_reward = (cos(s[2]) > g->height_threshold) && (fabs(s[0]) < g->x_threshold);
        reward = is_reward ? 1.0 : 0.0;
        raw_return[i] += reward; episode_return[i] += reward;
        end = (s[4] > g->max_time) || !is_reward;
      } else {                                          /* swingup:104-123 */
        int upright = (cos(s[2]) > g->height_threshold) && (fabs(s[3]) < g->theta_dot_threshold) &&
                      (fabs(s[0]) < g->x_reward_threshold);
        reward = -1.0 * fabs((double)(a - 1)) * g->move_cost;
        if (upright) { reward += 1.0; total_upright[i] += 1.0; }
        raw_return[i] += reward; episode_return[i] += reward;
        end = (s[4] > g->max_time) || (fabs(s[0]) > g->x_threshold);
      }
      if (end) {
        best_episode[i] = episode_return[i] > best_episode[i] ? episode_return[i] : best_episode[i];
        reset_next[i] = 1;
        type = LAST;
      } else type = MID;
    }
    o[0] = (float)(s[0] / g->x_threshold);              /* cartpole.py:171-176 / swingup:141-149 */
    o[1] = (float)(s[1] / g->x_threshold);
    o[2] = (float)sin(s[2]);
    o[3] = (float)cos(s[2]);
    o[4] = (float)s[3];
    o[5] = (float)(s[4] / g->max_time);
    if (g->swingup) {
      o[6] = (fabs(s[0]) < g->x_reward_threshold) ? 1.0f : -1.0f;
      o[7] = (fabs(s[3]) < g->theta_dot_threshold) ? 1.0f : -1.0f;
    }
    emit(c, i, type, reward);
  }
}

/* ------------------------------------------------------------------ mountain_car.py */
static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_mountain_car(const orc_call* c, int max_steps, double* position, double* velocity,
                      int32_t* timestep, int32_t* reset_next, double* raw_return) {
  const double min_pos = -1.2, max_pos = 0.6, max_speed = 0.07, goal_pos = 0.5, force = 0.001,
               gravity = 0.0025;                                       /* :47-52 */
  for (int64_t i = 0; i < c->n_lanes; i++) {
    float* o = c->obs + i * 3;
    int type; double reward = 0.0;
    if (c->force_reset || reset_next[i]) {                             /* :66-71 */
      draws_t d; draws_begin(&d, c->seed, c->lane_ids[i], c->step, 0);
      reset_next[i] = 0;
      timestep[i] = 0;
      position[i] = -0.6 + (-0.4 - -0.6) * draw_uniform(&d);
      velocity[i] = 0.0;
      type = FIRST;
    } else {
      timestep[i] += 1;                                                /* :74-76 */
      reward = -1.0;
      raw_return[i] += reward;
      velocity[i] += (double)(c->action[i] - 1) * force + cos(3.0 * position[i]) * -gravity;
      velocity[i] = clipd(velocity[i], -max_speed, max_speed);
      position[i] += velocity[i];
      position[i] = clipd(position[i], min_pos, max_pos);
      if (position[i] == min_pos) velocity[i] = clipd(velocity[i], 0.0, max_speed);
      if (position[i] >= goal_pos || timestep[i] >= max_steps) { reset_next[i] = 1; type = LAST; }
      else type = MID;
    }
    o[0] = (float)position[i];                                         /* :62-64 */
    o[1] = (float)velocity[i];
    o[2] = (float)((double)timestep[i] / (double)max_steps);
    emit(c, i, type, reward);
  }
}

/* ------------------------------------------------------------------ mnist.py */
void orc_mnist(const orc_call* c, int num_data, int num_pixels, const int8_t* images /* [num_data, num_pixels] */,
               const uint8_t* labels, int32_t* correct_label, int32_t* reset_next, double* total_regret) {
  for (int64_t i = 0; i < c->n_lanes; i++) {
    float* o = c->obs + i * (int64_t)num_pixels;
    if (c->force_reset || reset_next[i]) {                             /* :61-67 */
      draws_t d; draws_begin(&d, c->seed, c->lane_ids[i], c->step, 0);
      reset_next[i] = 0;
      uint32_t idx = draw_randint(&d, (uint32_t)num_data);
      for (int p = 0; p < num_pixels; p++)                             /* astype(float32) / 255 */
        o[p] = (float)images[(int64_t)idx * num_pixels + p] / 255.0f;
      correct_label[i] = labels[idx];
      emit(c, i, FIRST, 0.0);
      continue;
    }
    double reward = (c->action[i] == correct_label[i]) ? 1.0 : -1.0;   /* :71-72 */
    total_regret[i] += 1.0 - reward;                                   /* :73 */
    memset(o, 0, sizeof(float) * num_pixels);                          /* :74 */
    reset_next[i] = 1;
    emit(c, i, LAST, reward);
  }
}
