/* bsuite_amd.h — C ABI of the MI355X-native batched bsuite environment engine.
 *
 * The reference (google-deepmind/bsuite) has no FFI: its boundary is the Python protocol
 *   env.reset() -> dm_env.TimeStep ; env.step(action:int) -> dm_env.TimeStep
 * of bsuite/environments/base.py:54-65.  This header is the batched, device-side form of exactly
 * that protocol: one *lane* per environment instance, struct-of-arrays state columns, and one
 * entry point per environment family.  Every entry point is the drop-in for the `_step/_reset`
 * pair it cites.  All pointers are DEVICE pointers (HIP) unless a comment says "host"; nothing
 * here allocates, synchronises or reads back — calls are asynchronous on the caller's stream and
 * safe to capture into a hipGraph.  The `info` columns must be ordinary (coarse-grained) device
 * memory, e.g. hipMalloc: memory_chain's end-of-episode update is a hardware f64 atomic, which does
 * not reach fine-grained host mappings.
 *
 * Return value of every function: 0 = ok, <0 = argument error (see bsx_strerror), >0 = hipError_t.
 *
 * Batched TimeStep encoding (dm_env is third-party):
 *   step_type int8: 0 FIRST, 1 MID, 2 LAST.   FIRST lanes carry reward 0 / discount 1 in the
 *   batched arrays (the reference returns None/None there; the batch=1 Python view restores None).
 *
 * The random-draw stream ("bsx stream v1") is specified in include/bsx_stream.h.
 */
#ifndef BSUITE_AMD_H_
#define BSUITE_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSX_ABI_VERSION 12

#define BSX_FIRST 0
#define BSX_MID 1
#define BSX_LAST 2

/* error codes (<0) */
#define BSX_EINVAL (-1)     /* bad scalar argument (size, n_lanes, ...)            */
#define BSX_ENULL (-2)      /* a required pointer is NULL                          */
#define BSX_EALIGN (-3)     /* observation pointer not 16-byte aligned             */
#define BSX_ERANGE (-4)     /* parameter outside the supported range of the family */
#define BSX_EMODE (-5)      /* combination not available (obs_paint with a rollout / group / other family) */
#define BSX_ENOMEM (-6)     /* host allocation failed (group bookkeeping)          */

/* Random stream coordinates of one call (include/bsx_stream.h).  The reference gives every env its
 * own np.random.RandomState (e.g. deep_sea.py:77, catch.py:58); here a lane's draws are a pure
 * function of (seed, global lane id, step_index, draw#) so shards of any size reproduce the same
 * trajectories.  `step_base` (device, nullable) is added to `step_index` on the device so that a
 * captured hipGraph can advance its own call counter. */
typedef struct {
  uint64_t seed;            /* Philox key                                                     */
  uint64_t lane_offset;     /* global id of this shard's lane 0                               */
  uint64_t step_index;      /* index of this reset()/step() call (monotonic per env batch)    */
  const uint64_t* step_base; /* device pointer or NULL                                        */
  /* MT19937-exact mode (include/bsx_stream.h "mode B"); both NULL = counter-based
   * stream.  mt_state: device uint32 [624, B] (word k of lane i at [k*B + i]), each lane holding
   * np.random.RandomState's key; mt_pos: device int32 [B], its `pos`.  The kernels then draw with
   * numpy's legacy samplers from that generator and write the advanced state back.
   * mt_gauss / mt_has_gauss (ABI v8): device double [B] / int32 [B], RandomState's cached second
   * value of the polar Box-Muller pair (`gauss`, `has_gauss`); required (with mt_state) by the
   * stochastic deep_sea, whose `randn` then is numpy's own, log included (include/bsx_libm_log.h);
   * may be NULL for every other family.                                                         */
  uint32_t* mt_state;
  int32_t* mt_pos;
  double* mt_gauss;
  int32_t* mt_has_gauss;
} bsx_stream_t;

/* Fused reward epilogue = bsuite/utils/wrappers.py RewardNoise (:275-283) / RewardScale (:338-346):
 * applied to non-FIRST lanes only; bsuite_info accumulators see the un-perturbed reward. */
#define BSX_WRAP_NONE 0
#define BSX_WRAP_SCALE 1
#define BSX_WRAP_NOISE 2
/* the two wrappers stacked (the reference composes them freely, utils/wrappers_test.py:123-131):       */
#define BSX_WRAP_SCALE_NOISE 3   /* RewardNoise(RewardScale(env)): r * param  + param2 * randn()          */
#define BSX_WRAP_NOISE_SCALE 4   /* RewardScale(RewardNoise(env)): (r + param * randn()) * param2        */
typedef struct {
  int32_t kind;
  int32_t _pad;
  double param;        /* reward_scale or noise_scale (sigma); stacked kinds: the INNER wrapper's        */
  uint64_t seed;       /* key of the wrapper's own stream (RewardNoise has its own RNG, :267)   */
  /* MT19937-exact mode (ABI v8): RewardNoise owns a second np.random.RandomState(seed) per lane
   * (wrappers.py:267) that only ever draws randn; same layout as the bsx_stream_t members.  All four
   * non-NULL when kind involves noise (>= BSX_WRAP_NOISE) and stream.mt_state != NULL, else ignored.             */
  uint32_t* mt_state;
  int32_t* mt_pos;
  double* mt_gauss;
  int32_t* mt_has_gauss;
  double param2;       /* stacked kinds: the OUTER wrapper's parameter                                  */
} bsx_reward_wrap_t;

/* The batched dm_env.TimeStep, written in full on every call (dense contract). */
typedef struct {
  float* reward;        /* [B]                                   */
  float* discount;      /* [B]                                   */
  int8_t* step_type;    /* [B]                                   */
  float* observation;   /* [B, obs_numel], 16-byte aligned       */
} bsx_timestep_t;

/* Batched form of the `Logging` wrapper's bookkeeping (bsuite/utils/wrappers.py:85-125): per-lane
 * steps / episode / total_return / episode_len / episode_return, updated from every emitted
 * TimeStep with the f64 (wrapped) reward, and a per-lane buffer of snapshot rows
 *   [steps, episode, total_return, episode_len, episode_return, info_0 .. info_{n_info-1}]
 * appended whenever the reference would call `logger.write` — at the log-spaced episode (or step)
 * counts of `_logarithmic_logging` (:140-147), supplied as a sorted device table.  Optional: a NULL
 * `logging` pointer in bsx_call_t costs nothing. */
typedef struct {
  int64_t* steps;             /* [B]                                                              */
  int64_t* episode;           /* [B]                                                              */
  double* total_return;       /* [B]                                                              */
  int64_t* episode_len;       /* [B]                                                              */
  double* episode_return;     /* [B]                                                              */
  double* rows;               /* [B, max_rows, 5 + n_info]                                        */
  int32_t* n_rows;            /* [B] rows the reference would have written (may exceed max_rows;
                                 rows beyond max_rows are dropped, the count keeps running)       */
  const double* info;         /* the family's info columns [n_info, B], or NULL when n_info == 0   */
  const int64_t* log_points;  /* device, ascending: the counts at which a row is written           */
  int32_t n_log_points;
  int32_t max_rows;
  int32_t n_info;
  int32_t log_by_step;        /* wrappers.py:44 */
  int32_t log_every;          /* wrappers.py:45 */
  int32_t _pad;
} bsx_logging_t;

#define BSX_COUNTER_SHARDS 256
#define BSX_COUNTER_STRIDE 16   /* uint64 per shard = one 128-byte line */

/* Per-call control shared by all families. */
typedef struct {
  int64_t n_lanes;          /* B of this shard                                                   */
  int32_t force_reset;      /* 1: this call is env.reset() (base.py:54-57) for every lane        */
  int32_t n_steps;          /* 0 or 1: one step() call.  T > 1: a rollout of T consecutive step()
                               calls in ONE entry-point call: `action` is [T,B], every bsx_timestep_t
                               array gains a leading T dimension (observation [T,B,obs_numel]), call
                               index of step t is step_index + t.  The small-observation families
                               fuse the T steps into one kernel; deep_sea / catch / mnist launch
                               their kernel pair T times — or, deep_sea / catch with `state_alt`,
                               T + 1 launches (software-pipelined, below).  Not combinable with
                               force_reset.                                                      */
  bsx_stream_t stream;
  bsx_reward_wrap_t wrap;
  uint64_t* counters;       /* device, nullable: BSX_COUNTER_SHARDS x BSX_COUNTER_STRIDE uint64.
                               shard s = block % SHARDS; [s*STRIDE+0] += lanes that emitted LAST,
                               [s*STRIDE+1] += lanes that emitted FIRST.  Masks come from wavefront
                               ballots; one global atomic per workgroup per mask, spread over 256
                               cache lines (a single hot word serialises ~12 ns per arrival).
                               [s*STRIDE+2] += lane-steps whose action was outside the action_spec
                               where the reference raises IndexError (bandit.py:61, catch.py:84,
                               discounting_chain.py:80): the kernels clamp / clip instead of
                               faulting and count the event here (error word).                  */
  void* hip_stream;         /* hipStream_t                                                       */
  const bsx_logging_t* logging; /* host pointer or NULL (ABI v2)                                     */
  int32_t* obs_paint;       /* device [B] int32 or NULL (ABI v7).  NULL: the observation array is written
                               in full every call (dense contract).  Non-NULL selects the DELTA
                               observation mode of deep_sea / catch (SURVEY §7 hard part 1): the
                               caller keeps `out.observation` persistent between the calls that use it
                               and never writes to it; obs_paint[i] is the packed state whose hot
                               cells are currently 1.0 in lane i's board (-1: the board is all
                               zeros, the initial value next to a zero-filled array).  The call then
                               only clears the stale cells and sets the new ones (<= 4 four-byte
                               stores per lane instead of 4*N*N bytes) and updates obs_paint; the
                               array contents after the call are identical to the dense mode's.
                               One obs_paint column per observation buffer.  Not available with
                               n_steps > 1, in groups, or for the other families (BSX_EMODE).   */
  double* reward_f64;       /* device [B] ([T,B] in a rollout) or NULL (ABI v8): the reward BEFORE the
                               cast to f32 — the f64 value the reference's step() returns (e.g.
                               deep_sea's -0.01/N move cost, RewardScale's 0.001*r); 0.0 on FIRST.
                               The scalar dm_env view reads its TimeStep.reward from here.       */
  int32_t* state_alt;       /* device [B] int32 or NULL (ABI v9): a second packed-state column for the
                               two-kernel families (deep_sea, catch, mnist), whose observation stream
                               reads the state column their lane advance wrote.  It lets the advance of
                               step t+1 share ONE launch with the observation stream of step t (the
                               stream keeps reading the column the advance no longer writes):
                               - in a rollout (n_steps > 1; deep_sea, catch) it is scratch: after the
                                 first advance, every launch is {stream of step t, advance of step
                                 t+1}, the advances alternating between `state` and `state_alt`;
                                 T + 1 launches instead of 2T, the final state ends up in `state`;
                               - in a segment of a BSX_FAM_SWEEP_MIXED group the lane advance READS
                                 `state_alt` and writes `state` (which the segment's stream reads): two
                                 groups with the columns swapped step alternately, see
                                 bsx_group_step_pipelined.
                               Ignored by step()/reset() calls and by the other families.          */
  int32_t action_ring;      /* 0 or 1: `action` is [B].  R = 2^k > 1 (ABI v10): `action` is a ring [R,B] of
                               pre-generated actions and the call whose index is s (stream.step_index +
                               *stream.step_base) reads row s mod R — the batched form of an agent that acts
                               without looking (bsuite/baselines/random/agent.py:35-37) for callers whose
                               arguments are static: the segments of a group (every group step then feeds
                               fresh actions with no host work) and captured hipGraphs.  Not a power of two:
                               BSX_EINVAL; with n_steps > 1 or obs_paint: BSX_EMODE.                 */
  int32_t flags;            /* BSX_CALL_* bits (ABI v11; was padding, 0 = the v10 behaviour)                  */
  void* row_scratch;        /* device or NULL (ABI v12), memory_chain / umbrella_chain with an observation row of more
                               than 8 floats: bsx_row_scratch_bytes(family, obs_numel, n_lanes) bytes, 16-byte aligned,
                               contents irrelevant between calls.  With it a single step()/reset() call is lane advance
                               + store stream, like deep_sea / catch: every wave of the advance leaves its 64 lanes'
                               rows in the scratch as FLAT BIT PLANES (bit e of a plane = element e of the observation
                               array; the genuine floats of a row in one f32 column each: csrc/bsx_rows.h) and a
                               barrier-free store stream decodes them into `out.observation` (csrc/row_stream.h).
                               NULL selects the one launch that builds the rows as bit planes in LDS (three workgroup
                               barriers per step), which rollouts (n_steps > 1) always take.  In a BSX_FAM_SWEEP_MIXED
                               group such a segment's rows are decoded by the phase-1 store stream; the two groups of
                               a pipelined pair must bring DIFFERENT scratches (bsx_group_step_pipelined: BSX_EMODE);
                               the single-launch groups ignore it.  Ignored by the other families and by short rows. */
} bsx_call_t;

/* bsx_call_t.flags */
#define BSX_CALL_STATE_TAGGED 1 /* deep_sea: the caller guarantees that bit 18 of EVERY lane's packed state word equals
                                  the parity of this call's index (stream.step_index + *step_base) — true for a
                                  column that starts as (1<<17) | (index & 1) << 18 and is only ever advanced by
                                  this library with consecutive call indices (every advance writes the next
                                  parity).  A deterministic, un-wrapped single step()/reset() call is then ONE
                                  launch instead of two: the threads of the observation store stream recompute the
                                  transition of the lane whose row they write (deep_sea.hip).  Without the flag:
                                  lane advance + store stream, as in v10.                                      */

/* A catch segment of a BSX_FAM_SWEEP_MIXED group whose board has at most this many cells and that was set WITHOUT
 * state_alt has its boards written by phase 0 itself (one fused tile per workgroup) and takes no part in the phase-1
 * store stream: in a pipelined pair of groups it needs one state column and an observation buffer per group, like a
 * small-observation family.  Larger boards are two-kernel segments and need state_alt there. */
#define BSX_FUSED_CATCH_MAX_CELLS 128

/* ---- deep_sea : bsuite/environments/deep_sea.py:51-155 ------------------------------------ */
#define BSX_DEEP_SEA_MAX_SIZE 64
typedef struct {
  int32_t size;                 /* N (deep_sea.py:52); 2..64                                      */
  int32_t deterministic;        /* deep_sea.py:53                                                 */
  double move_cost;             /* unscaled_move_cost / size, evaluated in f64 on the host (:132) */
  double inv_size;              /* 1 / size in f64 (:130)                                         */
  uint32_t mapping_bits[BSX_DEEP_SEA_MAX_SIZE * BSX_DEEP_SEA_MAX_SIZE / 32];
                                /* action_mapping[row,col] bit (row*N+col), host-built (:77-85)   */
} bsx_deep_sea_t;
/* state: int32 [B] = row | col<<8 | bad_episode<<16 | reset_next<<17 | parity_of_next_call<<18  (initialise to 1<<17;
 * bit 18 is maintained by the library, see BSX_CALL_STATE_TAGGED)
 * info : double [2,B] = total_bad_episodes, denoised_return (deep_sea.py:153-155)
 * obs  : float [B, N, N] */
int bsx_deep_sea_step(const bsx_deep_sea_t* cfg /*host*/, const bsx_call_t* call /*host*/,
                      const int32_t* action, int32_t* state, bsx_timestep_t out, double* info);

/* ---- catch : bsuite/environments/catch.py:45-117 ------------------------------------------ */
typedef struct {
  int32_t rows;     /* catch.py:46 (default 10); 2..64 */
  int32_t columns;  /* catch.py:47 (default 5);  1..64 */
} bsx_catch_t;
/* state: int32 [B] = ball_x | ball_y<<8 | paddle_x<<16 | reset_next<<24 | pending_misses<<25 (initialise to 1<<24)
 * info : double [1,B] = total_regret (catch.py:116-117).  Accounting (ABI v10): with call->logging the column is
 *        updated at every episode end like the reference.  Without it the lane counts its misses (regret 2 each, a
 *        catch costs 0: catch.py:92-94) in bits 25..31 of the state word and adds 2*127 to the column once per 127
 *        misses: the reference's value is  info[0][i] + 2 * ((uint32_t)state[i] >> 25)  (bsx_bsuite_info computes it).
 * obs  : float [B, rows, columns] */
int bsx_catch_step(const bsx_catch_t* cfg, const bsx_call_t* call, const int32_t* action,
                   int32_t* state, bsx_timestep_t out, double* info);

/* ---- bandit : bsuite/environments/bandit.py:35-73 ----------------------------------------- */
#define BSX_BANDIT_MAX_ACTIONS 32
typedef struct {
  int32_t num_actions;                       /* bandit.py:35 (default 11) */
  int32_t _pad;
  double rewards[BSX_BANDIT_MAX_ACTIONS];    /* permuted linspace, host-built (bandit.py:45-47) */
} bsx_bandit_t;
/* state: int32 [B] = reset_next (initialise to 1); info: double [1,B] = total_regret; obs [B,1,1] */
int bsx_bandit_step(const bsx_bandit_t* cfg, const bsx_call_t* call, const int32_t* action,
                    int32_t* state, bsx_timestep_t out, double* info);

/* ---- memory_chain : bsuite/environments/memory_chain.py:37-112 ---------------------------- */
typedef struct {
  int32_t memory_length;  /* L  (memory_chain.py:39); >=1         */
  int32_t num_bits;       /* nb (memory_chain.py:40); 1..62       */
} bsx_memory_chain_t;
/* state: int32 [B] = timestep | query<<20 | reset_next<<28 (initialise to 1<<28)
 * context: uint64 [B] (bit i = context[i])
 * info : double [2,B] = total_perfect, total_regret (memory_chain.py:108-112)
 * obs  : float [B, 1, nb+2]; nb > 6: see bsx_call_t.row_scratch */
int bsx_memory_chain_step(const bsx_memory_chain_t* cfg, const bsx_call_t* call,
                          const int32_t* action, int32_t* state, uint64_t* context,
                          bsx_timestep_t out, double* info);

/* ---- umbrella_chain : bsuite/environments/umbrella_chain.py:37-114 ------------------------- */
typedef struct {
  int32_t chain_length;   /* L  (umbrella_chain.py:38); >=1   */
  int32_t n_distractor;   /* nd (umbrella_chain.py:39); 0..253 */
} bsx_umbrella_chain_t;
/* state: int32 [B] = timestep | need<<20 | has<<21 | reset_next<<22 (initialise to 1<<22)
 * info : double [1,B] = total_regret; obs float [B, 1, 3+nd]; nd > 5: see bsx_call_t.row_scratch */
int bsx_umbrella_chain_step(const bsx_umbrella_chain_t* cfg, const bsx_call_t* call,
                            const int32_t* action, int32_t* state, bsx_timestep_t out,
                            double* info);

/* ---- discounting_chain : bsuite/environments/discounting_chain.py:40-105 ------------------- */
typedef struct {
  int32_t bonus_chain;    /* mapping_seed % 5 (discounting_chain.py:52-56) */
  int32_t _pad;
} bsx_discounting_chain_t;
/* state: int32 [B] = timestep | (context+1)<<8 | reset_next<<12 (initialise to 1<<12)
 * info : none (discounting_chain.py:104-105 returns {}); obs float [B,1,2] */
int bsx_discounting_chain_step(const bsx_discounting_chain_t* cfg, const bsx_call_t* call,
                               const int32_t* action, int32_t* state, bsx_timestep_t out);

/* ---- cartpole (+ swingup) : bsuite/environments/cartpole.py:37-181,
 *      bsuite/experiments/cartpole_swingup/cartpole_swingup.py:30-155 ------------------------ */
typedef struct {
  int32_t swingup;              /* 0: Cartpole, 1: CartpoleSwingup                              */
  int32_t last_step;            /* first k with f64-accumulated time_elapsed > max_time (host)  */
  float height_threshold;       /* cartpole.py:84 / swingup:39                                   */
  float x_threshold;            /* cartpole.py:85 / swingup:43                                   */
  float theta_dot_threshold;    /* swingup:40                                                    */
  float x_reward_threshold;     /* swingup:41                                                    */
  float timescale;              /* cartpole.py:86                                                */
  float mass_cart, mass_pole, length, force_mag, gravity; /* cartpole.py:106-112                 */
  double move_cost;             /* swingup:42 (f64: the reward is formed in f64 like the ref)    */
  double init_range;            /* cartpole.py:88                                                */
  double theta_offset;          /* 0 (cartpole) or pi (swingup :87)                              */
  const float* time_frac;       /* device [last_step+1]: f32(time_elapsed_k / max_time), host-
                                   built from the f64 running sum (cartpole.py:63,176)           */
} bsx_cartpole_t;
/* state: float [4,B] = x, x_dot, theta, theta_dot ; steps: int32 [B] = k | reset_next<<30
 * (initialise to 1<<30); info double [4,B] = raw_return, best_episode, episode_return,
 * total_upright (cartpole.py:179-181, swingup:152-155); obs float [B,1,6] or [B,1,8].
 * |theta_offset| + init_range must be <= 32 (BSX_ERANGE otherwise).
 * Accounting of the info columns (ABI v8): swing-up, and any call with call->logging set, accumulate
 * them per step like the reference.  Classic cartpole without logging (rewards are 0/1, so an episode
 * of k steps returns exactly (k-1) + its last reward) folds raw_return and best_episode into the
 * columns when an episode ENDS and leaves episode_return untouched: mid-episode the reference's
 * raw_return is  info[0][i] + (steps[i] >> 30 ? 0 : steps[i] & 0x3FFFFFFF)  (bsx_bsuite_info computes it). */
int bsx_cartpole_step(const bsx_cartpole_t* cfg, const bsx_call_t* call, const int32_t* action,
                      float* state, int32_t* steps, bsx_timestep_t out, double* info);

/* ---- mountain_car : bsuite/environments/mountain_car.py:32-102 ----------------------------- */
typedef struct {
  int32_t max_steps;   /* mountain_car.py:36 */
  int32_t _pad;
} bsx_mountain_car_t;
/* state: float [2,B] = position, velocity ; steps: int32 [B] = timestep | reset_next<<30
 * info double [1,B] = raw_return; obs float [B,1,3].
 * Every step pays -1: without call->logging raw_return is folded into the column when an episode
 * ENDS (ABI v8); mid-episode the reference's value is
 * info[0][i] - (steps[i] >> 30 ? 0 : steps[i] & 0x3FFFFFFF)  (bsx_bsuite_info computes it).  With call->logging it is per step. */
int bsx_mountain_car_step(const bsx_mountain_car_t* cfg, const bsx_call_t* call,
                          const int32_t* action, float* state, int32_t* steps,
                          bsx_timestep_t out, double* info);

/* ---- mnist bandit : bsuite/environments/mnist.py:33-89, bsuite/utils/datasets.py:42-69 -------- */
typedef struct {
  int32_t num_data;        /* int(fraction * len(labels)) (mnist.py:46-48); 1..2^24               */
  int32_t num_pixels;      /* rows*cols of one image, multiple of 4 (28*28 = 784)                   */
  const int8_t* images;    /* device [num_data, num_pixels]: the idx bytes as int8, exactly as
                              datasets.py:55-56 parses them (bright pixels are negative)            */
  const uint8_t* labels;   /* device [num_data]                                                     */
  float pixel_lut[256];    /* host-built np.float32(int8 value) / 255 for byte b (index = b as u8)  */
} bsx_mnist_t;
/* state: int32 [B] = image_index | label<<24 | reset_next<<28 | showing_image<<29 (init 1<<28)
 * info : double [1,B] = total_regret; obs float [B, rows, cols] */
int bsx_mnist_step(const bsx_mnist_t* cfg, const bsx_call_t* call, const int32_t* action,
                   int32_t* state, bsx_timestep_t out, double* info);

/* ---- grouped launch: many independent segments of ONE family in one kernel launch -------------
 * BASELINE config 5 (the whole sweep as lane segments) issues ~10^3 tiny launches per sweep step
 * when every bsuite_id is stepped on its own; a group turns that into one launch (pair) per family.
 * Every segment keeps its own configuration, state columns and output buffers — exactly the
 * arguments of its `bsx_<family>_step` — recorded once with `bsx_group_set_<family>`; workgroups
 * find their segment through a block-offset table in device memory ("grouped GEMM" style).
 * Because the recorded arguments are static, each segment's call index must come from a
 * device-resident counter (`stream.step_base != NULL`, bumped by the caller once per group step
 * with bsx_counter_add); `force_reset` and `n_steps > 1` are not available in a group.
 * create / commit / destroy allocate device memory and synchronise (setup time only);
 * bsx_group_step is asynchronous and capturable like every other entry point. */
typedef struct bsx_group bsx_group_t;
#define BSX_FAM_DEEP_SEA 0
#define BSX_FAM_CATCH 1
#define BSX_FAM_BANDIT 2
#define BSX_FAM_MEMORY_CHAIN 3
#define BSX_FAM_UMBRELLA_CHAIN 4
#define BSX_FAM_DISCOUNTING_CHAIN 5
#define BSX_FAM_CARTPOLE 6
#define BSX_FAM_MOUNTAIN_CAR 7
#define BSX_FAM_MNIST 8
/* A group of this family accepts bsx_group_set_<family> of bandit, memory_chain, umbrella_chain,
 * discounting_chain, cartpole and mountain_car segments alike and advances them with ONE launch. */
#define BSX_FAM_SMALL_MIXED 9
#define BSX_FAM_PAIR_MIXED 10  /* segments of deep_sea, catch and mnist together: ONE advance launch + ONE
                                  observation-stream launch for all of them (bsx_group_set_deep_sea /
                                  _catch / _mnist accept such a group); csrc/pair_mixed.hip */
#define BSX_FAM_SWEEP_MIXED 11 /* segments of ALL families (every bsx_group_set_<family> accepts such a group):
                                  phase 0 advances every lane of the sweep in ONE launch — the advance of the
                                  two-kernel families and the whole step of the small-observation families —
                                  and, when its last workgroup retires, bumps the call counter itself (all
                                  segments must share one stream.step_base; do NOT bsx_counter_add it);
                                  phase 1 is the one observation store stream of BSX_FAM_PAIR_MIXED.
                                  EXCLUSIVITY: the self-bump and the retirement ticket use plain (unfenced) device
                                  accesses — every launch that reads or bumps that step_base (this group's steps,
                                  eager step() calls of its segments, bsx_counter_add) must be STREAM-ORDERED with
                                  the group's launches: same HIP stream, or an event between them.  Two groups
                                  over one counter may alternate (bsx_group_step_pipelined) but never overlap.   */
int bsx_group_create(int32_t family, int32_t n_segments, bsx_group_t** group);
int bsx_group_set_deep_sea(bsx_group_t* g, int32_t index, const bsx_deep_sea_t* cfg, const bsx_call_t* call,
                           const int32_t* action, int32_t* state, bsx_timestep_t out, double* info);
int bsx_group_set_catch(bsx_group_t* g, int32_t index, const bsx_catch_t* cfg, const bsx_call_t* call,
                        const int32_t* action, int32_t* state, bsx_timestep_t out, double* info);
int bsx_group_set_bandit(bsx_group_t* g, int32_t index, const bsx_bandit_t* cfg, const bsx_call_t* call,
                         const int32_t* action, int32_t* state, bsx_timestep_t out, double* info);
int bsx_group_set_memory_chain(bsx_group_t* g, int32_t index, const bsx_memory_chain_t* cfg,
                               const bsx_call_t* call, const int32_t* action, int32_t* state,
                               uint64_t* context, bsx_timestep_t out, double* info);
int bsx_group_set_umbrella_chain(bsx_group_t* g, int32_t index, const bsx_umbrella_chain_t* cfg,
                                 const bsx_call_t* call, const int32_t* action, int32_t* state,
                                 bsx_timestep_t out, double* info);
int bsx_group_set_discounting_chain(bsx_group_t* g, int32_t index, const bsx_discounting_chain_t* cfg,
                                    const bsx_call_t* call, const int32_t* action, int32_t* state,
                                    bsx_timestep_t out);
int bsx_group_set_cartpole(bsx_group_t* g, int32_t index, const bsx_cartpole_t* cfg, const bsx_call_t* call,
                           const int32_t* action, float* state, int32_t* steps, bsx_timestep_t out,
                           double* info);
int bsx_group_set_mountain_car(bsx_group_t* g, int32_t index, const bsx_mountain_car_t* cfg,
                               const bsx_call_t* call, const int32_t* action, float* state, int32_t* steps,
                               bsx_timestep_t out, double* info);
int bsx_group_set_mnist(bsx_group_t* g, int32_t index, const bsx_mnist_t* cfg, const bsx_call_t* call,
                        const int32_t* action, int32_t* state, bsx_timestep_t out, double* info);
/* Tile class (lanes per workgroup) a small-observation segment with `numel` observation floats gets
 * inside a group; segments of one group must share it.  Since ABI v9 always 256: the wide rows are
 * staged as bit planes (bsuite_amd/csrc/small_obs.hip), the 64-lane class is gone. */
int bsx_group_small_class(int32_t numel);
int bsx_group_commit(bsx_group_t* g);
int bsx_group_step(bsx_group_t* g, void* hip_stream);
/* A group step in its phases, for callers that schedule them on several HIP streams (ABI v8): a
 * two-kernel family (deep_sea, catch, mnist) has 2 — phase 0 the lane advance (latency-bound, little
 * traffic), phase 1 the observation stream (HBM-bound), which depends on phase 0 of the SAME group only;
 * the small-observation groups have 1.  bsx_group_step == all phases in order on one stream. */
int bsx_group_phases(const bsx_group_t* g);
int bsx_group_step_phase(bsx_group_t* g, int32_t phase, void* hip_stream);
/* Split closed-loop sweep step (ABI v12) of a BSX_FAM_SWEEP_MIXED group: two launches like bsx_group_step — same results,
 * TimeSteps of the step complete when it returns to the stream — cut so that only what the store stream depends on
 * stands in front of it: launch 1 runs the phase-0 workgroups of the segments that have a share of the stream (lane
 * advance of deep_sea / mnist / large catch boards, packed rows of the chains), launch 2 the store stream BESIDE the
 * whole step of every other small-observation segment, whose last workgroup bumps the call counter.  (When phase 0 is
 * more than one dispatch round, launch 1 also takes the LAST small-observation workgroups of the group, up to one round
 * in all: put the cheapest small-observation segments right before the ones with a stream share.)  Needs the
 * segments with a stream share to be the LAST segments of the group (set the others first): BSX_EMODE otherwise. */
int bsx_group_step_split(bsx_group_t* g, void* hip_stream);
/* Software-pipelined sweep step (ABI v9), for callers whose actions do not depend on the observations
 * (a rollout with given actions, BASELINE config 5): ONE launch runs phase 1 — the observation stream —
 * of `streams_of` beside phase 0 — every lane's advance — of `advances_of`.  Both are committed
 * BSX_FAM_SWEEP_MIXED groups over the same segments; their two-kernel segments were set with the state
 * columns swapped (group E: state = A, state_alt = B; group O: state = B, state_alt = A; B starts as a
 * copy of A) and each group has its own reward / discount / step_type buffers (and observation buffers
 * for the small-observation segments, which phase 0 writes).  Schedule:
 *     bsx_group_step_phase(E, 0)                       lane advance of sweep step 0
 *     bsx_group_step_pipelined(E, O)                   stream of step 0 | advance of step 1
 *     bsx_group_step_pipelined(O, E)                   stream of step 1 | advance of step 2 ...
 * After launch s the TimeStep of step s is complete in the buffers of group (s even ? E : O); the lanes
 * are one advance ahead of it.
 * BSX_EMODE: a two-kernel segment of either group keeps workgroups in the store stream but was set without
 * state_alt — the stream of step s would read the column the advance of step s+1 is writing. */
int bsx_group_step_pipelined(bsx_group_t* streams_of, bsx_group_t* advances_of, void* hip_stream);
/* Diagnostics (ABI v9): with a device buffer of `capacity` >= 3 * (phase-0 workgroups) uint64 (BSX_EINVAL if
 * smaller; the group must be committed), every phase-0 workgroup of a BSX_FAM_SWEEP_MIXED group records
 * buf[3b] = start, buf[3b+1] = end (wall_clock64(): 100 MHz), buf[3b+2] = its segment's family id on every
 * bsx_group_step / step_phase(0) that follows; NULL switches it off.  Where the latency-bound phase 0 of a
 * sweep step spends its time (tools/sweep_phase0_trace.py). */
int bsx_group_trace(bsx_group_t* g, uint64_t* buf, int64_t capacity);
int bsx_group_destroy(bsx_group_t* g);

/* ---- observation adapter ------------------------------------------------------
 * Replaces utils/wrappers.py:150-247 (`ImageObservation.step/reset` -> `to_image(shape, obs)`):
 * every lane's observation [in_rows,in_cols] becomes an image [out_rows,out_cols,tail] (tail =
 * product of the trailing dimensions of `shape`, 1 for a 2-D shape); each plane value is broadcast
 * over the tail.  mode BSX_IMAGE_SMALL is `_small_state_to_image` (:178-204, observation.size
 * <= 4: constant / left-right halves / quadrants, incl. the reference's quadrant order);
 * BSX_IMAGE_BILINEAR is `_interpolate_to_image` (:207-219), i.e. skimage.transform.resize(order=1,
 * mode='reflect') = scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True): per axis
 * cc = (k+0.5)*(in/out)-0.5 mirrored, weights (1-frac, frac), the 4 terms (v*wy)*wx summed in f64 in
 * scipy's order, cast to f32.  An axis that SHRINKS is first passed through skimage's anti-aliasing
 * Gaussian (ABI v8; scipy.ndimage.gaussian_filter, mode='mirror': per element
 * t = in[0]*w[0]; for j = radius..1: t += (in[-j] + in[+j])*w[j] in f64, rounded to f32, rows first
 * then columns): the caller passes the half kernel w[0..radius] it built the way scipy does
 * (sigma = (in/out - 1)/2, radius = int(4*sigma + 0.5), exp(-0.5/sigma^2 * x^2) normalised);
 * radius 0 = that axis is not filtered.
 * obs: f32 [n_lanes, in_rows*in_cols]; image: f32 [n_lanes, out_rows*out_cols*tail], 16-B aligned.
 * Limits: in_rows*in_cols <= 4096, out_rows/out_cols <= 1024, out_rows*out_cols*tail < 2^20,
 * radius <= BSX_IMAGE_MAX_RADIUS. */
#define BSX_IMAGE_SMALL 0
#define BSX_IMAGE_BILINEAR 1
#define BSX_IMAGE_MAX_RADIUS 64
typedef struct {
  int32_t mode;
  int32_t in_rows, in_cols;
  int32_t out_rows, out_cols, tail;
  int32_t radius_y, radius_x;                      /* anti-aliasing filter along rows / columns, 0 = none */
  double gauss_y[BSX_IMAGE_MAX_RADIUS + 1];        /* weights at distance 0..radius_y                     */
  double gauss_x[BSX_IMAGE_MAX_RADIUS + 1];
} bsx_image_t;
int bsx_image_observation(const bsx_image_t* cfg, int64_t n_lanes, const float* obs, float* image,
                          void* hip_stream);

/* ---- misc ---------------------------------------------------------------------------------- */
int bsx_abi_version(void);
/* bsuite_info() through the C ABI (v12): info_out [n_info, B] = the family's info columns as the REFERENCE would report
 * them right now (catch.py:116-117 total_regret, cartpole.py:179-181 raw_return / best_episode, mountain_car.py:99-100
 * raw_return, ...).  Without call->logging some families keep part of an accumulator in their state word until it is
 * cheap to fold (see each family's accounting note above): `folded` != 0 says the columns were maintained that way and
 * the pending part is added from `state` — catch: the packed state column; cartpole (variant 0; swing-up, variant 1,
 * accumulates per step) and mountain_car: the `steps` column; every other family (and folded == 0, i.e. columns kept
 * under call->logging): a plain copy, `state` may be NULL.  Cartpole's column 2 is its internal running
 * episode_return.  Asynchronous on hip_stream like every entry point; info_out may not alias info. */
int bsx_bsuite_info(int32_t family, int32_t variant, int64_t n_lanes, const int32_t* state, const double* info,
                    int32_t n_info, int32_t folded, double* info_out, void* hip_stream);
/* Bytes of bsx_call_t.row_scratch for `n_lanes` lanes of a `family` (BSX_FAM_*) observation row of `obs_numel` floats:
 * PLANES bit planes of ceil(n_lanes / 64) * 2 * obs_numel uint32 (memory_chain 2, umbrella_chain 1) + one f32 column
 * [n_lanes] per genuine float of the row (memory_chain 2: time, query; umbrella_chain 1: time).  0 = this family / row
 * length has no row path (the scratch would be ignored). */
int64_t bsx_row_scratch_bytes(int32_t family, int32_t obs_numel, int64_t n_lanes);
const char* bsx_strerror(int code);
/* Pure-store calibration: writes n_bytes of zeros with the same 16-B cooperative pattern the
 * observation writers use; the measured rate is the practical ceiling for store-bound families. */
int bsx_calib_fill(void* dst, int64_t n_bytes, int32_t nontemporal, void* hip_stream);
/* Copy calibration (ABI v12): reads n_bytes from src (one 16-byte load per thread) and writes them `writes_per_read`
 * (1..3) times to dst, dst + n_bytes, ...: the read/write mix of the small-observation families' eager step (2 = a third
 * read, two thirds written).  (1 + writes_per_read) * n_bytes / time is the ceiling of this box for kernels that mix
 * reads into their writes — lower than the pure-fill rate. */
int bsx_calib_copy(void* dst, const void* src, int64_t n_bytes, int32_t writes_per_read, void* hip_stream);
/* *counter += delta on the stream: advances a device-resident call counter (bsx_stream_t.step_base)
 * so a captured hipGraph of step launches replays with fresh draw-stream coordinates. */
int bsx_counter_add(uint64_t* counter, uint64_t delta, void* hip_stream);
/* Fill words [0,n) of the draw stream for (seed, lane, step, stream_id) — used by tests to pin the
 * device Philox/normal implementation against the oracle's independent one. */
int bsx_stream_dump(uint64_t seed, uint64_t lane0, int64_t n_lanes, uint64_t step, int32_t stream_id,
                    int32_t n_words, uint32_t* words /*[n_lanes,n_words]*/,
                    double* normals /*[n_lanes,n_words/2] or NULL*/, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* BSUITE_AMD_H_ */
