// deep_sea_fam.h — the device side of batched DeepSea (bsuite/environments/deep_sea.py:103-144): the lane
// advance (deep_sea_fam) and the hot-cell decoder of the observation stream (deep_sea_hot).  Shared by
// deep_sea.hip (single-family launches) and pair_mixed.hip (the sweep's mixed two-kernel group).
#ifndef BSX_DEEP_SEA_FAM_H_
#define BSX_DEEP_SEA_FAM_H_

#include "bsx_device.h"

#define DS_RESET_BIT (1 << 17)
// Bit 18 of the packed state: the parity of the call index that will READ the word next — every advance writes
// ((step + 1) & 1) there.  It costs nothing and lets the single-launch step (deep_sea_step1_kernel, deep_sea.hip) tell
// a word its lane's writer has already advanced in this launch from one it has not; nobody else looks at it.
#define DS_TAG_SHIFT 18
#define DS_TAG_BIT (1 << DS_TAG_SHIFT)
#define DS_MAP_WORDS (BSX_DEEP_SEA_MAX_SIZE * BSX_DEEP_SEA_MAX_SIZE / 32)

struct deep_sea_fam {
  struct args {
    bsx_ctl ctl;
    const int32_t* action;
    int32_t* state;
    bsx_timestep_t out;
    double* info;        // [2,B]: total_bad_episodes, denoised_return
    double move_cost;
    double inv_size;
    int32_t size;
    int32_t deterministic;
    uint32_t mapping_bits[DS_MAP_WORDS];
  };
  struct shared { uint32_t map[DS_MAP_WORDS]; };

  __device__ static __forceinline__ void stage(const args& a, shared& s) {
    const int map_words = (a.size * a.size + 31) >> 5;
    for (int w = threadIdx.x; w < map_words; w += BSX_BLOCK) s.map[w] = a.mapping_bits[w];
  }

  // One lane's reset()/step() (base.py:59-65 -> deep_sea.py:110-144).
  // LEAN: counter-based draws only (the MT19937-exact mode is compiled out)
  // commit = false: compute the transition only, leave the bsuite_info columns alone (a thread of the single-launch
  // step that needs the lane's new state but is not the lane's writer)
  // DET = 1: the caller knows the environment is deterministic (and LEAN): no draw exists, the Philox block and the f64
  // normal transform are compiled out — what the per-thread recomputation of the single-launch step can afford.
  // NOMT: counter-based draws in a call that is not lean otherwise (see catch_fam.h)
  template <bool LEAN = false, int DET = -1, bool NOMT = false>
  __device__ static __forceinline__ int advance(const args& a, const shared& s, int64_t i, uint64_t lane,
                                                uint64_t step, int32_t st, int act, int32_t& nst,
                                                double& reward, const bool commit = true) {
    BSX_NO_CONTRACT
    const int N = a.size;
    const bool deterministic = DET == 1 || a.deterministic;
    int row = st & 0xFF, col = (st >> 8) & 0xFF, bad = (st >> 16) & 1;
    int type;
    reward = 0.0;
    if (a.ctl.force_reset || (st & DS_RESET_BIT)) {            // base.py:61-62 -> deep_sea.py:110-114
      row = 0; col = 0; bad = 0;
      type = BSX_FIRST;
    } else {
      const int cell = row * N + col;
      const int mapped = (int)((s.map[cell >> 5] >> (cell & 31)) & 1u);
      const bool right = (act == mapped);                       // deep_sea.py:118
      bsx_draws d;
      bsx_draws_begin<(LEAN || NOMT) ? 0 : -1>(&d, a.ctl, i, lane, step);
      if (col == N - 1 && right) {                              // :121-123
        reward += 1.0;
        if (commit) a.info[a.ctl.n_lanes + i] += 1.0;
      }
      if (!deterministic && row == N - 1 && (col == 0 || col == N - 1))   // :124-126
        reward += bsx_normal(&d);
      if (right) {                                              // :129-132
        // The reference draws rand() here even when deterministic (the value is then unused).  The
        // counter-based stream restarts at every call, so an unused draw leaves no trace and is
        // skipped; the lane's own MT19937 generator (exact mode) must advance, so there it is drawn.
        bool moves = true;
        if (!deterministic || (!LEAN && !NOMT && a.ctl.mt_state != nullptr)) {
          const double u = bsx_uniform(&d);
          moves = (u > a.inv_size) || deterministic;
        }
        if (moves) col = col + 1 > N - 1 ? N - 1 : col + 1;
        reward -= a.move_cost;
      } else {                                                  // :133-136
        if (row == col) bad = 1;
        col = col - 1 < 0 ? 0 : col - 1;
      }
      bsx_draws_end<(LEAN || NOMT) ? 0 : -1>(&d, a.ctl, i);
      row += 1;                                                 // :137
      if (row == N) {                                           // :140-143
        if (bad && commit) a.info[i] += 1.0;
        type = BSX_LAST;
      } else {
        type = BSX_MID;
      }
    }
    nst = row | (col << 8) | (bad << 16) | (type == BSX_LAST ? DS_RESET_BIT : 0) | (int32_t)(((uint32_t)(step + 1) & 1u) << DS_TAG_SHIFT);
    return type;
  }
  template <bool LEAN, bool NOMT>
  __device__ static __forceinline__ int advance_nomt(const args& a, const shared& s, int64_t i, uint64_t lane, uint64_t step,
                                                     int32_t st, int act, int32_t& nst, double& reward) {
    return advance<LEAN, -1, NOMT>(a, s, i, lane, step, st, act, nst, reward);
  }
};

// hot cell of a lane from its packed state (observation stream kernel)
struct deep_sea_hot {
  int N;
  __device__ __forceinline__ void operator()(int32_t st, int& a, int& b) const {
    const int row = st & 0xFF, col = (st >> 8) & 0xFF;
    a = row < N ? row * N + col : -1;     // deep_sea.py:105-107 (terminal observation is all-zero)
    b = -1;
  }
};

#endif  // BSX_DEEP_SEA_FAM_H_
