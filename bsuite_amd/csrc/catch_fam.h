// catch_fam.h — the device side of batched Catch (bsuite/environments/catch.py:68-114): the lane advance
// (catch_fam) and the two-hot decoder of the observation stream (catch_hot).  Shared by catch.hip and
// pair_mixed.hip (the sweep's mixed two-kernel group).
#ifndef BSX_CATCH_FAM_H_
#define BSX_CATCH_FAM_H_

#include "bsx_device.h"

#define CATCH_RESET_BIT (1 << 24)
// Bits 25..31 of the packed state: misses not yet folded into the total_regret column (ABI v10).  A miss costs
// regret 2, a catch 0 (catch.py:92-94), so the column is an exact function of the number of misses: without the
// Logging wrapper (whose rows snapshot the column) the lane counts its misses in the state word it rewrites anyway and
// adds 2*127 to the f64 column once per 127 misses, instead of read-modify-writing the column on every episode end —
// in steady state that touched EVERY line of the column on every call (17 of the advance's 37 MB at 2^20 lanes) and
// put a memory round trip in the middle of every wave's advance.  total_regret = info + 2 * pending.
#define CATCH_PENDING_SHIFT 25
#define CATCH_PENDING_MAX 127

struct catch_fam {
  struct args {
    bsx_ctl ctl;
    const int32_t* action;
    int32_t* state;
    bsx_timestep_t out;
    double* info;        // [1,B]: total_regret
    int32_t rows, columns;
    // a segment of a whole-sweep group whose boards phase 0 writes itself (the fused tile of bsx_fused_tile_kernel)
    // instead of leaving them to the phase-1 store stream: the magic of rows*columns, 0 = not fused
    uint32_t tile_cells_magic;
    int32_t _pad;
  };
  struct shared { int unused; };
  __device__ static __forceinline__ void stage(const args&, shared&) {}

  // LEAN: counter-based draws only (the MT19937-exact mode is compiled out); NOMT: the same for a call that is not
  // lean otherwise (Logging / RewardNoise on the counter-based stream)
  template <bool LEAN = false, bool NOMT = false>
  __device__ static __forceinline__ int advance(const args& a, const shared&, int64_t i, uint64_t lane,
                                                uint64_t step, int32_t st, int act, int32_t& nst,
                                                double& reward) {
    const int rows = a.rows, cols = a.columns;
    int ball_x = st & 0xFF, ball_y = (st >> 8) & 0xFF, paddle_x = (st >> 16) & 0xFF;
    uint32_t pending = ((uint32_t)st >> CATCH_PENDING_SHIFT) & 0x7Fu;
    const bool fold = LEAN || a.ctl.log.steps == nullptr;      // uniform
    int type;
    reward = 0.0;
    if (a.ctl.force_reset || (st & CATCH_RESET_BIT)) {         // catch.py:80-81 -> :68-76
      bsx_draws d;
      bsx_draws_begin<(LEAN || NOMT) ? 0 : -1>(&d, a.ctl, i, lane, step);
      ball_x = (int)bsx_randint(&d, (uint32_t)cols);            // :71
      bsx_draws_end<(LEAN || NOMT) ? 0 : -1>(&d, a.ctl, i);
      ball_y = 0;
      paddle_x = cols / 2;
      type = BSX_FIRST;
    } else {
      if (act < 0 || act > 2) bsx_note_invalid_action(a.ctl, i);   // reference: IndexError (catch.py:84)
      const int dx = act - 1;                                   // _ACTIONS :27
      paddle_x = paddle_x + dx;                                 // :85 np.clip
      paddle_x = paddle_x < 0 ? 0 : (paddle_x > cols - 1 ? cols - 1 : paddle_x);
      ball_y += 1;                                              // :88
      if (ball_y == rows - 1) {                                 // :91-95
        reward = (paddle_x == ball_x) ? 1.0 : -1.0;
        if (!fold) {
          a.info[i] += (1.0 - reward);                          // :94 (per episode under the Logging wrapper)
        } else if (paddle_x != ball_x) {
          // (also tried: a fire-and-forget global_atomic_add_f64 per miss — 100k scattered 8-byte atomics per step cost
          // the L2 more than the round trip cost the wave: 45.0 -> 50.4 us at 2^20 lanes, profiles/r03/ab_info_atomics.log)
          if (++pending == CATCH_PENDING_MAX) { a.info[i] += 2.0 * CATCH_PENDING_MAX; pending = 0; }
        }
        type = BSX_LAST;
      } else {
        type = BSX_MID;                                         // :97
      }
    }
    nst = (int32_t)((uint32_t)(ball_x | (ball_y << 8) | (paddle_x << 16) | (type == BSX_LAST ? CATCH_RESET_BIT : 0)) |
                    (pending << CATCH_PENDING_SHIFT));
    return type;
  }
  template <bool LEAN, bool NOMT>
  __device__ static __forceinline__ int advance_nomt(const args& a, const shared& s, int64_t i, uint64_t lane, uint64_t step,
                                                     int32_t st, int act, int32_t& nst, double& reward) {
    return advance<LEAN, NOMT>(a, s, i, lane, step, st, act, nst, reward);
  }
};

struct catch_hot {
  int rows, cols;
  __device__ __forceinline__ void operator()(int32_t st, int& a, int& b) const {
    a = ((st >> 8) & 0xFF) * cols + (st & 0xFF);          // ball   (catch.py:111)
    b = (rows - 1) * cols + ((st >> 16) & 0xFF);          // paddle (catch.py:112)
  }
};

#endif  // BSX_CATCH_FAM_H_
