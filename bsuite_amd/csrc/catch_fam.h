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
// ball_x, ball_y, paddle_x are below 64 (rows, columns <= 64): the two top bits of each byte of the state word are the
// library's (ABI v12).  Bit 7: the parity of the call index that will READ the word next — every advance writes
// ((step + 1) & 1) there, like deep_sea's bit 18 — which lets the single-launch step (catch_step1_kernel, catch.hip) tell a
// word its lane's writer has already advanced in this launch from one it has not.  Bits 6, 14, 15, 22: the ball column
// of the lane's NEXT episode, drawn ahead of time by that kernel's writer thread (bit 6 says a value is parked; bits
// 14, 15, 22 hold it: columns <= 8) so that the threads that merely need the lane's new state on the call that resets it
// do not each walk a Philox block.  The draw itself is the one the stream specifies for that call — (seed, lane, the
// reset call's index), word 0 — computed early, not changed.  Every other advance writes the parked bits as 0.
#define CATCH_FIELD_MASK 0x3F
#define CATCH_TAG_SHIFT 7
#define CATCH_TAG_BIT (1 << CATCH_TAG_SHIFT)
#define CATCH_PARK_VALID (1 << 6)
#define CATCH_LIB_BITS (CATCH_TAG_BIT | CATCH_PARK_VALID | (3 << 14) | (3 << 22))
__host__ __device__ __forceinline__ int32_t catch_park_bits(uint32_t x) {          // x < 8
  return (int32_t)(CATCH_PARK_VALID | ((x & 3u) << 14) | (((x >> 2) & 1u) << 22));
}
__host__ __device__ __forceinline__ uint32_t catch_parked(int32_t st) { return (((uint32_t)st >> 14) & 3u) | ((((uint32_t)st >> 22) & 1u) << 2); }

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
  // STEP1: the single-launch step (catch_step1_kernel): a reset takes the ball column parked in the state word when there
  // is one; `commit` = this thread is the lane's writer (it alone touches the info column and the error word, and — on
  // the calls whose index is a multiple of `rows`, the episode period — parks the draw of the lane's next reset).
  template <bool LEAN = false, bool NOMT = false, bool STEP1 = false>
  __device__ static __forceinline__ int advance(const args& a, const shared&, int64_t i, uint64_t lane,
                                                uint64_t step, int32_t st, int act, int32_t& nst,
                                                double& reward, const bool commit = true, const bool park_now = false) {
    const int rows = a.rows, cols = a.columns;
    int ball_x = st & CATCH_FIELD_MASK, ball_y = (st >> 8) & CATCH_FIELD_MASK, paddle_x = (st >> 16) & CATCH_FIELD_MASK;
    uint32_t pending = ((uint32_t)st >> CATCH_PENDING_SHIFT) & 0x7Fu;
    const bool fold = LEAN || a.ctl.log.steps == nullptr;      // uniform
    int type;
    reward = 0.0;
    if (a.ctl.force_reset || (st & CATCH_RESET_BIT)) {         // catch.py:80-81 -> :68-76
      if (STEP1 && (st & CATCH_PARK_VALID)) {
        ball_x = (int)catch_parked(st);                         // :71, drawn ahead (the same word of the same stream)
      } else {
        bsx_draws d;
        bsx_draws_begin<(LEAN || NOMT) ? 0 : -1>(&d, a.ctl, i, lane, step);
        ball_x = (int)bsx_randint(&d, (uint32_t)cols);          // :71
        bsx_draws_end<(LEAN || NOMT) ? 0 : -1>(&d, a.ctl, i);
      }
      ball_y = 0;
      paddle_x = cols / 2;
      type = BSX_FIRST;
    } else {
      if ((act < 0 || act > 2) && commit) bsx_note_invalid_action(a.ctl, i);   // reference: IndexError (catch.py:84)
      const int dx = act - 1;                                   // _ACTIONS :27
      paddle_x = paddle_x + dx;                                 // :85 np.clip
      paddle_x = paddle_x < 0 ? 0 : (paddle_x > cols - 1 ? cols - 1 : paddle_x);
      ball_y += 1;                                              // :88
      if (ball_y == rows - 1) {                                 // :91-95
        reward = (paddle_x == ball_x) ? 1.0 : -1.0;
        if (!commit) {                                          // (a reader of the single-launch step: the transition only)
        } else if (!fold) {
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
                    (pending << CATCH_PENDING_SHIFT) | (((uint32_t)(step + 1) & 1u) << CATCH_TAG_SHIFT));
    if (STEP1 && commit && park_now) {
      // the lane's next reset happens on call `at`: episodes take exactly `rows` calls (a reset + rows - 1 steps, catch.py:88-97)
      const uint64_t at = step + (type == BSX_LAST ? 1ull : (uint64_t)(rows - ball_y));
      bsx_draws d;
      bsx_draws_init(&d, a.ctl.seed, lane, at, BSX_STREAM_ENV);
      nst |= catch_park_bits(bsx_randint(&d, (uint32_t)cols));
    }
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
    a = ((st >> 8) & CATCH_FIELD_MASK) * cols + (st & CATCH_FIELD_MASK);   // ball   (catch.py:111)
    b = (rows - 1) * cols + ((st >> 16) & CATCH_FIELD_MASK);               // paddle (catch.py:112)
  }
};

#endif  // BSX_CATCH_FAM_H_
