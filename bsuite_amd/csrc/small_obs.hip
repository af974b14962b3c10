// small_obs.hip — what the small-observation families share on the host side (device code: small_obs.h); their C-ABI entry
// points live one file per family (bandit.hip, memory_chain.hip, umbrella_chain.hip, discounting_chain.hip, cartpole.hip,
// mountain_car.hip):
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
