// chain_rows.h — host side of the wide-row path of memory_chain / umbrella_chain (bsx_call_t.row_scratch, bsx_rows.h).
#ifndef BSX_CHAIN_ROWS_H_
#define BSX_CHAIN_ROWS_H_

#include "small_obs.h"

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

#endif  // BSX_CHAIN_ROWS_H_
