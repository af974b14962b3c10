// pair_mixed.h — host interface of the mixed two-kernel group (BSX_FAM_PAIR_MIXED, pair_mixed.hip).
#ifndef BSX_PAIR_MIXED_H_
#define BSX_PAIR_MIXED_H_

#include "bsx_host.h"

// Records one segment of deep_sea / catch / mnist in a BSX_FAM_PAIR_MIXED group: `adv` is the family's
// advance-kernel argument struct, `str` its observation-stream argument struct (both copied verbatim
// into fixed-stride slots of the group's device tables), blocks1 / blocks2 their workgroup counts.
int bsx_pair_mixed_put(bsx_group* g, int32_t family, int32_t index, const bsx_call_t* call,
                       const void* adv, size_t adv_size, const void* str, size_t str_size,
                       uint64_t blocks1, uint64_t blocks2);

#endif  // BSX_PAIR_MIXED_H_
