// bsx_host.h — host-side glue shared by the C-ABI entry points (argument checks, launch math).
#ifndef BSX_HOST_H_
#define BSX_HOST_H_

#include <stdlib.h>

#include "bsx_device.h"

static inline int bsx_check_call(const bsx_call_t* call, const void* action, const bsx_timestep_t& out) {
  if (call == nullptr) return BSX_ENULL;
  if (call->n_lanes < 0 || call->n_lanes > ((int64_t)1 << 40)) return BSX_EINVAL;
  if (call->n_lanes == 0) return 0;
  if (out.reward == nullptr || out.discount == nullptr || out.step_type == nullptr ||
      out.observation == nullptr)
    return BSX_ENULL;
  if (action == nullptr && !call->force_reset) return BSX_ENULL;
  if ((reinterpret_cast<uintptr_t>(out.observation) & 15u) != 0) return BSX_EALIGN;
  if (call->wrap.kind < BSX_WRAP_NONE || call->wrap.kind > BSX_WRAP_NOISE) return BSX_EINVAL;
  return 0;
}

static inline bsx_ctl bsx_make_ctl(const bsx_call_t* call) {
  bsx_ctl c;
  c.n_lanes = call->n_lanes;
  c.seed = call->stream.seed;
  c.lane_offset = call->stream.lane_offset;
  c.step_index = call->stream.step_index;
  c.step_base = call->stream.step_base;
  c.counters = call->counters;
  c.wrap_param = call->wrap.param;
  c.wrap_seed = call->wrap.seed;
  c.wrap_kind = call->wrap.kind;
  c.force_reset = call->force_reset;
  return c;
}

// magic for q = n / d via __umulhi(n, magic): exact for n < 2^20, d <= 4096
static inline uint32_t bsx_div_magic(uint32_t d) { return (uint32_t)((0x100000000ull / d) + 1ull); }

static inline int bsx_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v != nullptr && *v != '\0') ? atoi(v) : dflt;
}

// Launch geometry of bsx_hot_stream_kernel for n_lanes lanes of `cells` floats.
static inline int bsx_stream_grid(int64_t n_lanes, uint32_t cells, uint32_t* subs_per_super, int64_t* blocks) {
  const uint32_t chunks_per_super = (BSX_SUPER * cells + 3u) >> 2;
  *subs_per_super = (chunks_per_super + BSX_STREAM_CHUNKS - 1) / BSX_STREAM_CHUNKS;
  const int64_t supers = (n_lanes + BSX_SUPER - 1) / BSX_SUPER;
  *blocks = supers * (int64_t)(*subs_per_super);
  return *blocks > 0x7FFFFFFF ? BSX_EINVAL : 0;
}

static inline int bsx_launch_status() { return (int)hipGetLastError(); }

#endif  // BSX_HOST_H_
