// misc.hip — ABI version / error strings, the pure-store calibration kernel and the draw-stream
// dump used by the tests to pin the device Philox / normal transform against the oracle.
#include <algorithm>

#include "bsx_host.h"
#include "pair_mixed.h"

extern "C" int bsx_abi_version(void) { return BSX_ABI_VERSION; }

extern "C" const char* bsx_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case BSX_EINVAL: return "invalid scalar argument";
    case BSX_ENULL: return "required pointer is NULL";
    case BSX_EALIGN: return "observation buffer (or row scratch) is not 16-byte aligned";
    case BSX_ERANGE: return "parameter outside the supported range of this family";
    case BSX_EMODE: return "combination not available (randn in MT19937-exact mode; obs_paint with a rollout, a group or a family without a board)";
    case BSX_ENOMEM: return "host allocation failed";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown bsx error";
  }
}

// Pure-store calibration: one 16-byte store per thread, block b writes the 4 KiB run b, no loop —
// the fastest plain fill shape found on MI355X (it matches torch's fill kernel at ~6.8-6.9 TB/s on
// boxes where a 4-stores-per-thread grid-stride fill reaches 5.8-6.0; profiles/r01/
// store_calibration4_noloop.log).  What this reaches is the practical ceiling for a dense
// observation stream on the box it runs on.
template <bool NT>
__global__ void __launch_bounds__(BSX_BLOCK) calib_fill_kernel(bsx_f4* __restrict__ p, int64_t n16) {
  const bsx_f4 z = {0.f, 0.f, 0.f, 0.f};
  const int64_t i = (int64_t)blockIdx.x * BSX_BLOCK + threadIdx.x;
  if (i < n16) { if (NT) __builtin_nontemporal_store(z, &p[i]); else p[i] = z; }
}

extern "C" int bsx_calib_fill(void* dst, int64_t n_bytes, int32_t nontemporal, void* hip_stream) {
  if (dst == nullptr) return BSX_ENULL;
  if (n_bytes < 0 || (n_bytes & 15) != 0) return BSX_EINVAL;
  if ((reinterpret_cast<uintptr_t>(dst) & 15u) != 0) return BSX_EALIGN;
  if (n_bytes == 0) return 0;
  const int64_t n16 = n_bytes / 16;
  const int64_t blocks = (n16 + BSX_BLOCK - 1) / BSX_BLOCK;
  if (blocks > 0x7FFFFFFF) return BSX_EINVAL;
  hipStream_t st = (hipStream_t)hip_stream;
  if (nontemporal) calib_fill_kernel<true><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>((bsx_f4*)dst, n16);
  else calib_fill_kernel<false><<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, st>>>((bsx_f4*)dst, n16);
  return bsx_launch_status();
}

// Copy calibration: the access mix of the small-observation families' eager step — it reads a third of what it writes
// (state in; state, TimeStep scalars and the row out).  One 16-byte load per thread, W 16-byte stores (to W regions n_bytes
// apart), no loop, blocks in address order: the rate (R + W bytes per second) such a mix reaches on this box is the
// ceiling for a kernel that mixes reads into its writes, as the fill rate is for a pure store stream.
template <int W>
__global__ void __launch_bounds__(BSX_BLOCK) calib_copy_kernel(const bsx_f4* __restrict__ src, bsx_f4* __restrict__ dst, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * BSX_BLOCK + threadIdx.x;
  if (i < n16) {
    const bsx_f4 v = src[i];
#pragma unroll
    for (int w = 0; w < W; ++w) dst[(int64_t)w * n16 + i] = v;
  }
}

extern "C" int bsx_calib_copy(void* dst, const void* src, int64_t n_bytes, int32_t writes_per_read, void* hip_stream) {
  if (dst == nullptr || src == nullptr) return BSX_ENULL;
  if (n_bytes < 0 || (n_bytes & 15) != 0 || writes_per_read < 1 || writes_per_read > 3) return BSX_EINVAL;
  if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) != 0) return BSX_EALIGN;
  if (n_bytes == 0) return 0;
  const int64_t n16 = n_bytes / 16;
  const int64_t blocks = (n16 + BSX_BLOCK - 1) / BSX_BLOCK;
  if (blocks > 0x7FFFFFFF) return BSX_EINVAL;
  hipStream_t st = (hipStream_t)hip_stream;
  const dim3 g((unsigned)blocks), b(BSX_BLOCK);
  if (writes_per_read == 1) calib_copy_kernel<1><<<g, b, 0, st>>>((const bsx_f4*)src, (bsx_f4*)dst, n16);
  else if (writes_per_read == 2) calib_copy_kernel<2><<<g, b, 0, st>>>((const bsx_f4*)src, (bsx_f4*)dst, n16);
  else calib_copy_kernel<3><<<g, b, 0, st>>>((const bsx_f4*)src, (bsx_f4*)dst, n16);
  return bsx_launch_status();
}

__global__ void counter_add_kernel(uint64_t* counter, uint64_t delta) { *counter += delta; }

extern "C" int bsx_counter_add(uint64_t* counter, uint64_t delta, void* hip_stream) {
  if (counter == nullptr) return BSX_ENULL;
  counter_add_kernel<<<dim3(1), dim3(1), 0, (hipStream_t)hip_stream>>>(counter, delta);
  return bsx_launch_status();
}

// bsuite_info() for callers of the C ABI: the columns as the reference would report them right now.  catch counts its
// misses in spare bits of the packed state word and cartpole / mountain_car fold exact per-episode sums into their columns
// only when an episode ends (include/bsuite_amd.h, each family's "Accounting" note): this adds the part that is still
// pending in the lane's state — what the Python classes' bsuite_info() does on the host side of the boundary.
//   bsuite/environments/catch.py:116-117, cartpole.py:179-181, mountain_car.py:99-100
__global__ void __launch_bounds__(BSX_BLOCK) bsuite_info_kernel(int32_t family, int32_t variant, int64_t n_lanes,
                                                                const int32_t* __restrict__ state,
                                                                const double* __restrict__ info, int32_t n_info,
                                                                int32_t folded, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * BSX_BLOCK + threadIdx.x;
  if (i >= n_lanes) return;
  double pend0 = 0.0, pend2 = 0.0;
  if (folded) {
    if (family == BSX_FAM_CATCH) {
      pend0 = 2.0 * (double)(((uint32_t)state[i] >> 25) & 0x7Fu);          // a miss costs regret 2 (catch.py:92-94)
    } else if ((family == BSX_FAM_CARTPOLE && variant == 0) || family == BSX_FAM_MOUNTAIN_CAR) {
      const int32_t sk = state[i];                                         // steps = k | reset_next << 30
      const double k = (sk >> 30) & 1 ? 0.0 : (double)(sk & 0x3FFFFFFF);   // the running episode has paid k rewards of +1 / -1
      pend0 = family == BSX_FAM_CARTPOLE ? k : -k;
      pend2 = family == BSX_FAM_CARTPOLE ? k : 0.0;                        // (cartpole's internal episode_return column)
    }
  }
  for (int c = 0; c < n_info; ++c)
    out[(int64_t)c * n_lanes + i] = info[(int64_t)c * n_lanes + i] + (c == 0 ? pend0 : c == 2 ? pend2 : 0.0);
}

extern "C" int bsx_bsuite_info(int32_t family, int32_t variant, int64_t n_lanes, const int32_t* state, const double* info,
                               int32_t n_info, int32_t folded, double* info_out, void* hip_stream) {
  if (family < BSX_FAM_DEEP_SEA || family > BSX_FAM_MNIST || n_lanes < 0 || n_info < 0 || n_info > 8) return BSX_EINVAL;
  if (n_lanes == 0 || n_info == 0) return 0;
  if (info == nullptr || info_out == nullptr) return BSX_ENULL;
  const bool pending = folded && (family == BSX_FAM_CATCH || (family == BSX_FAM_CARTPOLE && variant == 0) || family == BSX_FAM_MOUNTAIN_CAR);
  if (pending && state == nullptr) return BSX_ENULL;
  const int64_t blocks = (n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  if (blocks > 0x7FFFFFFF) return BSX_EINVAL;
  bsuite_info_kernel<<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, (hipStream_t)hip_stream>>>(
      family, variant, n_lanes, state, info, n_info, pending ? 1 : 0, info_out);
  return bsx_launch_status();
}

__global__ void __launch_bounds__(BSX_BLOCK) stream_dump_kernel(uint64_t seed, uint64_t lane0, int64_t n_lanes,
                                                                uint64_t step, uint32_t stream_id, int n_words,
                                                                uint32_t* words, double* normals) {
  int64_t i = (int64_t)blockIdx.x * BSX_BLOCK + threadIdx.x;
  if (i >= n_lanes) return;
  bsx_draws d;
  bsx_draws_init(&d, seed, lane0 + (uint64_t)i, step, stream_id);
  for (int w = 0; w < n_words; ++w) words[i * n_words + w] = bsx_word(&d);
  if (normals != nullptr) {
    bsx_draws_init(&d, seed, lane0 + (uint64_t)i, step, stream_id);
    for (int w = 0; w < n_words / 2; ++w) normals[i * (n_words / 2) + w] = bsx_normal(&d);
  }
}

extern "C" int bsx_stream_dump(uint64_t seed, uint64_t lane0, int64_t n_lanes, uint64_t step, int32_t stream_id,
                               int32_t n_words, uint32_t* words, double* normals, void* hip_stream) {
  if (words == nullptr) return BSX_ENULL;
  if (n_lanes < 0 || n_words < 0 || n_words > 1024) return BSX_EINVAL;
  if (n_lanes == 0 || n_words == 0) return 0;
  const int64_t blocks = (n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  stream_dump_kernel<<<dim3((unsigned)blocks), dim3(BSX_BLOCK), 0, (hipStream_t)hip_stream>>>(
      seed, lane0, n_lanes, step, (uint32_t)stream_id, n_words, words, normals);
  return bsx_launch_status();
}

// ------------------------------------------------------------------------------ grouped launch
extern "C" int bsx_group_create(int32_t family, int32_t n_segments, bsx_group_t** group) {
  if (group == nullptr) return BSX_ENULL;
  if (family < BSX_FAM_DEEP_SEA || family > BSX_FAM_SWEEP_MIXED || n_segments < 1 || n_segments > (1 << 20)) return BSX_EINVAL;
  bsx_group* g = nullptr;
  try {                                   // no C++ exception may cross the C boundary
    g = new bsx_group();
    g->family = family; g->n = n_segments;
    g->blocks.assign(n_segments, 0); g->blocks2.assign(n_segments, 0); g->is_set.assign(n_segments, 0);
  } catch (...) {
    delete g;
    return BSX_ENOMEM;
  }
  *group = g;
  return 0;
}

static void group_free_device(bsx_group* g) {
  void** ptrs[] = {&g->d_args, &g->d_args2, (void**)&g->d_start, (void**)&g->d_start2, (void**)&g->d_map, (void**)&g->d_map2,
                   (void**)&g->d_tags, (void**)&g->d_ticket};
  for (void** p : ptrs) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
}

static int upload(const void* src, size_t bytes, void** dst) {
  if (bytes == 0) { *dst = nullptr; return 0; }
  hipError_t e = hipMalloc(dst, bytes);
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
}

static int group_commit(bsx_group* g);

extern "C" int bsx_group_commit(bsx_group_t* g) {
  if (g == nullptr) return BSX_ENULL;
  if (g->committed || g->launch == nullptr) return BSX_EINVAL;
  int rc;
  try {
    rc = group_commit(g);
  } catch (...) {
    rc = BSX_ENOMEM;
  }
  if (rc != 0) group_free_device(g);      // a failed commit leaves nothing behind and may be retried
  return rc;
}

static int group_commit(bsx_group* g) {
  for (int i = 0; i < g->n; ++i) if (!g->is_set[i]) return BSX_EINVAL;
  std::vector<int32_t> start(g->n + 1, 0), start2(g->n + 1, 0);
  int64_t t1 = 0, t2 = 0;
  for (int i = 0; i < g->n; ++i) {
    start[i] = (int32_t)t1; start2[i] = (int32_t)t2;
    t1 += g->blocks[i]; t2 += g->blocks2[i];
    if (t1 > 0x7FFFFFFF || t2 > 0x7FFFFFFF) return BSX_EINVAL;
  }
  start[g->n] = (int32_t)t1; start2[g->n] = (int32_t)t2;
  g->total_blocks = t1; g->total_blocks2 = t2;
  int rc = upload(g->args.data(), g->args.size(), &g->d_args);
  if (rc == 0) rc = upload(g->args2.data(), g->args2.size(), &g->d_args2);
  if (rc == 0 && !g->tags.empty()) rc = upload(g->tags.data(), g->tags.size() * 4, (void**)&g->d_tags);
  if (rc == 0 && g->family == BSX_FAM_SWEEP_MIXED) {   // two-level retirement ticket: word 0 + 64 shards, one line each
    const std::vector<uint32_t> zeros(32 * 65, 0u);
    rc = upload(zeros.data(), zeros.size() * 4, (void**)&g->d_ticket);
  }
  if (rc == 0) rc = upload(start.data(), start.size() * 4, (void**)&g->d_start);
  if (rc == 0) rc = upload(start2.data(), start2.size() * 4, (void**)&g->d_start2);
  // (segment, local block) of every workgroup: one load per workgroup instead of a binary search;
  // launches beyond 2^24 workgroups (128 MiB of map) keep the search.
  static const int use_map = bsx_env_int("BSX_GROUP_MAP", 1);
  for (int pass = 0; pass < 2 && rc == 0 && use_map; ++pass) {
    const int64_t total = pass == 0 ? t1 : t2;
    const std::vector<int32_t>& blocks = pass == 0 ? g->blocks : g->blocks2;
    if (total == 0 || total > (1 << 24)) continue;
    std::vector<int2> map((size_t)total);
    size_t w = 0;
    for (int i = 0; i < g->n; ++i)
      // the segment's family tag rides in the top byte of the entry (mixed groups): the workgroup then needs ONE
      // load, not map entry -> tag, before it can fetch its argument slot
      for (int32_t b = 0; b < blocks[i]; ++b) {
        map[w].x = i | ((g->tags.empty() ? 0 : ((g->tags[i] + 1) & 0x7F)) << 24);   // tag + 1; 0 = none
        map[w].y = b;
        ++w;
      }
    rc = upload(map.data(), map.size() * sizeof(int2), (void**)(pass == 0 ? &g->d_map : &g->d_map2));
  }
  if (rc != 0) return rc;
  g->stream_without_alt = false;
  for (uint8_t f : g->needs_alt) g->stream_without_alt = g->stream_without_alt || f != 0;
  // the split step needs the segments with a share of the store stream to be the tail of the phase-0 grid
  g->rows_sorted.clear();
  for (const void* p : g->row_scratch) if (p != nullptr) g->rows_sorted.push_back(reinterpret_cast<uintptr_t>(p));
  std::sort(g->rows_sorted.begin(), g->rows_sorted.end());
  g->split_block = -1;
  g->split_round = -1;
  if (g->family == BSX_FAM_SWEEP_MIXED) {
    int first = g->n;
    while (first > 0 && g->blocks2[first - 1] > 0) --first;
    bool tail_only = true;
    for (int i = 0; i < first; ++i) tail_only = tail_only && g->blocks2[i] == 0;
    if (tail_only) g->split_block = start[first];
  }
  g->committed = true;
  return 0;
}

extern "C" int bsx_group_step(bsx_group_t* g, void* hip_stream) {
  if (g == nullptr) return BSX_ENULL;
  if (!g->committed) return BSX_EINVAL;
  return g->launch(g, -1, (hipStream_t)hip_stream);
}

extern "C" int bsx_group_phases(const bsx_group_t* g) { return g == nullptr ? BSX_ENULL : g->n_phases; }

extern "C" int bsx_group_step_phase(bsx_group_t* g, int32_t phase, void* hip_stream) {
  if (g == nullptr) return BSX_ENULL;
  if (!g->committed || phase < 0 || phase >= g->n_phases) return BSX_EINVAL;
  return g->launch(g, phase, (hipStream_t)hip_stream);
}

extern "C" int bsx_group_step_split(bsx_group_t* g, void* hip_stream) {
  if (g == nullptr) return BSX_ENULL;
  if (!g->committed) return BSX_EINVAL;
  if (g->family != BSX_FAM_SWEEP_MIXED || g->split_block < 0) return BSX_EMODE;
  return bsx_sweep_launch_split(g, (hipStream_t)hip_stream);
}

extern "C" int bsx_group_step_pipelined(bsx_group_t* streams_of, bsx_group_t* advances_of, void* hip_stream) {
  if (streams_of == nullptr || advances_of == nullptr) return BSX_ENULL;
  if (!streams_of->committed || !advances_of->committed || streams_of->family != BSX_FAM_SWEEP_MIXED ||
      advances_of->family != BSX_FAM_SWEEP_MIXED || streams_of->n != advances_of->n ||
      streams_of->shared_counter != advances_of->shared_counter)
    return BSX_EINVAL;
  if (streams_of->stream_without_alt || advances_of->stream_without_alt) return BSX_EMODE;
  // a chain segment on the row path must not share its row scratch between the two groups (the stream of step s would
  // decode the rows the advance of step s+1 is writing) — checked on EVERY call and against every segment of the other
  // group (one merge over the two sorted lists frozen at commit: no cached verdict that a recycled group address could outlive)
  for (size_t i = 0, j = 0; i < streams_of->rows_sorted.size() && j < advances_of->rows_sorted.size();) {
    const uintptr_t p = streams_of->rows_sorted[i], q = advances_of->rows_sorted[j];
    if (p == q) return BSX_EMODE;
    if (p < q) ++i; else ++j;
  }
  return bsx_sweep_launch_pipelined(streams_of, advances_of, (hipStream_t)hip_stream);
}

extern "C" int bsx_group_trace(bsx_group_t* g, uint64_t* buf, int64_t capacity) {
  if (g == nullptr) return BSX_ENULL;
  if (g->family != BSX_FAM_SWEEP_MIXED) return BSX_EMODE;
  if (buf != nullptr && (!g->committed || capacity < 3 * g->total_blocks)) return BSX_EINVAL;
  g->trace = buf;
  return 0;
}

extern "C" int bsx_group_destroy(bsx_group_t* g) {
  if (g == nullptr) return 0;
  group_free_device(g);
  delete g;
  return 0;
}
