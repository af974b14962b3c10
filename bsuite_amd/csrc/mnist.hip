// mnist.hip — batched MNIST contextual bandit for gfx950.  Replaces
// bsuite/environments/mnist.py:61-75 (`_reset`, `_step`) + base.py:54-65 auto-reset.
//
// Episode = reset (show image idx = randint(num_data), obs = int8 pixels / 255 in f32) + one step
// (reward +-1 by action == label, LAST, obs = zeros).  Per call and lane: 3136 B of observation
// stores, on FIRST calls plus a 784 B gather from the image table (L2-resident when small; the real
// 47 MB dataset sits in the Infinity Cache).
//   advance  mnist_advance_kernel: one lane per thread, packed state = idx | label | flags
//   observe  mnist_observe_kernel: pure 16-byte store stream over [B x 784] f32; a chunk is four
//            table bytes mapped through a per-wave 256-entry table in LDS, or zero (mnist_fam.h).
// The table is np.float32(int8(b)) / 255 — the int8 parsing quirk of the reference (datasets.py:55-56)
// and its f32 division, bit for bit: numpy's values from the host are recognised and then computed on
// the device (bsx_mnist_pixel_value, exact for all 256 bytes); any other table is read as given.
#include "bsx_host.h"
#include "mnist_fam.h"
#include "pair_mixed.h"

__global__ void __launch_bounds__(BSX_BLOCK) mnist_advance_kernel(const mnist_args a) {
  __shared__ unsigned int s_cnt[2];
  mnist_advance_body(a, blockIdx.x, s_cnt);
}

__global__ void __launch_bounds__(BSX_BLOCK) mnist_advance_group_kernel(const mnist_args* __restrict__ table,
                                                                        const bsx_group_index gi) {
  __shared__ unsigned int s_cnt[2];
  const bsx_group_slot w = bsx_group_find(gi, (int)blockIdx.x);
  mnist_advance_body(table[w.seg], w.block, s_cnt);
}

template <int K>
__global__ void __launch_bounds__(BSX_BLOCK) mnist_observe_kernel(const mnist_observe_args a) {
  __shared__ float s_lut[MNIST_LUT_FLOATS];
  mnist_observe_body<K>(a, blockIdx.x, s_lut);
}

template <int K>
__global__ void __launch_bounds__(BSX_BLOCK) mnist_observe_group_kernel(const mnist_observe_args* __restrict__ table,
                                                                        const bsx_group_index gi) {
  __shared__ float s_lut[MNIST_LUT_FLOATS];
  const bsx_group_slot w = bsx_group_find(gi, (int)blockIdx.x);
  mnist_observe_body<K>(table[w.seg], w.block, s_lut);
}

// 4 KiB-runs per wave: 16 KiB per workgroup is a sharp optimum of the straight-line body (K = 3: 5.95, 4: 7.0, 5: 5.9, 8: 5.9 TB/s
// at 2^20 lanes with 96 images; 4 stays best up to 20 000 images = 15.7 MB; profiles/r06/mnist_stream_microbench_2.log,
// _8_table_sizes.log).  The REAL dataset (60 000 images, 47 MB) is larger than the 32 MiB of L2 the chip has: the stream's own
// stores then evict the table rows between their uses, and NON-TEMPORAL stores are what helps — bare pattern on a 47 MB
// table: r05 body 3.9 TB/s, this body 5.0-5.2, six KiB-runs 5.3-5.4, **four + nt stores 5.8**; prefetching the rows a later
// workgroup of the same XCD will gather: 4.0-4.2, worse (_7_table47MB.log, _9_table47MB_nt_prefetch.log, _10_nt_by_table.log).
#define MNIST_K 4
#define MNIST_BIG_TABLE_BYTES (32ll << 20)

static int mnist_make(const bsx_mnist_t* cfg, const bsx_call_t* call, const int32_t* action, int32_t* state,
                      bsx_timestep_t out, double* info, mnist_args* a, mnist_observe_args* o) {
  if (cfg == nullptr) return BSX_ENULL;
  int rc = bsx_check_call(call, action, out);
  if (rc != 0) return rc;
  if (cfg->num_data < 1 || cfg->num_data > (1 << 24) || cfg->num_pixels < 4 || cfg->num_pixels > 4096 ||
      (cfg->num_pixels & 3) != 0)
    return BSX_ERANGE;
  if (call->n_lanes > 0 && (state == nullptr || info == nullptr || cfg->images == nullptr || cfg->labels == nullptr))
    return BSX_ENULL;
  if ((reinterpret_cast<uintptr_t>(cfg->images) & 3u) != 0) return BSX_EALIGN;
  a->ctl = bsx_make_ctl(call);
  a->action = action; a->state = state; a->out = out; a->info = info;
  a->images = cfg->images; a->labels = cfg->labels; a->num_data = cfg->num_data; a->num_pixels = cfg->num_pixels;
  o->obs = out.observation; o->state = state; o->images = cfg->images; o->n_lanes = call->n_lanes;
  o->cells = (uint32_t)cfg->num_pixels; o->cells_magic = bsx_div_magic(o->cells); o->dv = bsx_make_div64(o->cells);
  // the reference's table (np.float32(int8) / 255) is recognised and then COMPUTED per wave (bsx_mnist_pixel_value, exact);
  // any other table is read from the arguments
  o->arith = 1;
  o->nt = (int64_t)cfg->num_data * (int64_t)cfg->num_pixels > MNIST_BIG_TABLE_BYTES;
  for (int k = 0; k < 256; ++k) {
    o->lut[k] = cfg->pixel_lut[k];
    const float want = bsx_mnist_pixel_value((uint32_t)k, 0);
    if (memcmp(&want, &cfg->pixel_lut[k], sizeof(float)) != 0) o->arith = 0;
  }
  return 0;
}

extern "C" int bsx_mnist_step(const bsx_mnist_t* cfg, const bsx_call_t* call, const int32_t* action,
                              int32_t* state, bsx_timestep_t out, double* info) {
  mnist_args a;
  mnist_observe_args o;
  int rc = mnist_make(cfg, call, action, state, out, info, &a, &o);
  if (rc != 0) return rc;
  if (call->n_lanes == 0) return 0;
  hipStream_t st = (hipStream_t)call->hip_stream;
  const int64_t blocks_a = (call->n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  const uint64_t blocks_o = bsx_flat_blocks((uint64_t)call->n_lanes * o.cells, MNIST_K);
  if (blocks_a > 0x7FFFFFFF || blocks_o > 0x7FFFFFFFull) return BSX_EINVAL;
  const int n_steps = bsx_n_steps(call);
  for (int t = 0; t < n_steps; ++t) {       // rollout: the kernel pair once per step
    const int64_t off = (int64_t)t * call->n_lanes;
    a.ctl.step_index = call->stream.step_index + (uint64_t)t;
    a.ctl.reward_f64 = call->reward_f64 ? call->reward_f64 + off : nullptr;
    a.action = action ? action + off : action;
    a.out.reward = out.reward + off; a.out.discount = out.discount + off; a.out.step_type = out.step_type + off;
    mnist_advance_kernel<<<dim3((unsigned)blocks_a), dim3(BSX_BLOCK), 0, st>>>(a);
    o.obs = out.observation + off * (int64_t)o.cells;
    mnist_observe_kernel<MNIST_K><<<dim3((unsigned)blocks_o), dim3(BSX_BLOCK), 0, st>>>(o);
  }
  return bsx_launch_status();
}

static int mnist_group_launch(bsx_group* g, int phase, hipStream_t st) {
  if (phase != 1)
    mnist_advance_group_kernel<<<dim3((unsigned)g->total_blocks), dim3(BSX_BLOCK), 0, st>>>(
        (const mnist_args*)g->d_args, g->index1());
  if (phase == 0) return (int)hipGetLastError();
  mnist_observe_group_kernel<PAIR_MNIST_K><<<dim3((unsigned)g->total_blocks2), dim3(BSX_BLOCK), 0, st>>>(
      (const mnist_observe_args*)g->d_args2, g->index2());
  return (int)hipGetLastError();
}

extern "C" int bsx_group_set_mnist(bsx_group_t* g, int32_t index, const bsx_mnist_t* cfg, const bsx_call_t* call,
                                   const int32_t* action, int32_t* state, bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  int rc;
  mnist_args a;
  mnist_observe_args o;
  if (bsx_is_mixed_pair_group(g)) {            // one segment of the mixed two-kernel group (PAIR_MNIST_K x 4 KiB runs)
    rc = mnist_make(cfg, call, action, state, out, info, &a, &o);
    if (rc != 0) return rc;
    a.ctl.state_in = call->state_alt;            // pipelined sweeps: the advance reads the other column
    return bsx_mixed_put(g, BSX_FAM_MNIST, index, call, &a, sizeof(a), &o, sizeof(o),
                              (uint64_t)(call->n_lanes + BSX_BLOCK - 1) / BSX_BLOCK,
                              bsx_flat_blocks((uint64_t)call->n_lanes * o.cells, PAIR_MNIST_K), 0);
  }
  rc = bsx_group_check_set(g, BSX_FAM_MNIST, index, call, sizeof(mnist_args), sizeof(mnist_observe_args), 0);
  if (rc != 0) return rc;
  rc = mnist_make(cfg, call, action, state, out, info, &a, &o);
  if (rc != 0) return rc;
  memcpy(&g->args[(size_t)index * sizeof(a)], &a, sizeof(a));
  memcpy(&g->args2[(size_t)index * sizeof(o)], &o, sizeof(o));
  const uint64_t b1 = (uint64_t)(call->n_lanes + BSX_BLOCK - 1) / BSX_BLOCK;
  const uint64_t b2 = bsx_flat_blocks((uint64_t)call->n_lanes * o.cells, PAIR_MNIST_K);
  if (b1 > 0x3FFFFFFFull || b2 > 0x3FFFFFFFull) return BSX_EINVAL;
  g->blocks[index] = (int32_t)b1; g->blocks2[index] = (int32_t)b2;
  g->is_set[index] = 1;
  g->launch = mnist_group_launch;
  g->n_phases = 2;
  return 0;
}
