// mnist.hip — batched MNIST contextual bandit for gfx950.  Replaces
// bsuite/environments/mnist.py:61-75 (`_reset`, `_step`) + base.py:54-65 auto-reset.
//
// Episode = reset (show image idx = randint(num_data), obs = int8 pixels / 255 in f32) + one step
// (reward +-1 by action == label, LAST, obs = zeros).  Per call and lane: 3136 B of observation
// stores, on FIRST calls plus a 784 B gather from the L2/Infinity-cache resident image table.
//   advance  mnist_advance_kernel: one lane per thread, packed state = idx | label | flags
//   observe  mnist_observe_kernel: pure 16-byte store stream over [B x 784] f32; a chunk is four
//            table bytes mapped through the 256-entry LUT (LDS) or zero.
// The LUT holds np.float32(int8(b)) / 255 evaluated by numpy on the host, so the int8 parsing quirk
// of the reference (datasets.py:55-56) and its f32 division are reproduced bit for bit.
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

template <int K, int VAR>
__global__ void __launch_bounds__(BSX_BLOCK) mnist_observe_kernel(const mnist_observe_args a) {
  __shared__ float s_lut[MNIST_LUT_FLOATS];
  mnist_observe_body<K, VAR>(a, blockIdx.x, s_lut);
}

template <int K, int VAR>
__global__ void __launch_bounds__(BSX_BLOCK) mnist_observe_group_kernel(const mnist_observe_args* __restrict__ table,
                                                                        const bsx_group_index gi) {
  __shared__ float s_lut[MNIST_LUT_FLOATS];
  const bsx_group_slot w = bsx_group_find(gi, (int)blockIdx.x);
  mnist_observe_body<K, VAR>(table[w.seg], w.block, s_lut);
}

static int mnist_variant() {
  static const int v = bsx_env_int("BSX_MNIST_VARIANT", 3) & 7;
  return v;
}
static int mnist_group_k() {
  // grouped workgroups pay two extra dependent loads (map entry, argument table) before their first
  // store: 32 KiB runs amortise that better than the 16 KiB of the single-segment kernel
  // (profiles/r01/ab_mnist_group_k.log: 100 -> 86 us per sweep step)
  static const int v = bsx_env_int("BSX_MNIST_GROUP_K", 8);
  return (v == 2 || v == 4) ? v : 8;
}

#define MNIST_K 4

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
  // A/B knob (tuning build): compute the reference's table (np.float32(int8) / 255, bsx_mnist_pixel_value) in the kernel
  // instead of looking it up in LDS — exact, and measured slower (see pair_mixed.h): the product library always looks up
  static const int arith_env = bsx_env_int("BSX_MNIST_ARITH", 0);
  o->arith = arith_env != 0; o->_pad = 0;
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
    const dim3 go((unsigned)blocks_o), bo(BSX_BLOCK);
#if defined(BSX_TUNING)
    if (o.arith) { mnist_observe_kernel<MNIST_K, 3 | 8><<<go, bo, 0, st>>>(o); continue; }
#endif
    switch (mnist_variant()) {
      case 1: mnist_observe_kernel<MNIST_K, 1><<<go, bo, 0, st>>>(o); break;
      case 2: mnist_observe_kernel<MNIST_K, 2><<<go, bo, 0, st>>>(o); break;
      case 3: mnist_observe_kernel<MNIST_K, 3><<<go, bo, 0, st>>>(o); break;
      case 7: mnist_observe_kernel<MNIST_K, 7><<<go, bo, 0, st>>>(o); break;
      default: mnist_observe_kernel<MNIST_K, 0><<<go, bo, 0, st>>>(o); break;
    }
  }
  return bsx_launch_status();
}

static int mnist_group_launch(bsx_group* g, int phase, hipStream_t st) {
  if (phase != 1)
    mnist_advance_group_kernel<<<dim3((unsigned)g->total_blocks), dim3(BSX_BLOCK), 0, st>>>(
        (const mnist_args*)g->d_args, g->index1());
  if (phase == 0) return (int)hipGetLastError();
  const dim3 go((unsigned)g->total_blocks2), bo(BSX_BLOCK);
  const mnist_observe_args* tb = (const mnist_observe_args*)g->d_args2;
  const int k = mnist_group_k(), var = mnist_variant();
#define MN_GO(K, V) mnist_observe_group_kernel<K, V><<<go, bo, 0, st>>>(tb, g->index2())
  if (k == 8) { if (var == 3) MN_GO(8, 3); else MN_GO(8, 0); }
  else if (k == 2) { if (var == 3) MN_GO(2, 3); else MN_GO(2, 0); }
  else { if (var == 3) MN_GO(4, 3); else MN_GO(4, 0); }
#undef MN_GO
  return (int)hipGetLastError();
}

extern "C" int bsx_group_set_mnist(bsx_group_t* g, int32_t index, const bsx_mnist_t* cfg, const bsx_call_t* call,
                                   const int32_t* action, int32_t* state, bsx_timestep_t out, double* info) {
  if (g == nullptr) return BSX_ENULL;
  int rc;
  mnist_args a;
  mnist_observe_args o;
  if (bsx_is_mixed_pair_group(g)) {            // one segment of the mixed two-kernel group (8 x 4 KiB runs)
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
  const uint64_t b2 = bsx_flat_blocks((uint64_t)call->n_lanes * o.cells, mnist_group_k());
  if (b1 > 0x3FFFFFFFull || b2 > 0x3FFFFFFFull) return BSX_EINVAL;
  g->blocks[index] = (int32_t)b1; g->blocks2[index] = (int32_t)b2;
  g->is_set[index] = 1;
  g->launch = mnist_group_launch;
  g->n_phases = 2;
  return 0;
}
