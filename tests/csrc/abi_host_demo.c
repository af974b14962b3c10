/* Plain-C99 consumer of the C ABI (include/bsuite_amd.h): no C++, no Python, no torch.
 * Steps a batch of deterministic DeepSea environments (identity action mapping: action 1 = right)
 * with the always-right policy and checks the known answers of bsuite/environments/deep_sea.py:
 *   call 0            FIRST, reward 0, observation one-hot at [0,0]
 *   calls 1..N-1      MID, reward -0.01/N each (deep_sea.py:132), observation one-hot at [t,t]
 *   call N            LAST, reward 1 - 0.01/N (:121-123,132), discount 0, all-zero observation (:105-107)
 *   call N+1          FIRST again (base.py:61-62), action ignored
 * Build: gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/csrc/abi_host_demo.c
 *        -Lbsuite_amd/_lib -lbsuite_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,...   (tests/test_c_host.py) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bsuite_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s:%d hip error %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_BSX(x) do { int rc_ = (x); if (rc_ != 0) { \
  fprintf(stderr, "%s:%d bsx error %d: %s\n", __FILE__, __LINE__, rc_, bsx_strerror(rc_)); return 3; } } while (0)
#define EXPECT(c) do { if (!(c)) { fprintf(stderr, "%s:%d expectation failed: %s (call %d, lane %d)\n", \
  __FILE__, __LINE__, #c, t, i); return 4; } } while (0)

int main(void) {
  enum { N = 6, B = 1000 };
  if (bsx_abi_version() != BSX_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

  bsx_deep_sea_t cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.size = N; cfg.deterministic = 1;
  cfg.move_cost = 0.01 / N; cfg.inv_size = 1.0 / N;
  for (int c = 0; c < N * N; ++c) cfg.mapping_bits[c >> 5] |= 1u << (c & 31);   /* np.ones mapping (:85) */

  int32_t *d_action, *d_state; float *d_reward, *d_discount, *d_obs; int8_t* d_type; double* d_info;
  uint64_t* d_counters;
  CHECK_HIP(hipMalloc((void**)&d_action, B * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&d_state, B * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void**)&d_reward, B * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_discount, B * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_type, B));
  CHECK_HIP(hipMalloc((void**)&d_obs, (size_t)B * N * N * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_info, 2 * B * sizeof(double)));
  CHECK_HIP(hipMalloc((void**)&d_counters, BSX_COUNTER_SHARDS * BSX_COUNTER_STRIDE * sizeof(uint64_t)));
  CHECK_HIP(hipMemset(d_info, 0, 2 * B * sizeof(double)));
  CHECK_HIP(hipMemset(d_counters, 0, BSX_COUNTER_SHARDS * BSX_COUNTER_STRIDE * sizeof(uint64_t)));

  static int32_t h_i32[B]; static float h_reward[B], h_discount[B], h_obs[B * N * N]; static int8_t h_type[B];
  for (int i = 0; i < B; ++i) h_i32[i] = 1 << 17;            /* every lane starts with its reset flag set (base.py:52) */
  CHECK_HIP(hipMemcpy(d_state, h_i32, sizeof h_i32, hipMemcpyHostToDevice));
  for (int i = 0; i < B; ++i) h_i32[i] = 1;                  /* always "right" */
  CHECK_HIP(hipMemcpy(d_action, h_i32, sizeof h_i32, hipMemcpyHostToDevice));

  bsx_call_t call;
  memset(&call, 0, sizeof call);
  call.n_lanes = B; call.stream.seed = 42; call.counters = d_counters;    /* hip_stream NULL = default stream */
  bsx_timestep_t out = {d_reward, d_discount, d_type, d_obs};

  const float step_cost = (float)(0.0 - cfg.move_cost), goal = (float)((0.0 + 1.0) - cfg.move_cost);
  for (int t = 0; t <= N + 1; ++t) {
    call.stream.step_index = (uint64_t)t;
    CHECK_BSX(bsx_deep_sea_step(&cfg, &call, d_action, d_state, out, d_info));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(h_reward, d_reward, sizeof h_reward, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_discount, d_discount, sizeof h_discount, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_type, d_type, sizeof h_type, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_obs, d_obs, sizeof h_obs, hipMemcpyDeviceToHost));
    for (int i = 0; i < B; ++i) {
      const int phase = t % (N + 1);                         /* 0: FIRST, 1..N-1: MID, N: LAST */
      const float* board = h_obs + (size_t)i * N * N;
      float sum = 0.f;
      for (int c = 0; c < N * N; ++c) sum += board[c];
      if (phase == 0) {
        EXPECT(h_type[i] == BSX_FIRST); EXPECT(h_reward[i] == 0.0f); EXPECT(h_discount[i] == 1.0f);
        EXPECT(sum == 1.0f); EXPECT(board[0] == 1.0f);
      } else if (phase < N) {
        EXPECT(h_type[i] == BSX_MID); EXPECT(h_reward[i] == step_cost); EXPECT(h_discount[i] == 1.0f);
        EXPECT(sum == 1.0f); EXPECT(board[phase * N + phase] == 1.0f);
      } else {
        EXPECT(h_type[i] == BSX_LAST); EXPECT(h_reward[i] == goal); EXPECT(h_discount[i] == 0.0f);
        EXPECT(sum == 0.0f);
      }
    }
  }
  static double h_info[2 * B]; static uint64_t h_counters[BSX_COUNTER_SHARDS * BSX_COUNTER_STRIDE];
  CHECK_HIP(hipMemcpy(h_info, d_info, sizeof h_info, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_counters, d_counters, sizeof h_counters, hipMemcpyDeviceToHost));
  uint64_t last = 0, first = 0;
  for (int s = 0; s < BSX_COUNTER_SHARDS; ++s) { last += h_counters[s * BSX_COUNTER_STRIDE]; first += h_counters[s * BSX_COUNTER_STRIDE + 1]; }
  { int t = N + 1, i = 0;
    EXPECT(last == (uint64_t)B); EXPECT(first == 2u * (uint64_t)B);
    for (i = 0; i < B; ++i) { EXPECT(h_info[i] == 0.0); EXPECT(h_info[B + i] == 1.0); }   /* bad episodes 0, denoised return 1 */
  }
  /* argument errors come back as codes, never as faults */
  { int t = -1, i = -1;
    EXPECT(bsx_deep_sea_step(NULL, &call, d_action, d_state, out, d_info) == BSX_ENULL);
    call.n_lanes = -3;
    EXPECT(bsx_deep_sea_step(&cfg, &call, d_action, d_state, out, d_info) == BSX_EINVAL);
    call.n_lanes = B;
  }
  /* catch (bsuite/environments/catch.py:68-117): the paddle never moves (action 1 = stay, :27), so a lane's total_regret
   * is 2 x the episodes whose ball did not fall on the centre column — counted here from the rewards — and
   * bsx_bsuite_info() must report exactly that although the library keeps the misses in spare bits of the state word
   * and folds them into the f64 column only once per 127 (the header's accounting note). */
  {
    enum { ROWS = 10, COLS = 5, CALLS = 10 * 200 };           /* 200 episodes: every lane misses more than 127 times */
    bsx_catch_t ccfg = {ROWS, COLS};
    float* d_cobs; double *d_cinfo, *d_cinfo_out;
    static double h_cinfo[B], h_cinfo_out[B], want[B];
    static int32_t h_state[B];
    CHECK_HIP(hipMalloc((void**)&d_cobs, (size_t)B * ROWS * COLS * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&d_cinfo, B * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&d_cinfo_out, B * sizeof(double)));
    CHECK_HIP(hipMemset(d_cinfo, 0, B * sizeof(double)));
    for (int i = 0; i < B; ++i) { h_i32[i] = 1 << 24; want[i] = 0.0; }     /* reset_next (base.py:52) */
    CHECK_HIP(hipMemcpy(d_state, h_i32, sizeof h_i32, hipMemcpyHostToDevice));
    for (int i = 0; i < B; ++i) h_i32[i] = 1;                               /* "stay" */
    CHECK_HIP(hipMemcpy(d_action, h_i32, sizeof h_i32, hipMemcpyHostToDevice));
    call.n_lanes = B; call.stream.seed = 7;
    bsx_timestep_t cout = {d_reward, d_discount, d_type, d_cobs};
    int t, i = -1;
    for (t = 0; t < CALLS; ++t) {
      call.stream.step_index = (uint64_t)t;
      CHECK_BSX(bsx_catch_step(&ccfg, &call, d_action, d_state, cout, d_cinfo));
      if (t % 10 == 9) {                                                     /* the call that ends every lane's episode */
        CHECK_HIP(hipMemcpy(h_reward, d_reward, sizeof h_reward, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(h_type, d_type, sizeof h_type, hipMemcpyDeviceToHost));
        for (i = 0; i < B; ++i) { EXPECT(h_type[i] == BSX_LAST); EXPECT(h_reward[i] == 1.0f || h_reward[i] == -1.0f); want[i] += 1.0 - (double)h_reward[i]; }
      }
    }
    CHECK_BSX(bsx_bsuite_info(BSX_FAM_CATCH, 0, B, d_state, d_cinfo, 1, 1, d_cinfo_out, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(h_cinfo, d_cinfo, sizeof h_cinfo, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_cinfo_out, d_cinfo_out, sizeof h_cinfo_out, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_state, d_state, sizeof h_state, hipMemcpyDeviceToHost));
    int folded_lanes = 0, misses = 0;
    for (i = 0; i < B; ++i) {
      EXPECT(h_cinfo_out[i] == want[i]);                                      /* the reference's total_regret (:116-117) */
      EXPECT(h_cinfo[i] + 2.0 * (double)(((uint32_t)h_state[i]) >> 25) == want[i]);   /* the documented identity */
      folded_lanes += h_cinfo[i] != 0.0;
      misses += (int)(want[i] / 2.0);
    }
    EXPECT(folded_lanes > B / 2);                                             /* lanes crossed 127 misses: both parts were exercised */
    EXPECT(misses > 150 * B && misses < 170 * B);                             /* ~4/5 of 200 episodes miss with the paddle at rest */
    /* a plain copy where nothing is pending (deep_sea), and the argument checks */
    CHECK_BSX(bsx_bsuite_info(BSX_FAM_DEEP_SEA, 0, B, NULL, d_info, 1, 1, d_cinfo_out, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(h_cinfo_out, d_cinfo_out, sizeof h_cinfo_out, hipMemcpyDeviceToHost));
    for (i = 0; i < B; ++i) EXPECT(h_cinfo_out[i] == h_info[i]);
    EXPECT(bsx_bsuite_info(BSX_FAM_CATCH, 0, B, NULL, d_cinfo, 1, 1, d_cinfo_out, NULL) == BSX_ENULL);
    EXPECT(bsx_bsuite_info(99, 0, B, d_state, d_cinfo, 1, 1, d_cinfo_out, NULL) == BSX_EINVAL);
    EXPECT(bsx_row_scratch_bytes(BSX_FAM_UMBRELLA_CHAIN, 23, 64) == 4 * (2 * 23 + 64) && bsx_row_scratch_bytes(BSX_FAM_CATCH, 50, 64) == 0);
  }
  printf("abi_host_demo: ok (%d lanes x %d calls of deep_sea N=%d, catch total_regret via bsx_bsuite_info, through the C ABI)\n", B, N + 2, N);
  return 0;
}
