/* Test shim: exposes the MT19937-exact samplers of include/bsx_stream.h (plain C99 there) to
 * Python so tests can compare them draw for draw with numpy's own np.random.RandomState. */
#include "bsx_stream.h"

/* ops: 0 = U(), 1 = bern, 2 = randint(arg), 3 = next_u32, 4 = randn (legacy_gauss).
 * state: 624 words contiguous; pos / has_gauss / gauss in/out. */
void mt_run_gauss(uint32_t* state, int32_t* pos, int32_t* has_gauss, double* gauss, int n_ops, const int32_t* ops,
                  const uint32_t* args, double* out) {
  bsx_draws d;
  bsx_draws_init(&d, 0, 0, 0, 0);
  d.mt = state; d.mt_stride = 1; d.mt_pos = *pos; d.mt_has_gauss = *has_gauss; d.mt_gauss = *gauss;
  for (int i = 0; i < n_ops; i++) {
    switch (ops[i]) {
      case 0: out[i] = bsx_uniform(&d); break;
      case 1: out[i] = (double)bsx_bern(&d); break;
      case 2: out[i] = (double)bsx_randint(&d, args[i]); break;
      case 4: out[i] = bsx_normal(&d); break;
      default: out[i] = (double)bsx_mt_next(&d); break;
    }
  }
  *pos = d.mt_pos; *has_gauss = d.mt_has_gauss; *gauss = d.mt_gauss;
}

void mt_run(uint32_t* state, int32_t* pos, int n_ops, const int32_t* ops, const uint32_t* args,
            double* out) {
  bsx_draws d;
  bsx_draws_init(&d, 0, 0, 0, 0);
  d.mt = state; d.mt_stride = 1; d.mt_pos = *pos;
  for (int i = 0; i < n_ops; i++) {
    switch (ops[i]) {
      case 0: out[i] = bsx_uniform(&d); break;
      case 1: out[i] = (double)bsx_bern(&d); break;
      case 2: out[i] = (double)bsx_randint(&d, args[i]); break;
      default: out[i] = (double)bsx_mt_next(&d); break;
    }
  }
  *pos = d.mt_pos;
}
