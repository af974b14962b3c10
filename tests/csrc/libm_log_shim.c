/* Test shim: bsx_libm_log (include/bsx_libm_log.h) evaluated on the host by gcc. */
#include <math.h>
#include <stdint.h>
#include "bsx_libm_log.h"
void shim_log(const double* x, int64_t n, double* y) { for (int64_t i = 0; i < n; ++i) y[i] = bsx_libm_log(x[i]); }
/* the host libm itself (scalar calls, like numpy legacy_gauss makes them) */
void shim_host_log(const double* x, int64_t n, double* y) { for (int64_t i = 0; i < n; ++i) y[i] = log(x[i]); }
