// misc.hip — ABI version / error strings, the pure-store calibration kernel and the draw-stream
// dump used by the tests to pin the device Philox / normal transform against the oracle.
#include "bsx_host.h"

extern "C" int bsx_abi_version(void) { return BSX_ABI_VERSION; }

extern "C" const char* bsx_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case BSX_EINVAL: return "invalid scalar argument";
    case BSX_ENULL: return "required pointer is NULL";
    case BSX_EALIGN: return "observation buffer is not 16-byte aligned";
    case BSX_ERANGE: return "parameter outside the supported range of this family";
    case BSX_EMODE: return "not available in MT19937-exact mode (RewardNoise / stochastic deep_sea need randn)";
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

__global__ void counter_add_kernel(uint64_t* counter, uint64_t delta) { *counter += delta; }

extern "C" int bsx_counter_add(uint64_t* counter, uint64_t delta, void* hip_stream) {
  if (counter == nullptr) return BSX_ENULL;
  counter_add_kernel<<<dim3(1), dim3(1), 0, (hipStream_t)hip_stream>>>(counter, delta);
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
