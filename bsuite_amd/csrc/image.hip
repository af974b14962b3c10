// ImageObservation adapter: bsuite/utils/wrappers.py:150-247 `to_image` for a whole
// batch as one store stream.  Output per lane is an [H, W, tail] f32 image (84x84x4 = 110 KiB for
// the Atari-like format), so the kernel is bound by HBM stores exactly like the deep_sea
// observation stream; the source observation of a lane (<= 16 KiB) is staged in LDS.
//
// Work split: a workgroup owns one run of IMG_K * 1024 consecutive floats of ONE lane's image, so the
// per-axis interpolation tables (source indices + f64 weights, identical for every lane) and the
// lane's observation are built once per workgroup in LDS; each thread then issues IMG_K
// lane-interleaved 16-byte stores.
//
// Arithmetic of the bilinear rule restates scipy.ndimage.zoom(order=1, mode='mirror',
// grid_mode=True) — what skimage.transform.resize(order=1, mode='reflect') calls, after its
// anti-aliasing Gaussian along the axes that shrink (img_gauss_pass) — operation for operation in
// f64 without FMA:
//   zoom = in / out;  cc = ((k + 0.5) * zoom) - 0.5;  cc = mirror(cc);  f = floor(cc);
//   weights (1 - (cc - f), cc - f) on source indices mirror(f), mirror(f + 1);
//   t = 0 + (v00*wy0)*wx0 + (v01*wy0)*wx1 + (v10*wy1)*wx0 + (v11*wy1)*wx1;  out = (float)t
// (pinned bit for bit against scipy in tests/test_image_oracle.py and tests/test_gpu_image.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bsuite_amd.h"
#include "bsx_device.h"
#include "bsx_host.h"

// The anti-aliasing half kernels travel BY VALUE in the kernel arguments (2 x 65 doubles = 1040 B) and are copied to
// LDS by the workgroups that filter: no device buffer owned by the library, so no allocation or synchronous upload
// inside the entry point (capturable from the first call) and nothing tied to the device that happened to be current
// when a configuration was first seen (a process-global cache of device pointers faulted on a second GPU).
struct image_gauss { double w[2][BSX_IMAGE_MAX_RADIUS + 1]; };   // [0] rows (y), [1] columns (x)

struct image_args {
  const float* obs; float* image; int64_t n_lanes;
  int32_t mode, in_rows, in_cols, out_rows, out_cols, tail;
  int32_t radius_y, radius_x;                     // anti-aliasing Gaussian (down-scaling), 0 = none
  uint32_t numel;                                 // out_rows*out_cols*tail (< 2^20)
  uint32_t tail_magic, cols_magic;                // bsx_div_magic(tail), bsx_div_magic(out_cols)
  uint32_t blocks_per_lane;
};


// scipy ni_interpolation.c map_coordinate(), NI_EXTEND_MIRROR, for the (-0.5, len-0.5) range the
// grid-mode zoom produces (one reflection suffices; the general fold is kept for safety).
__device__ __forceinline__ double img_mirror_coord(double c, int len) {
  BSX_NO_CONTRACT
  if (len <= 1) return (c < 0.0 || c > (double)(len - 1)) ? 0.0 : c;
  const int s2 = 2 * len - 2;
  if (c < 0.0) {
    c = (double)(s2 * (int)(-c / (double)s2)) + c;
    return c <= (double)(1 - len) ? c + (double)s2 : -c;
  }
  if (c > (double)(len - 1)) {
    c -= (double)(s2 * (int)(c / (double)s2));
    if (c >= (double)len) c = (double)s2 - c;
  }
  return c;
}

__device__ __forceinline__ int img_mirror_index(int i, int len) {
  if (len <= 1) return 0;
  const int s2 = 2 * len - 2;
  if (i < 0) {
    i = s2 * (-i / s2) + i;
    return i <= 1 - len ? i + s2 : -i;
  }
  if (i >= len) {
    i -= s2 * (i / s2);
    if (i >= len) i = s2 - i;
  }
  return i;
}

// One axis-table entry: source index pair (packed 16+16) and the two weights.
__device__ __forceinline__ void img_axis_entry(int k, int n_in, int n_out, int* idx, double* w0, double* w1) {
  BSX_NO_CONTRACT
  const double zoom = (double)n_in / (double)n_out;
  double cc = (double)k;
  cc = cc + 0.5;
  cc = cc * zoom;
  cc = cc - 0.5;
  cc = img_mirror_coord(cc, n_in);
  const double fl = floor(cc);
  const int start = (int)fl;
  const double t = cc - fl;
  *w0 = 1.0 - t;
  *w1 = t;
  *idx = img_mirror_index(start, n_in) | (img_mirror_index(start + 1, n_in) << 16);
}

struct img_tables {
  const float* s_obs; const int* s_yi; const int* s_xi;
  const double* s_yw; const double* s_xw;          // [2*k], [2*k+1]
};

__device__ __forceinline__ float img_pixel(const image_args& a, const img_tables& tb, uint32_t p) {
  BSX_NO_CONTRACT
  const uint32_t y = bsx_div_cells(p, (uint32_t)a.out_cols, a.cols_magic);
  const uint32_t x = p - y * (uint32_t)a.out_cols;
  if (a.mode == BSX_IMAGE_SMALL) {                 // wrappers.py:178-204
    const int size = a.in_rows * a.in_cols;
    const bool right = (int)x >= (a.out_cols >> 1), lower = (int)y >= (a.out_rows >> 1);
    int j;
    if (size == 1) j = 0;
    else if (size == 2) j = right ? 1 : 0;
    else j = right ? (lower ? size - 1 : 2) : (lower ? 1 : 0);
    return tb.s_obs[j];
  }
  const int yi = tb.s_yi[y], xi = tb.s_xi[x];
  const int y0 = (yi & 0xFFFF) * a.in_cols, y1 = (yi >> 16) * a.in_cols;
  const int x0 = xi & 0xFFFF, x1 = xi >> 16;
  const double wy0 = tb.s_yw[2 * y], wy1 = tb.s_yw[2 * y + 1];
  const double wx0 = tb.s_xw[2 * x], wx1 = tb.s_xw[2 * x + 1];
  double t = 0.0;
  t = t + ((double)tb.s_obs[y0 + x0] * wy0) * wx0;
  t = t + ((double)tb.s_obs[y0 + x1] * wy0) * wx1;
  t = t + ((double)tb.s_obs[y1 + x0] * wy1) * wx0;
  t = t + ((double)tb.s_obs[y1 + x1] * wy1) * wx1;
  return (float)t;
}

// skimage's anti-aliasing pre-filter = scipy.ndimage.gaussian_filter(mode='mirror') on the f32
// observation staged in LDS: one pass per filtered axis (rows, then columns), each element
//   t = in[0]*w[0];  for j = radius..1: t += (in[-j] + in[+j]) * w[j]        (f64, NI_Correlate1D's
// symmetric branch), rounded to f32 like scipy's float32 output array.  src -> dst, both [rows x cols].
__device__ __forceinline__ void img_gauss_pass(const float* src, float* dst, int rows, int cols, int axis,
                                               int radius, const double* __restrict__ w) {
  BSX_NO_CONTRACT
  const int n = rows * cols, len = axis == 0 ? rows : cols, stride = axis == 0 ? cols : 1;
  for (int e = threadIdx.x; e < n; e += BSX_BLOCK) {
    const int y = e / cols, x = e - y * cols;
    const int pos = axis == 0 ? y : x;
    const int base = e - pos * stride;               // element 0 of this line
    double t = (double)src[e] * w[0];
    for (int j = radius; j >= 1; --j) {
      const double lo = (double)src[base + img_mirror_index(pos - j, len) * stride];
      const double hi = (double)src[base + img_mirror_index(pos + j, len) * stride];
      t = t + (lo + hi) * w[j];
    }
    dst[e] = (float)t;
  }
}

template <int IMG_K>
__global__ void __launch_bounds__(BSX_BLOCK) bsx_image_kernel(const image_args a, const image_gauss gw) {
  constexpr uint32_t IMG_RUN = IMG_K * BSX_BLOCK * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ double s_gauss[2][BSX_IMAGE_MAX_RADIUS + 1];
  if (a.radius_y > 0 || a.radius_x > 0) {         // uniform
    for (int j = threadIdx.x; j < 2 * (BSX_IMAGE_MAX_RADIUS + 1); j += BSX_BLOCK)
      s_gauss[j / (BSX_IMAGE_MAX_RADIUS + 1)][j % (BSX_IMAGE_MAX_RADIUS + 1)] = gw.w[j / (BSX_IMAGE_MAX_RADIUS + 1)][j % (BSX_IMAGE_MAX_RADIUS + 1)];
  }
  // LDS layout: y weights f64 [2*H] | x weights f64 [2*W] | y idx i32 [H] | x idx i32 [W] | obs f32
  double* s_yw = reinterpret_cast<double*>(s_raw);
  double* s_xw = s_yw + 2 * a.out_rows;
  int* s_yi = reinterpret_cast<int*>(s_xw + 2 * a.out_cols);
  int* s_xi = s_yi + a.out_rows;
  float* s_obs = reinterpret_cast<float*>(s_xi + a.out_cols);
  float* s_tmp = s_obs + a.in_rows * a.in_cols;    // second plane: only allocated when a filter runs

  const uint32_t lane = blockIdx.x / a.blocks_per_lane;
  const uint32_t run = blockIdx.x - lane * a.blocks_per_lane;
  const int in_numel = a.in_rows * a.in_cols;
  const float* __restrict__ src = a.obs + (int64_t)lane * in_numel;
  for (int j = threadIdx.x; j < in_numel; j += BSX_BLOCK) s_obs[j] = src[j];
  if (a.mode == BSX_IMAGE_BILINEAR) {
    for (int k = threadIdx.x; k < a.out_rows + a.out_cols; k += BSX_BLOCK) {
      int idx; double w0, w1;
      if (k < a.out_rows) {
        img_axis_entry(k, a.in_rows, a.out_rows, &idx, &w0, &w1);
        s_yi[k] = idx; s_yw[2 * k] = w0; s_yw[2 * k + 1] = w1;
      } else {
        const int kx = k - a.out_rows;
        img_axis_entry(kx, a.in_cols, a.out_cols, &idx, &w0, &w1);
        s_xi[kx] = idx; s_xw[2 * kx] = w0; s_xw[2 * kx + 1] = w1;
      }
    }
  }
  __syncthreads();
  if (a.radius_y > 0) {                            // uniform branches: the barriers are safe
    img_gauss_pass(s_obs, s_tmp, a.in_rows, a.in_cols, 0, a.radius_y, s_gauss[0]);
    __syncthreads();
    float* t = s_obs; s_obs = s_tmp; s_tmp = t;
  }
  if (a.radius_x > 0) {
    img_gauss_pass(s_obs, s_tmp, a.in_rows, a.in_cols, 1, a.radius_x, s_gauss[1]);
    __syncthreads();
    float* t = s_obs; s_obs = s_tmp; s_tmp = t;
  }
  img_tables tb; tb.s_obs = s_obs; tb.s_yi = s_yi; tb.s_xi = s_xi; tb.s_yw = s_yw; tb.s_xw = s_xw;

  float* __restrict__ dst = a.image + (int64_t)lane * a.numel;
  const uint32_t f_begin = run * IMG_RUN;
  if ((a.numel & 3u) == 0u) {
    // 16-byte stores; each wave owns IMG_K consecutive KiB (same order as the observation stream)
    const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
    const bool same = (a.tail & 3) == 0;           // the 4 elements of a chunk share their pixel
#pragma unroll
    for (int j = 0; j < IMG_K; ++j) {
      const uint32_t f = f_begin + ((wave * IMG_K + j) * 64u + wl) * 4u;
      if (f < a.numel) {
        bsx_f4 v;
        if (same) {
          const float p = img_pixel(a, tb, bsx_div_cells(f, (uint32_t)a.tail, a.tail_magic));
          v.x = p; v.y = p; v.z = p; v.w = p;
        } else {
          uint32_t p0 = bsx_div_cells(f, (uint32_t)a.tail, a.tail_magic);
          uint32_t rem = f - p0 * (uint32_t)a.tail;  // position inside pixel p0's tail
          float val = img_pixel(a, tb, p0);
          float e[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            e[q] = val;
            if (++rem == (uint32_t)a.tail && q < 3) { rem = 0; ++p0; val = img_pixel(a, tb, p0); }
          }
          v.x = e[0]; v.y = e[1]; v.z = e[2]; v.w = e[3];
        }
        *reinterpret_cast<bsx_f4*>(dst + f) = v;
      }
    }
  } else {
    // images whose size is not a multiple of 4 floats are not 16-byte aligned per lane: 4-byte stores
    const uint32_t f_end = f_begin + IMG_RUN < a.numel ? f_begin + IMG_RUN : a.numel;
    for (uint32_t f = f_begin + threadIdx.x; f < f_end; f += BSX_BLOCK)
      dst[f] = img_pixel(a, tb, bsx_div_cells(f, (uint32_t)a.tail, a.tail_magic));
  }
}

extern "C" int bsx_image_observation(const bsx_image_t* cfg, int64_t n_lanes, const float* obs, float* image,
                                     void* hip_stream) {
  if (cfg == nullptr) return BSX_ENULL;
  if (n_lanes < 0) return BSX_EINVAL;
  if (cfg->mode != BSX_IMAGE_SMALL && cfg->mode != BSX_IMAGE_BILINEAR) return BSX_EINVAL;
  if (cfg->in_rows < 1 || cfg->in_cols < 1 || cfg->out_rows < 1 || cfg->out_cols < 1 || cfg->tail < 1)
    return BSX_ERANGE;
  const int64_t in_numel = (int64_t)cfg->in_rows * cfg->in_cols;
  const int64_t numel = (int64_t)cfg->out_rows * cfg->out_cols * cfg->tail;
  if (in_numel > 4096 || cfg->out_rows > 1024 || cfg->out_cols > 1024 || cfg->tail > 4096 || numel >= (1 << 20))
    return BSX_ERANGE;
  if (cfg->mode == BSX_IMAGE_SMALL && in_numel > 4) return BSX_ERANGE;        // wrappers.py:200-201
  if (cfg->radius_y < 0 || cfg->radius_x < 0 || cfg->radius_y > BSX_IMAGE_MAX_RADIUS || cfg->radius_x > BSX_IMAGE_MAX_RADIUS)
    return BSX_ERANGE;
  const bool filtered = cfg->mode == BSX_IMAGE_BILINEAR && (cfg->radius_y > 0 || cfg->radius_x > 0);
  if (n_lanes == 0) return 0;
  if (obs == nullptr || image == nullptr) return BSX_ENULL;
  if (((uintptr_t)image & 15u) != 0 || ((uintptr_t)obs & 3u) != 0) return BSX_EALIGN;
  image_args a;
  a.obs = obs; a.image = image; a.n_lanes = n_lanes;
  a.mode = cfg->mode; a.in_rows = cfg->in_rows; a.in_cols = cfg->in_cols;
  a.out_rows = cfg->out_rows; a.out_cols = cfg->out_cols; a.tail = cfg->tail;
  a.radius_y = filtered ? cfg->radius_y : 0; a.radius_x = filtered ? cfg->radius_x : 0;
  a.numel = (uint32_t)numel;
  a.tail_magic = bsx_div_magic((uint32_t)cfg->tail);
  a.cols_magic = bsx_div_magic((uint32_t)cfg->out_cols);
  // Run length per workgroup: every workgroup re-stages its lane's observation and rebuilds the axis
  // tables before its first store, so large observations want longer runs (measured on 84x84x4,
  // profiles/r01/ab_image_k.log: 10x5 input best at 4 KiB x 4, 30x30 at x16; images whose
  // channel count is not a multiple of 4 evaluate up to four pixels per store and prefer x8).  BSX_IMAGE_K overrides.
  static const int k_knob = bsx_env_int("BSX_IMAGE_K", 0);
  const int k = (k_knob == 2 || k_knob == 4 || k_knob == 8 || k_knob == 16) ? k_knob
                : (in_numel > 256 && (cfg->tail & 3) == 0 ? 16 : (in_numel > 64 ? 8 : 4));
  const int64_t run = (int64_t)k * BSX_BLOCK * 4;
  a.blocks_per_lane = (uint32_t)((numel + run - 1) / run);
  const int64_t blocks = n_lanes * (int64_t)a.blocks_per_lane;
  if (blocks > 0x7FFFFFFF) return BSX_EINVAL;
  const size_t lds = (size_t)(cfg->out_rows + cfg->out_cols) * (16 + 4) + (size_t)in_numel * 4 * (filtered ? 2 : 1);
  const dim3 grid((unsigned)blocks), block(BSX_BLOCK);
  hipStream_t st = (hipStream_t)hip_stream;
  image_gauss gw;
  memset(&gw, 0, sizeof(gw));
  if (filtered) {
    for (int j = 0; j <= cfg->radius_y; ++j) gw.w[0][j] = cfg->gauss_y[j];
    for (int j = 0; j <= cfg->radius_x; ++j) gw.w[1][j] = cfg->gauss_x[j];
  }
  switch (k) {
    case 2: bsx_image_kernel<2><<<grid, block, lds, st>>>(a, gw); break;
    case 8: bsx_image_kernel<8><<<grid, block, lds, st>>>(a, gw); break;
    case 16: bsx_image_kernel<16><<<grid, block, lds, st>>>(a, gw); break;
    default: bsx_image_kernel<4><<<grid, block, lds, st>>>(a, gw); break;
  }
  return bsx_launch_status();
}
