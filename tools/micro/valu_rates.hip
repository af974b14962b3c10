// Issue cost of single VALU instructions on gfx950, measured: cycles one SIMD spends per wave64 instruction
// (4 = full rate).  Each thread runs ITER x 16 independent instructions of one kind; 8 waves per SIMD, every SIMD
// busy.  Build: hipcc --offload-arch=gfx950 -O2 tools/micro/valu_rates.hip -o tools/ab/valu_rates   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096

#define KERNEL(NAME, DECL, BODY)                                                                      \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {                         \
    DECL;                                                                                             \
    for (int it = 0; it < ITER; ++it) {                                                               \
      BODY BODY BODY BODY                                                                             \
    }                                                                                                 \
    if (seed == 0x12345u) out[threadIdx.x] = sink(r0, r1, r2, r3);                                    \
  }

template <class T> __device__ uint32_t sink(T a, T b, T c, T d) { return (uint32_t)(a + b + c + d); }

#define U32 uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3u, r2 = r0 ^ 5u, r3 = r0 + 7u; uint32_t k = seed | 1u
#define F32 float r0 = (float)(threadIdx.x + seed), r1 = r0 * 3.f, r2 = r0 + 5.f, r3 = r0 + 7.f; float k = (float)seed + 0.5f
#define F64 double r0 = (double)(threadIdx.x + seed), r1 = r0 * 3., r2 = r0 + 5., r3 = r0 + 7.; double k = (double)seed + 0.5
#define U64 uint64_t r0 = threadIdx.x + seed, r1 = r0 * 3u, r2 = r0 ^ 5u, r3 = r0 + 7u; uint32_t k = seed | 1u

#define ASM4(OP, C)                                                                                   \
  asm volatile(OP " %0, %0, %4\n\t" OP " %1, %1, %4\n\t" OP " %2, %2, %4\n\t" OP " %3, %3, %4"       \
               : "+" C(r0), "+" C(r1), "+" C(r2), "+" C(r3) : C(k));
#define V "v"

KERNEL(k_add_u32, U32, ASM4("v_add_u32", V))
KERNEL(k_xor_b32, U32, ASM4("v_xor_b32", V))
KERNEL(k_mul_lo_u32, U32, ASM4("v_mul_lo_u32", V))
KERNEL(k_mul_hi_u32, U32, ASM4("v_mul_hi_u32", V))
KERNEL(k_mul_u32_u24, U32, ASM4("v_mul_u32_u24", V))
KERNEL(k_mul_f32, F32, ASM4("v_mul_f32", V))
KERNEL(k_add_f64, F64, ASM4("v_add_f64", V))
KERNEL(k_mul_f64, F64, ASM4("v_mul_f64", V))
KERNEL(k_fma_f32, F32, asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4"
                                    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(k));)
KERNEL(k_fma_f64, F64, asm volatile("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4"
                                    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(k));)
KERNEL(k_rcp_f32, F32, asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3"
                                    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)
KERNEL(k_cvt_f64_f32, F64, { float f0 = (float)it; asm volatile("v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %4\n\tv_cvt_f64_f32 %2, %4\n\tv_cvt_f64_f32 %3, %4"
                                    : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(f0)); })
KERNEL(k_mad_u64_u32, U64, asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n\tv_mad_u64_u32 %1, vcc, %4, %4, %1\n\tv_mad_u64_u32 %2, vcc, %4, %4, %2\n\tv_mad_u64_u32 %3, vcc, %4, %4, %3"
                                    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(k) : "vcc");)
KERNEL(k_pk_fma_f32, F64, asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\tv_pk_fma_f32 %3, %3, %3, %3"
                                    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)
// one wave's dependent chain (latency, not throughput): a single wave per SIMD
__global__ void __launch_bounds__(64) k_dep_mul_lo(uint32_t* out, uint32_t seed) {
  uint32_t r0 = threadIdx.x + seed, k = seed | 1u;
  for (int it = 0; it < ITER; ++it)
    asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\t"
                 "v_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\t"
                 "v_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\t"
                 "v_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %1" : "+v"(r0) : "v"(k));
  if (seed == 0x12345u) out[threadIdx.x] = r0;
}
__global__ void __launch_bounds__(64) k_dep_add(uint32_t* out, uint32_t seed) {
  uint32_t r0 = threadIdx.x + seed, k = seed | 1u;
  for (int it = 0; it < ITER; ++it)
    asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
                 "v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
                 "v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
                 "v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(r0) : "v"(k));
  if (seed == 0x12345u) out[threadIdx.x] = r0;
}

template <class K>
static void run(const char* name, K kern, int threads, int blocks, uint32_t* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  kern<<<blocks, threads>>>(out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) kern<<<blocks, threads>>>(out, 1u);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double waves = (double)blocks * threads / 64.0, insts = 16.0 * ITER;
  const double cyc = ms / 5 * 1e-3 * 2.4e9 * 1024.0 / (waves * insts);   // SIMD-cycles per wave-instruction
  printf("%-16s %8.3f ms  %6.2f SIMD-cycles per wave64 instruction (at 2.4 GHz)\n", name, ms / 5, cyc);
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 4096);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, clock %.0f MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1e3);
  const int B = 256 * 8;                    // 8 workgroups of 4 waves per CU: 8 waves per SIMD
#define RUN(K) run(#K, K, 256, B, out)
  RUN(k_add_u32); RUN(k_xor_b32); RUN(k_mul_u32_u24); RUN(k_mul_lo_u32); RUN(k_mul_hi_u32); RUN(k_mad_u64_u32);
  RUN(k_mul_f32); RUN(k_fma_f32); RUN(k_pk_fma_f32); RUN(k_rcp_f32); RUN(k_add_f64); RUN(k_mul_f64); RUN(k_fma_f64); RUN(k_cvt_f64_f32);
  run("dep k_mul_lo_u32", k_dep_mul_lo, 64, 1024, out);   // one wave per SIMD, a dependent chain: latency per instruction
  run("dep k_add_u32", k_dep_add, 64, 1024, out);
  return 0;
}
