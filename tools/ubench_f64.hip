// ubench_f64.hip -- issue cost / dependent latency of the VALU instructions the sweep kernel is
// made of, measured on gfx950 with s_memtime around unrolled inline-asm streams (one wave per
// SIMD).  Numbers feed DESIGN.md; not part of the product.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_f64.hip -o gpurun_out/ubench_f64 && gpurun_out/ubench_f64
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// INDEP: four independent register sets; DEP: one chain
#define BENCH(NAME, INDEP_ASM, DEP_ASM)                                                         \
  __global__ void NAME(unsigned long long* out, double seed, int iseed) {                      \
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = 1.0000001, c = 0.25;     \
    int i0 = iseed, i1 = iseed + 1, i2 = iseed + 2, i3 = iseed + 3;                            \
    float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                               \
    unsigned long long t0, t1, t2;                                                              \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                   \
    for (int it = 0; it < 16; ++it) {                                                           \
      asm volatile(REP16(INDEP_ASM)                                                             \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) \
                   : "v"(b), "v"(c));                                                           \
    }                                                                                           \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                   \
    for (int it = 0; it < 16; ++it) {                                                           \
      asm volatile(REP64(DEP_ASM)                                                               \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) \
                   : "v"(b), "v"(c));                                                           \
    }                                                                                           \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                                                  \
      out[0] = t1 - t0;                                                                         \
      out[1] = t2 - t1;                                                                         \
    }                                                                                           \
    if (a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3 + f0 + f1 + f2 + f3 == 12345.678) out[2] = 1;     \
  }

// operands: %0-%3 doubles a0..a3, %4-%7 ints, %8-%11 floats, %12 b, %13 c
BENCH(k_fma_f64, "v_fma_f64 %0, %0, %12, %13\n v_fma_f64 %1, %1, %12, %13\n v_fma_f64 %2, %2, %12, %13\n v_fma_f64 %3, %3, %12, %13\n",
      "v_fma_f64 %0, %0, %12, %13\n")
BENCH(k_add_f64, "v_add_f64 %0, %0, %12\n v_add_f64 %1, %1, %12\n v_add_f64 %2, %2, %12\n v_add_f64 %3, %3, %12\n", "v_add_f64 %0, %0, %12\n")
BENCH(k_mul_f64, "v_mul_f64 %0, %0, %12\n v_mul_f64 %1, %1, %12\n v_mul_f64 %2, %2, %12\n v_mul_f64 %3, %3, %12\n", "v_mul_f64 %0, %0, %12\n")
BENCH(k_ldexp_f64, "v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %5\n v_ldexp_f64 %2, %2, %6\n v_ldexp_f64 %3, %3, %7\n", "v_ldexp_f64 %0, %0, %4\n")
BENCH(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0\n v_frexp_mant_f64 %1, %1\n v_frexp_mant_f64 %2, %2\n v_frexp_mant_f64 %3, %3\n", "v_frexp_mant_f64 %0, %0\n")
BENCH(k_frexp_exp_f64, "v_frexp_exp_i32_f64 %4, %0\n v_frexp_exp_i32_f64 %5, %1\n v_frexp_exp_i32_f64 %6, %2\n v_frexp_exp_i32_f64 %7, %3\n",
      "v_frexp_exp_i32_f64 %4, %0\n")
BENCH(k_rndne_f64, "v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n", "v_rndne_f64 %0, %0\n")
BENCH(k_cvt_i32_f64, "v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_cvt_i32_f64 %6, %2\n v_cvt_i32_f64 %7, %3\n", "v_cvt_i32_f64 %4, %0\n")
BENCH(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n", "v_cvt_f64_f32 %0, %8\n")
BENCH(k_cvt_f32_f64, "v_cvt_f32_f64 %8, %0\n v_cvt_f32_f64 %9, %1\n v_cvt_f32_f64 %10, %2\n v_cvt_f32_f64 %11, %3\n", "v_cvt_f32_f64 %8, %0\n")
BENCH(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %12\n v_cmp_lt_f64 vcc, %1, %12\n v_cmp_lt_f64 vcc, %2, %12\n v_cmp_lt_f64 vcc, %3, %12\n", "v_cmp_lt_f64 vcc, %0, %12\n")
BENCH(k_max_f64, "v_max_f64 %0, %0, %12\n v_max_f64 %1, %1, %12\n v_max_f64 %2, %2, %12\n v_max_f64 %3, %3, %12\n", "v_max_f64 %0, %0, %12\n")
BENCH(k_fma_f32, "v_fma_f32 %8, %8, %8, %9\n v_fma_f32 %9, %9, %9, %10\n v_fma_f32 %10, %10, %10, %11\n v_fma_f32 %11, %11, %11, %8\n", "v_fma_f32 %8, %8, %8, %8\n")
BENCH(k_add_u32, "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %4\n", "v_add_u32 %4, %4, %4\n")
BENCH(k_cndmask, "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %4, vcc\n",
      "v_cndmask_b32 %4, %4, %5, vcc\n")
BENCH(k_mov_dpp, "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 " v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n",
      "s_nop 1\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
BENCH(k_exp_f32, "v_exp_f32 %8, %8\n v_exp_f32 %9, %9\n v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n", "v_exp_f32 %8, %8\n")
BENCH(k_lshl_add, "v_lshl_add_u32 %4, %4, 3, %5\n v_lshl_add_u32 %5, %5, 3, %6\n v_lshl_add_u32 %6, %6, 3, %7\n v_lshl_add_u32 %7, %7, 3, %4\n",
      "v_lshl_add_u32 %4, %4, 3, %4\n")

#define RUN(NAME, WAVES_PER_SIMD)                                                              \
  do {                                                                                         \
    hipMemset(d, 0, 64);                                                                       \
    hipLaunchKernelGGL(NAME, dim3(256 * (WAVES_PER_SIMD)), dim3(256), 0, 0, d, 1.5, 3);          \
    hipDeviceSynchronize();                                                                    \
    unsigned long long h[3];                                                                   \
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);                                                \
    printf("%-18s waves/SIMD=%d  indep %.2f cyc/instr   dep %.2f cyc/instr\n", #NAME, WAVES_PER_SIMD, \
           (double)h[0] / (16.0 * 64.0), (double)h[1] / (16.0 * 64.0));                          \
  } while (0)

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  for (int w = 1; w <= 2; ++w) {
    RUN(k_fma_f64, w); RUN(k_add_f64, w); RUN(k_mul_f64, w); RUN(k_ldexp_f64, w); RUN(k_frexp_mant_f64, w);
    RUN(k_frexp_exp_f64, w); RUN(k_rndne_f64, w); RUN(k_cvt_i32_f64, w); RUN(k_cvt_f64_f32, w); RUN(k_cvt_f32_f64, w);
    RUN(k_cmp_f64, w); RUN(k_max_f64, w); RUN(k_fma_f32, w); RUN(k_add_u32, w); RUN(k_cndmask, w); RUN(k_mov_dpp, w);
    RUN(k_exp_f32, w); RUN(k_lshl_add, w);
  }
  return 0;
}
