// ubench4.hip -- issue cost of the double-precision helper instructions the sweep's producer uses
// (conversions, rounding, compares), lone wave and two waves per SIMD (gfx950).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench4.hip -o tools/ubench4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define BENCH(NAME, ASM)                                                                        \
  __global__ void NAME(unsigned long long* out, int iseed) {                                   \
    double a = iseed * 1.25 + threadIdx.x, b = 1.0000001;                                       \
    float f = iseed * 0.5f + threadIdx.x;                                                       \
    int i0 = iseed + threadIdx.x;                                                               \
    unsigned long long t0, t1;                                                                  \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                   \
    for (int it = 0; it < 16; ++it) {                                                           \
      asm volatile(REP64(ASM) : "+v"(i0), "+v"(a), "+v"(f) : "v"(b) : "memory", "vcc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"); \
    }                                                                                           \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                   \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;            \
    if (i0 == 123456789 || a == 1.5 || f == 7.f) out[15] = 1;                                    \
  }
BENCH(k_fma64, "v_fma_f64 v[100:101], %1, %3, %3\n")
BENCH(k_fma64_dep, "v_fma_f64 %1, %1, %3, %3\n")
BENCH(k_mul64, "v_mul_f64 v[100:101], %1, %3\n")
BENCH(k_add64, "v_add_f64 v[100:101], %1, %3\n")
BENCH(k_max64, "v_max_f64 v[100:101], %1, %3\n")
BENCH(k_cvt_f64_f32, "v_cvt_f64_f32 v[100:101], %2\n")
BENCH(k_cvt_i32_f64, "v_cvt_i32_f64 v100, %1\n")
BENCH(k_rndne64, "v_rndne_f64 v[100:101], %1\n")
BENCH(k_cmp64, "v_cmp_gt_f64 vcc, %1, %3\n")
BENCH(k_cndmask, "v_cndmask_b32 v100, %0, %0, vcc\n")
BENCH(k_mul32, "v_mul_f32 v100, %2, %2\n")
BENCH(k_and32, "v_and_b32 v100, %0, %0\n")
BENCH(k_bfe, "v_bfe_u32 v100, %0, 28, 1\n")
BENCH(k_ldexp64, "v_ldexp_f64 v[100:101], %1, %0\n")
BENCH(k_frexpm64, "v_frexp_mant_f64 v[100:101], %1\n")
BENCH(k_frexpe64, "v_frexp_exp_i32_f64 v100, %1\n")
BENCH(k_dpp, "v_mov_b32_dpp v100, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
BENCH(k_cvt_f64_i32, "v_cvt_f64_i32 v[100:101], %0\n")
#define RUN(NAME, THREADS)                                                                     \
  do {                                                                                         \
    hipMemset(d, 0, 128);                                                                      \
    hipLaunchKernelGGL(NAME, dim3(256), dim3(THREADS), 0, 0, d, 3);                             \
    hipDeviceSynchronize();                                                                    \
    unsigned long long h[10];                                                                  \
    hipMemcpy(h, d, 80, hipMemcpyDeviceToHost);                                                \
    printf("%-16s threads %4d: %.2f cyc per instr (wave 0)\n", #NAME, THREADS, (double)h[0] / (16.0 * 64.0)); \
  } while (0)
#define RUN2(NAME) RUN(NAME, 256); RUN(NAME, 512)
#define SCAN(NAME)                                                                             \
  for (int grid : {1, 256, 1024})                                                              \
    for (int th : {64, 128, 256, 512, 1024}) {                                                 \
      hipMemset(d, 0, 128);                                                                    \
      hipLaunchKernelGGL(NAME, dim3(grid), dim3(th), 0, 0, d, 3);                               \
      hipDeviceSynchronize();                                                                  \
      unsigned long long h[16];                                                                \
      hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);                                             \
      printf("%-12s grid %4d threads %4d: wave0 %.2f  last wave %.2f cyc per instr\n", #NAME, grid, th, \
             (double)h[0] / 1024.0, (double)h[th / 64 - 1] / 1024.0);                           \
    }
int main() {
  unsigned long long* d;
  hipMalloc(&d, 128);
  SCAN(k_fma64); SCAN(k_fma64_dep); SCAN(k_mul32); SCAN(k_cndmask);
  RUN2(k_fma64); RUN2(k_fma64_dep); RUN2(k_mul64); RUN2(k_add64); RUN2(k_max64); RUN2(k_cvt_f64_f32);
  RUN2(k_cvt_i32_f64); RUN2(k_rndne64); RUN2(k_cmp64); RUN2(k_cndmask); RUN2(k_mul32); RUN2(k_and32); RUN2(k_bfe);
  RUN2(k_ldexp64); RUN2(k_frexpm64); RUN2(k_frexpe64); RUN2(k_dpp); RUN2(k_cvt_f64_i32);
  return 0;
}
