// ubench5.hip -- what makes v_cndmask_b32 cost ~19 cycles once four waves share a CU?  Scalar
// operands of VALU instructions (lane masks, SGPR constants) under multi-wave load (gfx950).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench5.hip -o tools/ubench5.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define BENCH(NAME, ASM)                                                                        \
  __global__ void NAME(unsigned long long* out, int iseed, double sc) {                        \
    double a = iseed * 1.25 + threadIdx.x, b = 1.0000001;                                       \
    float f = iseed * 0.5f + threadIdx.x;                                                       \
    int i0 = iseed + threadIdx.x;                                                               \
    unsigned long long t0, t1;                                                                  \
    asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 vcc, exec\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory", "s20", "s21", "vcc"); \
    for (int it = 0; it < 16; ++it) {                                                           \
      asm volatile(REP64(ASM) : "+v"(i0), "+v"(a), "+v"(f) : "v"(b), "s"(sc), "s"(iseed) : "memory", "vcc", "s20", "s21", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"); \
    }                                                                                           \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                   \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;            \
    if (i0 == 123456789 || a == 1.5 || f == 7.f) out[15] = 1;                                   \
  }
BENCH(k_cnd_vcc, "v_cndmask_b32 v100, %0, %0, vcc\n")
BENCH(k_cnd_sgpr, "v_cndmask_b32_e64 v100, %0, %0, s[20:21]\n")
BENCH(k_cnd_vcc_2dst, "v_cndmask_b32 v100, %0, %0, vcc\n v_cndmask_b32 v101, %0, %0, vcc\n")
BENCH(k_fma64_sgpr, "v_fma_f64 v[100:101], %1, %3, %4\n")
BENCH(k_fma64_vgpr, "v_fma_f64 v[100:101], %1, %3, %3\n")
BENCH(k_mul32_sgpr, "v_mul_f32 v100, %5, %2\n")
BENCH(k_add32_sgpr, "v_add_u32 v100, %5, %0\n")
BENCH(k_add32_vgpr, "v_add_u32 v100, %0, %0\n")
BENCH(k_cmp_then_cnd, "v_cmp_gt_i32 vcc, %0, v100\n v_cndmask_b32 v101, %0, %0, vcc\n")
BENCH(k_cmp_vcc, "v_cmp_gt_i32 vcc, %0, v100\n")
BENCH(k_cmp_sgpr, "v_cmp_gt_i32_e64 s[20:21], %0, v100\n")
BENCH(k_bfi, "v_bfi_b32 v100, %0, %0, %0\n")
BENCH(k_addc, "v_addc_co_u32 v100, vcc, %0, %0, vcc\n")
BENCH(k_mov_lit, "v_mov_b32 v100, 0x12345678\n")
BENCH(k_fma64_lit, "v_fma_f64 v[100:101], %1, %3, 1.0\n")
BENCH(k_readlane, "v_readfirstlane_b32 s22, %0\n")
BENCH(k_cnd_vcc_nop, "v_cndmask_b32 v100, %0, %0, vcc\n v_add_u32 v101, %0, %0\n v_add_u32 v102, %0, %0\n v_add_u32 v103, %0, %0\n")
BENCH(k_cmp_2cnd, "v_cmp_gt_i32 vcc, %0, v100\n v_cndmask_b32 v101, %0, %0, vcc\n v_cndmask_b32 v102, %0, %0, vcc\n")
BENCH(k_cmp_3cnd, "v_cmp_gt_i32 vcc, %0, v100\n v_cndmask_b32 v101, %0, %0, vcc\n v_cndmask_b32 v102, %0, %0, vcc\n v_cndmask_b32 v103, %0, %0, vcc\n")
BENCH(k_cmp_2cnd_s, "v_cmp_gt_i32_e64 s[20:21], %0, v100\n v_cndmask_b32_e64 v101, %0, %0, s[20:21]\n v_cndmask_b32_e64 v102, %0, %0, s[20:21]\n")
BENCH(k_cmp_3cnd_s, "v_cmp_gt_i32_e64 s[20:21], %0, v100\n v_cndmask_b32_e64 v101, %0, %0, s[20:21]\n v_cndmask_b32_e64 v102, %0, %0, s[20:21]\n v_cndmask_b32_e64 v103, %0, %0, s[20:21]\n")
BENCH(k_cmp_2cnd_e64vcc, "v_cmp_gt_i32 vcc, %0, v100\n v_cndmask_b32_e64 v101, %0, %0, vcc\n v_cndmask_b32_e64 v102, %0, %0, vcc\n")
BENCH(k_cmpf64_2cnd, "v_cmp_gt_f64 vcc, %1, %3\n v_cndmask_b32 v101, 0, %0, vcc\n v_cndmask_b32 v102, 0, %0, vcc\n")
#define SCAN(NAME)                                                                             \
  for (int grid : {1, 256})                                                                    \
    for (int th : {64, 256, 512}) {                                                            \
      hipMemset(d, 0, 128);                                                                    \
      hipLaunchKernelGGL(NAME, dim3(grid), dim3(th), 0, 0, d, 3, 1.5);                          \
      hipDeviceSynchronize();                                                                  \
      unsigned long long h[16];                                                                \
      hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);                                             \
      printf("%-16s grid %4d threads %4d: wave0 %.2f  last wave %.2f cyc per group\n", #NAME, grid, th, \
             (double)h[0] / 1024.0, (double)h[th / 64 - 1] / 1024.0);                           \
    }
int main() {
  unsigned long long* d;
  hipMalloc(&d, 128);
  SCAN(k_cmp_2cnd); SCAN(k_cmp_3cnd); SCAN(k_cmp_2cnd_s); SCAN(k_cmp_3cnd_s); SCAN(k_cmp_2cnd_e64vcc); SCAN(k_cmpf64_2cnd);
  if (0) {SCAN(k_cnd_vcc); SCAN(k_cnd_sgpr); SCAN(k_cnd_vcc_2dst); SCAN(k_fma64_sgpr); SCAN(k_fma64_vgpr); SCAN(k_mul32_sgpr);
  SCAN(k_add32_sgpr); SCAN(k_add32_vgpr); SCAN(k_cmp_then_cnd); SCAN(k_cmp_vcc); SCAN(k_cmp_sgpr); SCAN(k_bfi); SCAN(k_addc);
  SCAN(k_mov_lit); SCAN(k_fma64_lit); SCAN(k_readlane); SCAN(k_cnd_vcc_nop); }
  return 0;
}
