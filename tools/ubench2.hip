// ubench2.hip -- follow-up: select / compare / LDS instruction costs on gfx950 (lone wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define BENCH(NAME, ASM)                                                                        \
  __global__ void NAME(unsigned long long* out, int iseed) {                                   \
    __shared__ int lds[4096];                                                                   \
    int i0 = iseed + threadIdx.x, i1 = iseed + 1, i2 = iseed + 2, i3 = iseed + 3;               \
    int addr = (threadIdx.x & 63) * 16;                                                         \
    lds[threadIdx.x] = iseed;                                                                   \
    __syncthreads();                                                                            \
    unsigned long long t0, t1;                                                                  \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                   \
    for (int it = 0; it < 16; ++it) {                                                           \
      asm volatile(REP64(ASM) : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(addr), "s"(iseed) : "vcc", "s10", "s11", "memory"); \
    }                                                                                           \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                  \
    if (i0 + i1 + i2 + i3 == 123456789) out[1] = lds[5];                                        \
  }
BENCH(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc\n")
BENCH(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n")
BENCH(k_cmp_cnd, "v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %3, vcc\n")
BENCH(k_cmp_cnd_sgpr, "v_cmp_lt_i32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %2, %3, s[10:11]\n")
BENCH(k_cmp_only, "v_cmp_lt_i32 vcc, %0, %1\n")
BENCH(k_cmp4_then_cnd, "v_cmp_lt_i32 vcc, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %1\n v_cndmask_b32 %0, %2, %3, vcc\n")
BENCH(k_max_i32, "v_max_i32 %0, %0, %1\n")
BENCH(k_max_dpp, "s_nop 1\n v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
BENCH(k_ds_read128_dep, "ds_read_b128 v[100:103], %4\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xff0, v100\n")
BENCH(k_ds_read128_pipe, "ds_read_b128 v[100:103], %4\n")
BENCH(k_ds_write128, "ds_write_b128 %4, v[100:103]\n")
BENCH(k_ds_wr_rd, "ds_write_b128 %4, v[100:103]\n ds_read_b128 v[104:107], %4\n s_waitcnt lgkmcnt(0)\n")
BENCH(k_readlane, "v_readfirstlane_b32 s10, %0\n v_add_u32 %0, s10, %0\n")
BENCH(k_and_or, "v_and_or_b32 %0, %0, %1, %2\n")
BENCH(k_bfe, "v_bfe_u32 %0, %0, 3, 5\n")
BENCH(k_salu, "s_add_u32 s10, s10, %5\n")
BENCH(k_branch, "s_cmp_lg_u32 %5, 0\n s_cbranch_scc0 1f\n v_add_u32 %0, %0, %1\n1:\n")
#define RUN(NAME, NI)                                                                          \
  do {                                                                                         \
    hipMemset(d, 0, 64);                                                                       \
    hipLaunchKernelGGL(NAME, dim3(256), dim3(256), 0, 0, d, 3);                                 \
    hipDeviceSynchronize();                                                                    \
    unsigned long long h[2];                                                                   \
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);                                                \
    printf("%-20s %.2f cyc per group of %d instr\n", #NAME, (double)h[0] / (16.0 * 64.0), NI); \
  } while (0)
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  RUN(k_cndmask_vcc, 1); RUN(k_cndmask_sgpr, 1); RUN(k_cmp_cnd, 2); RUN(k_cmp_cnd_sgpr, 2); RUN(k_cmp_only, 1);
  RUN(k_cmp4_then_cnd, 5); RUN(k_max_i32, 1); RUN(k_max_dpp, 2); RUN(k_ds_read128_dep, 3); RUN(k_ds_read128_pipe, 1);
  RUN(k_ds_write128, 1); RUN(k_ds_wr_rd, 3); RUN(k_readlane, 2); RUN(k_and_or, 1); RUN(k_bfe, 1); RUN(k_salu, 1); RUN(k_branch, 3);
  return 0;
}
