// ubench3.hip -- LDS access width costs for a lone wave (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define BENCH(NAME, ASM)                                                                        \
  __global__ void NAME(unsigned long long* out, int iseed) {                                   \
    __shared__ int lds[8192];                                                                   \
    int i0 = iseed + threadIdx.x;                                                               \
    int addr = (threadIdx.x & 63) * 16, addr8 = (threadIdx.x & 63) * 8, addr4 = (threadIdx.x & 63) * 4 + 16384; \
    lds[threadIdx.x] = iseed;                                                                   \
    __syncthreads();                                                                            \
    unsigned long long t0, t1;                                                                  \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                   \
    for (int it = 0; it < 16; ++it) {                                                           \
      asm volatile(REP64(ASM) : "+v"(i0) : "v"(addr), "v"(addr8), "v"(addr4) : "memory", "v100", "v101", "v102", "v103", "v104", "v105"); \
    }                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                  \
    if (i0 == 123456789) out[1] = lds[5];                                                       \
  }
BENCH(k_wr128, "ds_write_b128 %1, v[100:103]\n")
BENCH(k_wr64, "ds_write_b64 %2, v[100:101]\n")
BENCH(k_wr32, "ds_write_b32 %3, v100\n")
BENCH(k_wr64_32, "ds_write_b64 %2, v[100:101]\n ds_write_b32 %3, v102\n")
BENCH(k_rd128, "ds_read_b128 v[100:103], %1\n")
BENCH(k_rd64, "ds_read_b64 v[100:101], %2\n")
BENCH(k_rd32, "ds_read_b32 v100, %3\n")
BENCH(k_rd64_32, "ds_read_b64 v[100:101], %2\n ds_read_b32 v102, %3\n")
BENCH(k_rd64_dep, "ds_read_b64 v[100:101], %2\n s_waitcnt lgkmcnt(0)\n")
BENCH(k_wr64_valu8, "ds_write_b64 %2, v[100:101]\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n")
BENCH(k_wr128_valu8, "ds_write_b128 %1, v[100:103]\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n")
BENCH(k_valu8, "v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n")
#define RUN(NAME, NI)                                                                          \
  do {                                                                                         \
    hipMemset(d, 0, 64);                                                                       \
    hipLaunchKernelGGL(NAME, dim3(256), dim3(256), 0, 0, d, 3);                                 \
    hipDeviceSynchronize();                                                                    \
    unsigned long long h[2];                                                                   \
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);                                                \
    printf("%-16s %.2f cyc per group of %d instr\n", #NAME, (double)h[0] / (16.0 * 64.0), NI);  \
  } while (0)
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  RUN(k_wr128, 1); RUN(k_wr64, 1); RUN(k_wr32, 1); RUN(k_wr64_32, 2); RUN(k_rd128, 1); RUN(k_rd64, 1); RUN(k_rd32, 1);
  RUN(k_rd64_32, 2); RUN(k_rd64_dep, 2); RUN(k_wr64_valu8, 9); RUN(k_wr128_valu8, 9); RUN(k_valu8, 8);
  return 0;
}
