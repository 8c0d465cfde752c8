// ubench_clk.hip -- s_memtime vs s_memrealtime (100 MHz) around a VALU loop: what a "tick" is.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__global__ void k(unsigned long long* out, float seed, int iters) {
  float f = seed;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) asm volatile(REP64("v_fma_f32 %0, %0, %0, %0\n") : "+v"(f));
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (f == 12345.f) out[2] = 1;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  for (int blocks : {1, 256, 1024}) for (int iters : {1000, 20000}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.5f, iters);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    double ns = h[1] * 10.0;
    printf("blocks=%d iters=%d: memtime ticks=%llu realtime=%.0f ns -> %.3f ticks/ns ; %.2f ticks/instr ; %.2f ns/instr\n", blocks, iters,
           h[0], ns, h[0] / ns, (double)h[0] / (64.0 * iters), ns / (64.0 * iters));
  }
  return 0;
}
