// Where do the two waves of a 128-thread workgroup land?  Records HW_ID / XCC_ID of every wave of
// a grid shaped like the sweep's (1250 workgroups x 2 waves, ~30 KB LDS each) and prints, per CU,
// how many "wave 0"s and "wave 1"s each SIMD received.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_place.hip -o tools/ubench_place.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(128) void place(uint32_t* out, int spin) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6;
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 0.5);
  lds[threadIdx.x] = x;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 2 + wave) * 2 + 0] = hw;
    out[(blockIdx.x * 2 + wave) * 2 + 1] = xcc;
  }
  if (lds[(threadIdx.x + 1) & 127] == 12345.678) out[0] = 0;
}
int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 1250;
  uint32_t* d;
  hipMalloc(&d, sizeof(uint32_t) * nwg * 4);
  hipLaunchKernelGGL(place, dim3(nwg), dim3(128), 30 * 1024, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(nwg * 4);
  hipMemcpy(h.data(), d, sizeof(uint32_t) * nwg * 4, hipMemcpyDeviceToHost);
  // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
  std::map<uint32_t, std::vector<int>> cu;  // key (xcc,se,sh,cu) -> counts [simd][wave]
  int same_simd = 0, slot_differs = 0;
  for (int b = 0; b < nwg; ++b) {
    uint32_t hw0 = h[(b * 2) * 2], hw1 = h[(b * 2 + 1) * 2];
    for (int w = 0; w < 2; ++w) {
      const uint32_t hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1] & 0xf;
      const uint32_t key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xf);
      auto& v = cu[key];
      if (v.empty()) v.assign(8, 0);
      v[((hw >> 4) & 3) * 2 + w]++;
    }
    same_simd += ((hw0 >> 4) & 3) == ((hw1 >> 4) & 3);
    slot_differs += (hw0 & 0xf) != (hw1 & 0xf);
    if (b < 24) printf("wg %4d  w0: simd %u slot %u cu %03x xcc %u | w1: simd %u slot %u\n", b, (hw0 >> 4) & 3, hw0 & 0xf,
                       (hw0 >> 8) & 0xff, h[(b * 2) * 2 + 1] & 0xf, (hw1 >> 4) & 3, hw1 & 0xf);
  }
  printf("CUs seen %zu, workgroups with both waves on one SIMD %d, with different slot ids %d\n", cu.size(), same_simd, slot_differs);
  std::map<std::vector<int>, int> hist;
  for (auto& kv : cu) hist[kv.second]++;
  for (auto& kv : hist) {
    printf("%4d CUs: ", kv.second);
    for (int s = 0; s < 4; ++s) printf(" simd%d[w0=%d w1=%d]", s, kv.first[s * 2], kv.first[s * 2 + 1]);
    printf("\n");
  }
  return 0;
}
