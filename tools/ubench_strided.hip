// ubench_strided.hip -- can the lockstep streams be written LATTICE-major?  A job's stream interleaves its 64/G lattices
// 16 bytes at a time (klat_device.h: unit = 4 rounds of one lane); pack_fill_kernel assembles whole tiles per JOB in LDS
// so that they leave as full lines.  A workgroup per LATTICE would write 16-byte pieces 256 bytes apart (G = 4) and rely
// on L2 / Infinity Cache to merge the pieces of the job's 16 lattices before they reach HBM.  This measures that:
//   mode 0  job-major: one workgroup per (job, 8 tiles of 32 rounds), full lines (the reference point)
//   mode 1  lattice-major, one lattice a workgroup (16-byte pieces)
//   mode 2  two lattices a workgroup (32-byte pieces)     mode 3  four (64-byte)     mode 4  eight (128-byte: whole lines)
//   xcd = 1: the workgroups of one job get block ids with equal id % 8 (one XCD's L2 under round-robin placement), next to
//   each other in dispatch order; jitter = n: every workgroup idles hash % n microseconds-ish between tiles (drift).
// usage: ubench_strided [mode] [xcd] [jitter] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kJobs = 1250, kChunks = 576, kLPW = 16;  // C2: 10 000 lattices, G = 4, 2 304 rounds
struct Args {
  uint4* sw;
  uint4* sg;
  uint2* sl;
  int mode, xcd, jitter, per_wg;
};

__device__ __forceinline__ void idle(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
}

// job j of direction dir (AL = 4 forward, 3 backward): first unit
__device__ __forceinline__ long long job_base(int dir, int j) {
  const long long fwd_units = (long long)kChunks * 4 * kLPW, bwd_units = (long long)kChunks * 3 * kLPW;
  return dir == 0 ? (long long)j * fwd_units : (long long)kJobs * fwd_units + (long long)j * bwd_units;
}

__global__ __launch_bounds__(256) void job_major(Args a) {
  // block = (dir, job, run of 8 tiles = 64 chunks)
  const int runs = kChunks / 64;
  int b = blockIdx.x;
  const int run = b % runs;
  b /= runs;
  const int dir = b / kJobs, j = b % kJobs, AL = dir == 0 ? 4 : 3;
  const long long base = job_base(dir, j) + (long long)run * 64 * AL * kLPW;
  for (int t = 0; t < 8; ++t) {
    const int units = 8 * AL * kLPW;  // one tile: 8 chunks
    for (int i = threadIdx.x; i < units; i += 256) {
      const long long u = base + (long long)t * units + i;
      a.sw[u] = make_uint4(i, t, j, dir);
      a.sg[u] = make_uint4(i, t, j, 1);
      a.sl[u] = make_uint2(i, t);
    }
    if (a.jitter) idle((blockIdx.x * 2654435761u >> 20) % a.jitter);
  }
}

__global__ __launch_bounds__(256) void lattice_major(Args a) {
  // block -> (dir, job, group of per_wg lattice slots)
  const int per = a.per_wg, groups = kLPW / per;
  int b = blockIdx.x, dir, j, grp;
  if (a.xcd) {
    // blocks b with equal b % 8 share an XCD: a job's `groups` blocks are b = (jj * groups + grp) * 8 + x
    const int x = b & 7, q = b >> 3;
    grp = q % groups;
    const int jj = q / groups;        // job among this XCD's
    const int job_all = jj * 8 + x;   // 0 .. 2 * kJobs - 1 (+ padding)
    if (job_all >= 2 * kJobs) return;
    dir = job_all / kJobs, j = job_all % kJobs;
  } else {
    grp = b % groups;
    b /= groups;
    if (b >= 2 * kJobs) return;
    dir = b / kJobs, j = b % kJobs;
  }
  const int AL = dir == 0 ? 4 : 3;
  const long long base = job_base(dir, j);
  const int k0 = grp * per;
  // tiles of 16 chunks (64 rounds): 16 * AL * per units a tile
  for (int c0 = 0; c0 < kChunks; c0 += 16) {
    const int units = 16 * AL * per;
    for (int i = threadIdx.x; i < units; i += 256) {
      const int kk = i % per, sub = (i / per) % AL, c = c0 + i / (per * AL);
      const long long u = base + (long long)c * AL * kLPW + sub * kLPW + k0 + kk;
      a.sw[u] = make_uint4(i, c, j, dir);
      a.sg[u] = make_uint4(i, c, j, 1);
      a.sl[u] = make_uint2(i, c);
    }
    if (a.jitter) idle(((blockIdx.x * 2654435761u + c0 * 40503u) >> 20) % a.jitter);
  }
}

int main(int argc, char** argv) {
  Args a;
  a.mode = argc > 1 ? atoi(argv[1]) : 1;
  a.xcd = argc > 2 ? atoi(argv[2]) : 1;
  a.jitter = argc > 3 ? atoi(argv[3]) : 0;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  const long long units = (long long)kJobs * kChunks * kLPW * 7;
  hipMalloc(&a.sw, units * 16), hipMalloc(&a.sg, units * 16), hipMalloc(&a.sl, units * 8);
  hipMemset(a.sw, 0, units * 16);
  a.per_wg = a.mode == 0 ? 0 : 1 << (a.mode - 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0, 0);
    if (a.mode == 0) {
      hipLaunchKernelGGL(job_major, dim3(2 * kJobs * (kChunks / 64)), dim3(256), 0, 0, a);
    } else {
      const int groups = kLPW / a.per_wg;
      const int jobs_pad = (2 * kJobs + 7) / 8 * 8;
      hipLaunchKernelGGL(lattice_major, dim3(jobs_pad * groups), dim3(256), 0, 0, a);
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d xcd %d jitter %d: %.3f ms, %.2f GB written -> %.0f GB/s\n", a.mode, a.xcd, a.jitter, ms, units * 40 / 1e9, units * 40 / 1e6 / ms);
  }
  // spot check: every unit written
  std::vector<uint4> h(1 << 16);
  hipMemcpy(h.data(), a.sw + units - (1 << 16), (size_t)(1 << 16) * 16, hipMemcpyDeviceToHost);
  long long zero = 0;
  for (auto& v : h) zero += (v.w == 0 && v.z == 0 && v.x == 0 && v.y == 0);
  printf("unwritten units in the last 64k: %lld\n", zero);
  return 0;
}
