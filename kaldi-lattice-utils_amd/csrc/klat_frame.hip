// klat_frame.hip -- frame posteriors (a9 + a12 of SURVEY.md 8a).
//
// latbin/lattice-to-word-frame-post.cc:93-117: for every word arc (ilabel != 0) and every frame
// k in [times[src], times[dst]): acc[k][word] = LogAdd(acc[k][word], x) with
//   x = fw[src] + bw[dst] - (float)(Value1 + Value2)          (a FLOAT sum, :102-104)
// then :121-135: logp = acc - total narrowed to float, per-frame sort by (logp desc, label asc).
//
// One workgroup owns one lattice: arcs are expanded into (frame,label | arc) keys, the keys are
// bitonic-sorted, runs of equal (frame,label) are folded in arc order with the reference's
// LogAdd, and the folded entries are sorted by (frame, float logp desc, label).  The gather
// kernel lays them out densely with per-frame offsets.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_accum.h"
#include "klat_math.h"

namespace klat {

static constexpr int kFrThreads = 512;
static constexpr int kFrTile = 2048;  // 16-byte keys per LDS tile (32 KiB)

__device__ __forceinline__ double fr_log_add(double x, double y) {
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  if (diff >= -36.04365338911715) return x + log1p(exp(diff));  // log(DBL_EPSILON)
  return x;
}

__device__ __forceinline__ uint32_t fr_bias(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
__device__ __forceinline__ int32_t fr_unbias(uint32_t v) { return (int32_t)(v ^ 0x80000000u); }
// descending float -> ascending uint32
__device__ __forceinline__ uint32_t fr_desc_key(float x) {
  uint32_t b = __float_as_uint(x);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ~b;
}
__device__ __forceinline__ float fr_desc_unkey(uint32_t k) {
  uint32_t b = ~k;
  b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
  return __uint_as_float(b);
}

struct FKey {
  uint64_t a;  // pass 1: frame|label      pass 2: frame|desc(float logp)
  uint64_t b;  // pass 1: arc              pass 2: label
};
struct FKeyLess {
  __device__ bool operator()(const FKey& x, const FKey& y) const { return x.a < y.a || (x.a == y.a && x.b < y.b); }
};

template <class Less>
__device__ void fr_bitonic_sort(FKey* data, int n, FKey* tile, Less less) {
  const int tid = threadIdx.x;
  if (n <= kFrTile) {
    for (int i = tid; i < n; i += kFrThreads) tile[i] = data[i];
    __syncthreads();
    for (int k = 2; k <= n; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < n; i += kFrThreads) {
          const int l = i ^ j;
          if (l > i) {
            const FKey x = tile[i], y = tile[l];
            const bool up = (i & k) == 0;
            if (up ? less(y, x) : less(x, y)) {
              tile[i] = y;
              tile[l] = x;
            }
          }
        }
        __syncthreads();
      }
    for (int i = tid; i < n; i += kFrThreads) data[i] = tile[i];
    __syncthreads();
    return;
  }
  for (int k = 2; k <= n; k <<= 1) {
    int j = k >> 1;
    for (; j >= kFrTile; j >>= 1) {
      for (int i = tid; i < n; i += kFrThreads) {
        const int l = i ^ j;
        if (l > i) {
          const FKey x = data[i], y = data[l];
          const bool up = (i & k) == 0;
          if (up ? less(y, x) : less(x, y)) {
            data[i] = y;
            data[l] = x;
          }
        }
      }
      __syncthreads();
    }
    for (int base = 0; base < n; base += kFrTile) {
      for (int i = tid; i < kFrTile; i += kFrThreads) tile[i] = data[base + i];
      __syncthreads();
      for (int jj = j; jj > 0; jj >>= 1) {
        for (int i = tid; i < kFrTile; i += kFrThreads) {
          const int l = i ^ jj;
          if (l > i) {
            const FKey x = tile[i], y = tile[l];
            const bool up = ((base + i) & k) == 0;
            if (up ? less(y, x) : less(x, y)) {
              tile[i] = y;
              tile[l] = x;
            }
          }
        }
        __syncthreads();
      }
      for (int i = tid; i < kFrTile; i += kFrThreads) data[base + i] = tile[i];
      __syncthreads();
    }
  }
}

// block-wide exclusive scan of one int per thread
__device__ int fr_block_scan(int v, int* total, int* sh /* [kFrThreads/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  int pre = 0, tot = 0;
  for (int w = 0; w < kFrThreads / 64; ++w) {
    if (w < wave) pre += sh[w];
    tot += sh[w];
  }
  __syncthreads();
  *total = tot;
  return pre + inc - v;
}

// clipped frame span of arc i of lattice u (0 for epsilon / empty spans)
__device__ __forceinline__ void fr_span(const FrameArgs& a, int64_t a0, int64_t s0, int T, int i, int& k0, int& k1) {
  k0 = k1 = 0;
  if (a.ilabel[a0 + i] == 0) return;
  int t0 = a.state_time[s0 + a.src[a0 + i]], t1 = a.state_time[s0 + a.dst[a0 + i]];
  t0 = max(t0, 0);
  t1 = min(t1, T);
  if (t1 > t0) {
    k0 = t0;
    k1 = t1;
  }
}

// Number of (frame,label) expansions per lattice, to size the scratch.
__global__ __launch_bounds__(256) void frame_count_kernel(FrameArgs a) {
  __shared__ unsigned long long s_sum;
  const int u = blockIdx.x;
  const int64_t a0 = a.arc_off[u];
  const int nA = (int)(a.arc_off[u + 1] - a0);
  const int64_t s0 = a.state_off[u];
  const int T = a.num_frames[u];
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  unsigned long long mine = 0;
  for (int i = threadIdx.x; i < nA; i += 256) {
    int k0, k1;
    fr_span(a, a0, s0, T, i, k0, k1);
    mine += (unsigned long long)(k1 - k0);
  }
  atomicAdd(&s_sum, mine);
  __syncthreads();
  if (threadIdx.x == 0) a.num_expansions[u] = (int64_t)s_sum;
}

__global__ __launch_bounds__(kFrThreads) void frame_kernel(FrameArgs a) {
  __shared__ __attribute__((aligned(16))) FKey tile[kFrTile];
  __shared__ int sh[kFrThreads / 64];
  const int u = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t a0 = a.arc_off[u];
  const int nA = (int)(a.arc_off[u + 1] - a0);
  const int64_t s0 = a.state_off[u];
  const int T = a.num_frames[u];
  FKey* keys = reinterpret_cast<FKey*>(a.key_scratch) + a.scratch_off[u];
  FKey* recs = reinterpret_cast<FKey*>(a.rec_scratch) + a.scratch_off[u];
  const ScaleParams sp = a.sp;

  // ---- 1. expand arcs into (frame,label | arc) keys ----
  int n = 0;
  for (int base = 0; base < nA; base += kFrThreads) {
    const int i = base + tid;
    int k0 = 0, k1 = 0;
    if (i < nA) fr_span(a, a0, s0, T, i, k0, k1);
    int tot;
    const int off = fr_block_scan(k1 - k0, &tot, sh);
    if (k1 > k0) {
      const uint32_t lab = fr_bias(a.ilabel[a0 + i]);
      for (int k = k0; k < k1; ++k) {
        FKey key;
        key.a = ((uint64_t)(uint32_t)k << 32) | lab;
        key.b = (uint32_t)i;
        keys[n + off + (k - k0)] = key;
      }
    }
    n += tot;
  }
  int np = 1;
  while (np < n) np <<= 1;
  if (n == 0) np = 0;
  for (int i = n + tid; i < np; i += kFrThreads) {
    FKey pad;
    pad.a = pad.b = ~0ull;
    keys[i] = pad;
  }
  __syncthreads();
  if (np > 1) fr_bitonic_sort(keys, np, tile, FKeyLess());

  // ---- 2. fold runs of equal (frame,label) in arc order ----
  const double tot_u = a.total[u];
  (void)tot_u;
  int m = 0;
  for (int base = 0; base < n; base += kFrThreads) {
    const int i = base + tid;
    bool head = false;
    FKey rec;
    rec.a = rec.b = 0;
    if (i < n) {
      const FKey ki = keys[i];
      head = (i == 0) || (keys[i - 1].a != ki.a);
      if (head) {
        double acc = 0.0;
        for (int j = i; j < n; ++j) {
          const FKey kj = keys[j];
          if (kj.a != ki.a) break;
          const int64_t arc = a0 + (int64_t)(uint32_t)kj.b;
          // logpost holds alpha - (double)g - (double)a + beta - total; the reference forms the
          // arc cost here as a float sum, so swap the double sum for the float one
          float g, ac;
          arc_cost_parts(a.graph[arc], a.acoustic[arc], true, sp, g, ac);
          // (an arc of infinite cost is log-zero either way; the correction would be inf - inf)
          const double lpa = a.logpost[arc];
          const double x = (lpa > -INFINITY) ? lpa + ((double)g + (double)ac) - (double)(g + ac) : lpa;
          acc = (j == i) ? x : fr_log_add(acc, x);
        }
        const float lp = (float)acc;  // BaseFloat narrowing of :126
        rec.a = (ki.a & 0xFFFFFFFF00000000ull) | fr_desc_key(lp);
        rec.b = (uint32_t)ki.a;  // biased label
      }
    }
    // compact heads
    const uint64_t ballot = __ballot(head);
    const int lane = tid & 63, wave = tid >> 6;
    const int within = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) sh[wave] = __popcll(ballot);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < kFrThreads / 64; ++w) {
      if (w < wave) off += sh[w];
      tot += sh[w];
    }
    __syncthreads();
    if (head) recs[m + off + within] = rec;
    m += tot;
  }
  int mp = 1;
  while (mp < m) mp <<= 1;
  if (m == 0) mp = 0;
  for (int i = m + tid; i < mp; i += kFrThreads) {
    FKey pad;
    pad.a = pad.b = ~0ull;
    recs[i] = pad;
  }
  __syncthreads();
  // ---- 3. per-frame order: float logp descending, then label ascending ----
  if (mp > 1) fr_bitonic_sort(recs, mp, tile, FKeyLess());
  if (tid == 0) a.num_records[u] = m;
}

// Dense output: entries of lattice u at entry_off[u].., frame offsets at frame_base[u]...
__global__ __launch_bounds__(256) void frame_gather_kernel(FrameArgs a, const int64_t* entry_off,
                                                           const int64_t* frame_base, int64_t* frame_off,
                                                           int32_t* labels, float* logps) {
  const int u = blockIdx.x;
  const FKey* recs = reinterpret_cast<const FKey*>(a.rec_scratch) + a.scratch_off[u];
  const int64_t o = entry_off[u];
  const int m = (int)(entry_off[u + 1] - o);
  const int64_t fb = frame_base[u];
  const int T = a.num_frames[u];
  for (int i = threadIdx.x; i < m; i += 256) {
    const FKey r = recs[i];
    labels[o + i] = fr_unbias((uint32_t)r.b);
    logps[o + i] = fr_desc_unkey((uint32_t)r.a);
    const int f = (int)(uint32_t)(r.a >> 32);
    const int fp = (i == 0) ? -1 : (int)(uint32_t)(recs[i - 1].a >> 32);
    for (int g = fp + 1; g <= f; ++g) frame_off[fb + g] = o + i;
  }
  if (threadIdx.x == 0) {
    const int fl = (m == 0) ? -1 : (int)(uint32_t)(recs[m - 1].a >> 32);
    for (int g = fl + 1; g < T; ++g) frame_off[fb + g] = o + m;
  }
}

hipError_t launch_frame_count(const FrameArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(frame_count_kernel, dim3(num_lattices), dim3(256), 0, stream, a);
  return hipGetLastError();
}
hipError_t launch_frame(const FrameArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(frame_kernel, dim3(num_lattices), dim3(kFrThreads), 0, stream, a);
  return hipGetLastError();
}
hipError_t launch_frame_gather(const FrameArgs& a, int num_lattices, const int64_t* entry_off,
                               const int64_t* frame_base, int64_t* frame_off, int32_t* labels, float* logps,
                               hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(frame_gather_kernel, dim3(num_lattices), dim3(256), 0, stream, a, entry_off, frame_base,
                     frame_off, labels, logps);
  return hipGetLastError();
}

}  // namespace klat
