// klat_pack_common.h -- device helpers shared by the two stream packers (klat_pack_device.hip: the general passes;
// klat_pack_lean.hip: the passes for lattices whose arcs are short).
#ifndef KLAT_PACK_COMMON_H_
#define KLAT_PACK_COMMON_H_
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_math.h"
#include "klat_pack_device.h"

namespace klat {
namespace {

__device__ __forceinline__ bool dev_is_final(const PackDevArgs& a, int64_t gs) {
  return !(a.final_graph[gs] == INFINITY && a.final_acoustic[gs] == INFINITY);
}

// exclusive prefix sum of one int per thread; returns the thread's prefix, *total = block sum
template <int T>
__device__ int block_scan(int v, int* total, int* sh /* [T/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();  // sh may still be read from a previous call
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < T / 64; ++w) {
    if (w < wave) off += sh[w];
    tot += sh[w];
  }
  *total = tot;
  return off + incl - v;
}

__device__ __forceinline__ void report_bad(const PackDevArgs& a, int code, int u) {
  if (atomicCAS(&a.status[0], 0, code) == 0) a.status[1] = u;
}

constexpr int kM = 0x7FFFFFFF;

struct TermCtx {
  int lg, W, LPW, k;  // lg = log2(lanes per lattice): no integer division in this kernel (runtime divisors cost ~40 instructions each)
};
// term word of term j of position p (first round r, nr rounds) of the lattice in slot k
__device__ __forceinline__ uint32_t term_word(const TermCtx& c, int p, int r, int nr, int j, int rel, bool word, bool nil, bool* far) {
  const uint32_t cell_mask = (uint32_t)(c.W - 1);
  const uint32_t own_cell = (uint32_t)((p & cell_mask) * c.LPW + c.k);
  const int rr = r + (j >> c.lg);
  uint32_t w;
  *far = false;
  if (nil) {
    w = kNil | own_cell;
  } else if (rel == 0) {
    w = kOne | own_cell;
  } else if (rel >= c.W) {
    w = kFar | (uint32_t)rel;
    *far = true;
  } else {
    w = (uint32_t)(((p - rel) & cell_mask) * c.LPW + c.k);
    if (rel == 1 && (j >> c.lg) == 0) w |= kBypass;
  }
  if (word) w |= kWord;
  if (!*far) w |= (uint32_t)(c.W * c.LPW + c.k) << kWrShift;  // the lattice's dummy cell unless COMMIT
  if (rr == r) w |= kFirst;
  if (rr == r + nr - 1) {
    w |= kLast;
    if ((j & ((1 << c.lg) - 1)) == 0 && !*far) w = (w & ~(kCellMask << kWrShift)) | kCommit | (own_cell << kWrShift);
  }
  return w;
}

}  // namespace
}  // namespace klat
#endif
