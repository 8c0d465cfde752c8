// klat_slice.h -- where a long lattice may be cut (index_slice_kernel, klat_accum_hash.hip; frame_slice_kernel,
// klat_frame.hip).
//
// Both accumulators cut a sequence -- the arcs by their source states' times for the index, the states by their times
// for the frame posteriors -- at a position i such that everything before i lies strictly below everything from i on:
//     max t[0 .. i-1]  <  min t[i .. n-1].
// For times that never descend that is every place where the time changes; times that descend here and there (a
// top-sorted lattice need not be time-sorted) still leave such places wherever the disorder is local.  The maxima and
// minima of every block of 64 positions go to LDS, one thread turns them into the maxima before and the minima from
// every block on, and a wave tests 64 aligned positions at a time with a prefix maximum / suffix minimum over its lanes.
#ifndef KLAT_SLICE_H_
#define KLAT_SLICE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace klat {

static constexpr int kSliceBlocks = 4096;  // blocks of 64 positions a lattice may have (262 144 arcs or states)

// before[b] = max t over the blocks before b, from[b] = min t over the blocks from b on (from[nb] = INT_MAX).
// All threads of the workgroup (whole waves) call it; the arrays hold kSliceBlocks + 1 entries.
template <class TimeOf>
__device__ __forceinline__ void slice_prepare(int n, TimeOf t, int32_t* before, int32_t* from) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int nb = (n + 63) >> 6;
  for (int b0 = wave * 8; b0 < nb; b0 += waves * 8) {  // (wave-uniform; eight blocks' loads in flight together: t() is two dependent loads)
    int32_t v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = (b0 + q) * 64 + lane;
      v[q] = i < n ? t(i) : 0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = (b0 + q) * 64 + lane;
      int32_t hi = i < n ? v[q] : (int32_t)0x80000000, lo = i < n ? v[q] : 0x7FFFFFFF;
      for (int d = 32; d > 0; d >>= 1) hi = max(hi, __shfl_xor(hi, d, 64)), lo = min(lo, __shfl_xor(lo, d, 64));
      if (lane == 0 && b0 + q < nb) before[b0 + q] = hi, from[b0 + q] = lo;  // (the block's own extremes for now)
    }
  }
  __syncthreads();
  // the maxima before every block, the minima from every block on: one wave each, 64 blocks a step (a workgroup of a
  // single wave takes both in turn)
  if (wave == 0) {
    int32_t run = (int32_t)0x80000000;
    for (int b0 = 0; b0 < nb; b0 += 64) {
      const int b = b0 + lane;
      const int32_t own = b < nb ? before[b] : (int32_t)0x80000000;
      int32_t up = own;
      for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(up, d, 64);
        if (lane >= d) up = max(up, o);
      }
      int32_t excl = __shfl_up(up, 1, 64);
      excl = lane == 0 ? run : max(run, excl);
      if (b < nb) before[b] = excl;
      run = max(run, __shfl(up, 63, 64));
    }
  }
  if (wave == (waves > 1 ? 1 : 0)) {
    int32_t run = 0x7FFFFFFF;
    if (lane == 0) from[nb] = run;
    for (int b1 = nb; b1 > 0; b1 -= 64) {  // blocks [b1 - 64, b1), lane l takes block b1 - 1 - l
      const int b = b1 - 1 - lane;
      const int32_t own = b >= 0 ? from[b] : 0x7FFFFFFF;
      int32_t dn = own;
      for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(dn, d, 64);
        if (lane >= d) dn = min(dn, o);
      }
      dn = min(dn, run);
      if (b >= 0) from[b] = dn;
      run = __shfl(dn, 63, 64);
    }
  }
  __syncthreads();
}

// The valid cut among the positions [lo, hi] (1 <= lo, hi < n): the largest one if `backwards`, the smallest else; -1 if
// there is none.  A whole wave calls it (wave-uniform arguments).
template <class TimeOf>
__device__ __forceinline__ int slice_find(int n, TimeOf t, const int32_t* before, const int32_t* from, int lo, int hi, bool backwards,
                                          int32_t* least_from_cut = nullptr) {
  const int lane = threadIdx.x & 63;
  const int b_lo = lo >> 6, b_hi = hi >> 6;
  for (int k = 0; k <= b_hi - b_lo; ++k) {
    const int b = backwards ? b_hi - k : b_lo + k;
    const int i = b * 64 + lane;
    const int32_t v = i < n ? t(i) : 0;
    // max of the positions before mine: the blocks before, and the lanes below
    int32_t up = i < n ? v : (int32_t)0x80000000;
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t o = __shfl_up(up, d, 64);
      if (lane >= d) up = max(up, o);
    }
    int32_t below = __shfl_up(up, 1, 64);
    below = lane == 0 ? before[b] : max(before[b], below);
    // min of the positions from mine on: the lanes from mine, and the blocks behind
    int32_t down = i < n ? v : 0x7FFFFFFF;
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t o = __shfl_down(down, d, 64);
      if (lane + d < 64) down = min(down, o);
    }
    down = min(down, from[b + 1]);
    const unsigned long long ok = __ballot(i >= lo && i <= hi && i < n && below < down);
    if (ok) {
      const int at = backwards ? 63 - __clzll((long long)ok) : __ffsll((long long)ok) - 1;
      if (least_from_cut) *least_from_cut = __shfl(down, at, 64);  // the smallest t from the cut on
      return b * 64 + at;
    }
  }
  return -1;
}

}  // namespace klat
#endif
