// klat_frame.hip -- frame posteriors (a9 + a12 of SURVEY.md 8a).
//
// latbin/lattice-to-word-frame-post.cc:93-117: for every word arc (ilabel != 0) and every frame
// k in [times[src], times[dst]): acc[k][word] = LogAdd(acc[k][word], x) with
//   x = fw[src] + bw[dst] - (float)(Value1 + Value2)          (a FLOAT sum, :102-104)
// then :121-135: logp = acc - total narrowed to float, per-frame sort by (logp desc, label asc).
//
// Two kernels.  frame_lean_kernel (below, "the lean kernel") takes the lattices that fit it -- on the synthetic
// and the shipped lattices all of them -- and lists the others for frame_kernel, the general one, described
// first: the lean kernel is the same plan over the lattice's distinct state times instead of its frames, with
// its records out of a pool.
//
// One workgroup owns one lattice.  A frame holds only the few words whose arcs cross it, so the
// (frame, word) expansions are bucketed by frame instead of sorted as a whole:
//   0. every arc forms its clipped span and its x once (16 B per arc in scratch);
//   per tile of kFrTileFrames frames, with the per-frame counters in LDS:
//   1. entries per frame from a difference array over the spans (two LDS atomics per arc),
//      prefix sums -> bucket offsets;
//   2. every arc drops its number into the buckets of its frames (LDS cursors).  The buckets of a
//      tile live in LDS as 16-bit arc numbers -- 2 bytes per (frame, word) expansion instead of a
//      16-byte entry written to and read back from HBM -- unless the tile holds more than
//      kFrLdsEntries expansions, a frame more than kFrSerialMax, or the lattice more than 65 535
//      arcs: then they go through the lattice's scratch rows as (label | arc, x) entries;
//   3. one thread per frame, 256 frames at a time: a bucket of <= 16 entries is sorted by (label,
//      arc) in registers (bitonic network, 8 or 16 wide), runs of one label are folded in arc
//      order with the reference's LogAdd and narrowed to float, the folded keys (float logp desc,
//      label asc) sorted by the same network; a block scan places the frame's records, frames
//      ascending.  Buckets of up to 256 entries take insertion sorts in place (L2), larger ones
//      are sorted by the whole workgroup first (bitonic network through an LDS tile).
// The gather kernel lays the records out densely with per-frame offsets.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <math.h>
#include <stdint.h>

#include "klat_accum.h"
#include "klat_slice.h"
#include "klat_math.h"

namespace klat {

static constexpr int kFrThreads = 256;
static constexpr int kFrLeanThreads = 512;  // the lean kernel's workgroup (frame_lean_kernel)
static constexpr int kFrTile = 2048;    // 16-byte keys per LDS tile of the cooperative sort (32 KiB)
static constexpr int kFrRegs = 16;      // bucket entries a frame's thread sorts in registers
static constexpr int kFrBigList = 64;
static constexpr int kFrTileFrames = 1920;  // frames whose counters live in LDS at a time (15 KiB; two workgroups per CU)
static constexpr int kFrLdsEntries = 16384; // bucket entries of a tile kept in LDS as 16-bit arc numbers (32 KiB, over the sort tile)
static constexpr int kFrSpanBatch = 8;      // arcs a thread loads together in the span pass
static constexpr int kFrLdsSpans = 6144;    // arcs whose clipped spans stay in LDS as two 16-bit frame numbers (24 KiB: two workgroups per CU)

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

// The same LogAdd for the lean kernel, ~45 instructions instead of the library's ~200 and no constants held
// in vector registers: exp(d) = 2^k 2^f with 2^f by the degree-11 polynomial of tools/exp2_poly.py (relative
// error 2e-17 before rounding), ln(1 + exp(d)) by the sweep's table ln (klat_math.h); the result is within
// ~3e-16 of the library's before the final sum rounds it (the reference narrows it to float, :126).
__device__ __forceinline__ double fr_log_add_fast(double x, double y, const LnTabEntry* lntab /* LDS */) {
  double hi, lo;
  asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
  asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
  const double d = lo - hi;  // <= 0; -inf or NaN (both log-zero) fail the test below
  const double kf = rint(d * kLog2e);
  const double f = fma(d, kLog2e, -kf);
  int k;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(k) : "v"(kf));  // (saturates)
  double p, c11;
  asm("v_mov_b64 %0, %1" : "=v"(c11) : "s"(4.45581790833606449045e-10));
  asm("v_fma_f64 %0, %1, %2, %3\n\t"
      "v_fma_f64 %0, %0, %2, %4\n\t"
      "v_fma_f64 %0, %0, %2, %5\n\t"
      "v_fma_f64 %0, %0, %2, %6\n\t"
      "v_fma_f64 %0, %0, %2, %7\n\t"
      "v_fma_f64 %0, %0, %2, %8\n\t"
      "v_fma_f64 %0, %0, %2, %9\n\t"
      "v_fma_f64 %0, %0, %2, %10\n\t"
      "v_fma_f64 %0, %0, %2, %11\n\t"
      "v_fma_f64 %0, %0, %2, %12\n\t"
      "v_fma_f64 %0, %0, %2, 1.0"
      : "=&v"(p)
      : "v"(c11), "v"(f), "s"(7.07419429728852105635e-09), "s"(1.01780570877339407473e-07),
        "s"(1.32154325359123752878e-06), "s"(1.52527338415567733184e-05), "s"(1.54035304637243529923e-04),
        "s"(1.33335581464064708429e-03), "s"(9.61812910758725637950e-03), "s"(5.55041086648216247523e-02),
        "s"(2.40226506959101582250e-01), "s"(6.93147180559945286227e-01));
  const double s1 = 1.0 + ldexp(p, k);  // in (1, 2]
  double m;
  int e;
  asm("v_frexp_mant_f64 %0, %1" : "=v"(m) : "v"(s1));
  asm("v_frexp_exp_i32_f64 %0, %1" : "=v"(e) : "v"(s1));
  const double r = hi + table_ln(m, e, lntab[ln_table_index(m)]);
  return select64(lane_mask(d >= -36.04365338911715), r, hi);  // log(DBL_EPSILON)
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
  uint64_t a;  // bucket entry: label|arc   folded: desc(float logp)|label   record: frame|desc(float logp)
  uint64_t b;  // bucket entry: x (double)  folded: 0                        record: label
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

// block-wide exclusive scan of one int per thread (NT threads)
// (TAIL = false: no barrier behind the last read of sh -- the caller does not write sh again before its next barrier)
// inclusive prefix sum over the wave by DPP: four shifts within the rows of sixteen lanes, then the rows' last lanes handed
// on (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- six adds, no trip through the LDS crossbar
__device__ __forceinline__ int fr_wave_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return v;
}
template <int NT = kFrThreads, bool TAIL = true>
__device__ int fr_block_scan(int v, int* total, int* sh /* [NT/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inc = fr_wave_scan(v);
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  int pre = 0, tot = 0;
  for (int w = 0; w < NT / 64; ++w) {
    if (w < wave) pre += sh[w];
    tot += sh[w];
  }
  if (TAIL) __syncthreads();
  *total = tot;
  return pre + inc - v;
}

// the same for two ints per thread at once (one pair of barriers)
template <int NT = kFrThreads, bool TAIL = true>
__device__ int2 fr_block_scan2(int v, int w, int* total_v, int* total_w, int* sh /* [2 * NT/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int iv = fr_wave_scan(v), iw = fr_wave_scan(w);
  if (lane == 63) sh[wave] = iv, sh[NT / 64 + wave] = iw;
  __syncthreads();
  int pv = 0, tv = 0, pw = 0, tw = 0;
  for (int k = 0; k < NT / 64; ++k) {
    if (k < wave) pv += sh[k], pw += sh[NT / 64 + k];
    tv += sh[k], tw += sh[NT / 64 + k];
  }
  if (TAIL) __syncthreads();
  *total_v = tv, *total_w = tw;
  return make_int2(pv + iv - v, pw + iw - w);
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
  for (int base = 0; base < nA; base += 8 * 256) {  // eight arcs per thread at a time: their loads in flight together
    int il[8], t0[8], t1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = min(base + q * 256 + (int)threadIdx.x, nA - 1);
      il[q] = a.ilabel[a0 + i];
      t0[q] = a.src[a0 + i];
      t1[q] = a.dst[a0 + i];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      t0[q] = a.state_time[s0 + t0[q]];
      t1[q] = a.state_time[s0 + t1[q]];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int lo = max(t0[q], 0), hi = min(t1[q], T);
      if (base + q * 256 + (int)threadIdx.x < nA && il[q] != 0 && hi > lo) mine += (unsigned long long)(hi - lo);
    }
  }
  atomicAdd(&s_sum, mine);
  __syncthreads();
  if (threadIdx.x == 0) a.num_expansions[u] = (int64_t)s_sum;
}

// Bitonic network over the first N of 16 register-resident keys (ascending), payload alongside.
// Sizes and strides are template parameters so that every index is a constant and the arrays
// stay in registers.
template <int N, int SIZE, int STRIDE, bool WITH_VAL>
__device__ __forceinline__ void fr_sort_stage(uint64_t (&k)[kFrRegs], double (&v)[kFrRegs]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int l = i ^ STRIDE;
    if (l > i) {
      const bool up = (i & SIZE) == 0;
      // (values first, selects on the values: a select between k[i] and k[l] themselves becomes a
      // select of the index, and the arrays leave the registers)
      const uint64_t ki = k[i], kl = k[l];
      const double vi = v[i], vl = v[l];
      // VOP3 selects with the lane mask in a scalar pair (klat_math.h: runs of the VOP2 / VCC
      // form cost ~16 cycles each once several waves share a CU)
      const uint64_t sw = lane_mask(up ? (ki > kl) : (ki < kl));
      k[i] = (uint64_t)__double_as_longlong(select64(sw, __longlong_as_double((long long)kl), __longlong_as_double((long long)ki)));
      k[l] = (uint64_t)__double_as_longlong(select64(sw, __longlong_as_double((long long)ki), __longlong_as_double((long long)kl)));
      if (WITH_VAL) {
        v[i] = select64(sw, vl, vi);
        v[l] = select64(sw, vi, vl);
      }
    }
  }
  if constexpr (STRIDE > 1) fr_sort_stage<N, SIZE, STRIDE / 2, WITH_VAL>(k, v);
}
template <int N, int SIZE, bool WITH_VAL>
__device__ __forceinline__ void fr_sort_sizes(uint64_t (&k)[kFrRegs], double (&v)[kFrRegs]) {
  fr_sort_stage<N, SIZE, SIZE / 2, WITH_VAL>(k, v);
  if constexpr (SIZE < N) fr_sort_sizes<N, SIZE * 2, WITH_VAL>(k, v);
}
template <int N, bool WITH_VAL>
__device__ __forceinline__ void fr_sort_regs(uint64_t (&k)[kFrRegs], double (&v)[kFrRegs]) {
  fr_sort_sizes<N, 2, WITH_VAL>(k, v);
}

// The same network over 32-bit keys without payload: a compare-exchange is v_min_u32 + v_max_u32.
template <int N, int SIZE, int STRIDE, int R = kFrRegs>
__device__ __forceinline__ void fr_sort32_stage(uint32_t (&k)[R]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int l = i ^ STRIDE;
    if (l > i) {
      const bool up = (i & SIZE) == 0;
      const uint32_t lo = min(k[i], k[l]), hi = max(k[i], k[l]);
      k[i] = up ? lo : hi;
      k[l] = up ? hi : lo;
    }
  }
  if constexpr (STRIDE > 1) fr_sort32_stage<N, SIZE, STRIDE / 2, R>(k);
}
template <int N, int SIZE, int R = kFrRegs>
__device__ __forceinline__ void fr_sort32_sizes(uint32_t (&k)[R]) {
  fr_sort32_stage<N, SIZE, SIZE / 2, R>(k);
  if constexpr (SIZE < N) fr_sort32_sizes<N, SIZE * 2, R>(k);
}

// The same network over keys that are positive finite doubles as bit patterns (top two bits clear):
// their order as doubles is their order as integers, and a compare-exchange is v_min_f64 + v_max_f64
// instead of a 64-bit compare and four selects.  (Bare instructions: fmin / fmax would first quiet
// signalling NaNs, which these patterns never are.)
template <int N, int SIZE, int STRIDE, int R = kFrRegs>
__device__ __forceinline__ void fr_sortd_stage(double (&k)[R]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int l = i ^ STRIDE;
    if (l > i) {
      const bool up = (i & SIZE) == 0;
      double lo, hi;
      asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(k[i]), "v"(k[l]));
      asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(k[i]), "v"(k[l]));
      k[i] = up ? lo : hi;
      k[l] = up ? hi : lo;
    }
  }
  if constexpr (STRIDE > 1) fr_sortd_stage<N, SIZE, STRIDE / 2, R>(k);
}
template <int N, int SIZE, int R = kFrRegs>
__device__ __forceinline__ void fr_sortd_sizes(double (&k)[R]) {
  fr_sortd_stage<N, SIZE, SIZE / 2, R>(k);
  if constexpr (SIZE < N) fr_sortd_sizes<N, SIZE * 2, R>(k);
}

// Sixteen keys by the 60-comparator network of ten layers (the bitonic network above takes 80; checked over all
// 2^16 zero-one inputs by tools/sort_net_check.py).  ce(lo, hi) leaves the smaller key in lo.
template <class T, class CE>
__device__ __forceinline__ void fr_sort16_net(T (&k)[kFrRegs], CE ce) {
  static_assert(kFrRegs == 16, "the network is for sixteen keys");
  ce(k[0], k[13]); ce(k[1], k[12]); ce(k[2], k[15]); ce(k[3], k[14]); ce(k[4], k[8]); ce(k[5], k[6]); ce(k[7], k[11]); ce(k[9], k[10]);
  ce(k[0], k[5]); ce(k[1], k[7]); ce(k[2], k[9]); ce(k[3], k[4]); ce(k[6], k[13]); ce(k[8], k[14]); ce(k[10], k[15]); ce(k[11], k[12]);
  ce(k[0], k[1]); ce(k[2], k[3]); ce(k[4], k[5]); ce(k[6], k[8]); ce(k[7], k[9]); ce(k[10], k[11]); ce(k[12], k[13]); ce(k[14], k[15]);
  ce(k[0], k[2]); ce(k[1], k[3]); ce(k[4], k[10]); ce(k[5], k[11]); ce(k[6], k[7]); ce(k[8], k[9]); ce(k[12], k[14]); ce(k[13], k[15]);
  ce(k[1], k[2]); ce(k[3], k[12]); ce(k[4], k[6]); ce(k[5], k[7]); ce(k[8], k[10]); ce(k[9], k[11]); ce(k[13], k[14]);
  ce(k[1], k[4]); ce(k[2], k[6]); ce(k[5], k[8]); ce(k[7], k[10]); ce(k[9], k[13]); ce(k[11], k[14]);
  ce(k[2], k[4]); ce(k[3], k[6]); ce(k[9], k[12]); ce(k[11], k[13]);
  ce(k[3], k[5]); ce(k[6], k[8]); ce(k[7], k[9]); ce(k[10], k[12]);
  ce(k[3], k[4]); ce(k[5], k[6]); ce(k[7], k[8]); ce(k[9], k[10]); ce(k[11], k[12]);
  ce(k[6], k[7]); ce(k[8], k[9]);
}
// The same for eight keys (Batcher's odd-even merge: 19 comparators) and for twelve (the network above without the
// comparators that touch the lines 12 ... 15 -- those lines hold padding, which no comparator moves down -- and without
// those that then compare nothing that can be out of order: 40).  Both checked like the one above.
template <class T, class CE>
__device__ __forceinline__ void fr_sort8_net(T (&k)[8], CE ce) {
  ce(k[0], k[1]); ce(k[2], k[3]); ce(k[4], k[5]); ce(k[6], k[7]);
  ce(k[0], k[2]); ce(k[1], k[3]); ce(k[4], k[6]); ce(k[5], k[7]);
  ce(k[1], k[2]); ce(k[5], k[6]);
  ce(k[0], k[4]); ce(k[1], k[5]); ce(k[2], k[6]); ce(k[3], k[7]);
  ce(k[2], k[4]); ce(k[3], k[5]);
  ce(k[1], k[2]); ce(k[3], k[4]); ce(k[5], k[6]);
}
template <class T, class CE>
__device__ __forceinline__ void fr_sort12_net(T (&k)[12], CE ce) {
  ce(k[4], k[8]); ce(k[5], k[6]); ce(k[7], k[11]); ce(k[9], k[10]);
  ce(k[0], k[5]); ce(k[1], k[7]); ce(k[2], k[9]); ce(k[3], k[4]); ce(k[6], k[8]); ce(k[10], k[11]);
  ce(k[0], k[1]); ce(k[2], k[3]); ce(k[4], k[5]); ce(k[7], k[9]);
  ce(k[0], k[2]); ce(k[1], k[3]); ce(k[4], k[10]); ce(k[5], k[11]); ce(k[6], k[7]); ce(k[8], k[9]);
  ce(k[1], k[2]); ce(k[4], k[6]); ce(k[5], k[7]); ce(k[8], k[10]); ce(k[9], k[11]);
  ce(k[1], k[4]); ce(k[2], k[6]); ce(k[5], k[8]); ce(k[7], k[10]);
  ce(k[2], k[4]); ce(k[3], k[6]); ce(k[7], k[9]);
  ce(k[3], k[5]); ce(k[6], k[8]); ce(k[9], k[10]);
  ce(k[3], k[4]); ce(k[5], k[6]); ce(k[7], k[8]);
  ce(k[6], k[7]); ce(k[8], k[9]);
}
// Twenty-four and thirty-two keys (the dense pass): Batcher's merge exchange (Knuth 5.2.2 M, which sorts any number of
// keys): 127 and 191 comparators, where the bitonic network takes 240 for thirty-two and has no form for twenty-four.
// (tools/sort_net_check.py: the first over all 2^24 zero-one inputs, the second over a sample of them.)
template <class T, class CE>
__device__ __forceinline__ void fr_sort24_net(T (&k)[24], CE ce) {
  ce(k[0], k[16]); ce(k[1], k[17]); ce(k[2], k[18]); ce(k[3], k[19]); ce(k[4], k[20]); ce(k[5], k[21]); ce(k[6], k[22]); ce(k[7], k[23]);
  ce(k[0], k[8]); ce(k[1], k[9]); ce(k[2], k[10]); ce(k[3], k[11]); ce(k[4], k[12]); ce(k[5], k[13]); ce(k[6], k[14]); ce(k[7], k[15]);
  ce(k[8], k[16]); ce(k[9], k[17]); ce(k[10], k[18]); ce(k[11], k[19]); ce(k[12], k[20]); ce(k[13], k[21]); ce(k[14], k[22]); ce(k[15], k[23]); ce(k[0], k[4]); ce(k[1], k[5]); ce(k[2], k[6]); ce(k[3], k[7]);
  ce(k[8], k[12]); ce(k[9], k[13]); ce(k[10], k[14]); ce(k[11], k[15]); ce(k[16], k[20]); ce(k[17], k[21]); ce(k[18], k[22]); ce(k[19], k[23]); ce(k[0], k[2]); ce(k[1], k[3]);
  ce(k[4], k[16]); ce(k[5], k[17]); ce(k[6], k[18]); ce(k[7], k[19]); ce(k[20], k[22]); ce(k[21], k[23]); ce(k[0], k[1]);
  ce(k[4], k[8]); ce(k[5], k[9]); ce(k[6], k[10]); ce(k[7], k[11]); ce(k[12], k[16]); ce(k[13], k[17]); ce(k[14], k[18]); ce(k[15], k[19]); ce(k[22], k[23]);
  ce(k[4], k[6]); ce(k[5], k[7]); ce(k[8], k[10]); ce(k[9], k[11]); ce(k[12], k[14]); ce(k[13], k[15]); ce(k[16], k[18]); ce(k[17], k[19]);
  ce(k[2], k[16]); ce(k[3], k[17]); ce(k[6], k[20]); ce(k[7], k[21]);
  ce(k[2], k[8]); ce(k[3], k[9]); ce(k[6], k[12]); ce(k[7], k[13]); ce(k[10], k[16]); ce(k[11], k[17]); ce(k[14], k[20]); ce(k[15], k[21]);
  ce(k[2], k[4]); ce(k[3], k[5]); ce(k[6], k[8]); ce(k[7], k[9]); ce(k[10], k[12]); ce(k[11], k[13]); ce(k[14], k[16]); ce(k[15], k[17]); ce(k[18], k[20]); ce(k[19], k[21]);
  ce(k[2], k[3]); ce(k[4], k[5]); ce(k[6], k[7]); ce(k[8], k[9]); ce(k[10], k[11]); ce(k[12], k[13]); ce(k[14], k[15]); ce(k[16], k[17]); ce(k[18], k[19]); ce(k[20], k[21]);
  ce(k[1], k[16]); ce(k[3], k[18]); ce(k[5], k[20]); ce(k[7], k[22]);
  ce(k[1], k[8]); ce(k[3], k[10]); ce(k[5], k[12]); ce(k[7], k[14]); ce(k[9], k[16]); ce(k[11], k[18]); ce(k[13], k[20]); ce(k[15], k[22]);
  ce(k[1], k[4]); ce(k[3], k[6]); ce(k[5], k[8]); ce(k[7], k[10]); ce(k[9], k[12]); ce(k[11], k[14]); ce(k[13], k[16]); ce(k[15], k[18]); ce(k[17], k[20]); ce(k[19], k[22]);
  ce(k[1], k[2]); ce(k[3], k[4]); ce(k[5], k[6]); ce(k[7], k[8]); ce(k[9], k[10]); ce(k[11], k[12]); ce(k[13], k[14]); ce(k[15], k[16]); ce(k[17], k[18]); ce(k[19], k[20]); ce(k[21], k[22]);
}
template <class T, class CE>
__device__ __forceinline__ void fr_sort32_net(T (&k)[32], CE ce) {
  ce(k[0], k[16]); ce(k[1], k[17]); ce(k[2], k[18]); ce(k[3], k[19]); ce(k[4], k[20]); ce(k[5], k[21]); ce(k[6], k[22]); ce(k[7], k[23]); ce(k[8], k[24]); ce(k[9], k[25]); ce(k[10], k[26]); ce(k[11], k[27]); ce(k[12], k[28]); ce(k[13], k[29]); ce(k[14], k[30]); ce(k[15], k[31]);
  ce(k[0], k[8]); ce(k[1], k[9]); ce(k[2], k[10]); ce(k[3], k[11]); ce(k[4], k[12]); ce(k[5], k[13]); ce(k[6], k[14]); ce(k[7], k[15]); ce(k[16], k[24]); ce(k[17], k[25]); ce(k[18], k[26]); ce(k[19], k[27]); ce(k[20], k[28]); ce(k[21], k[29]); ce(k[22], k[30]); ce(k[23], k[31]);
  ce(k[8], k[16]); ce(k[9], k[17]); ce(k[10], k[18]); ce(k[11], k[19]); ce(k[12], k[20]); ce(k[13], k[21]); ce(k[14], k[22]); ce(k[15], k[23]); ce(k[0], k[4]); ce(k[1], k[5]); ce(k[2], k[6]); ce(k[3], k[7]); ce(k[24], k[28]); ce(k[25], k[29]); ce(k[26], k[30]); ce(k[27], k[31]);
  ce(k[8], k[12]); ce(k[9], k[13]); ce(k[10], k[14]); ce(k[11], k[15]); ce(k[16], k[20]); ce(k[17], k[21]); ce(k[18], k[22]); ce(k[19], k[23]); ce(k[0], k[2]); ce(k[1], k[3]); ce(k[28], k[30]); ce(k[29], k[31]);
  ce(k[4], k[16]); ce(k[5], k[17]); ce(k[6], k[18]); ce(k[7], k[19]); ce(k[12], k[24]); ce(k[13], k[25]); ce(k[14], k[26]); ce(k[15], k[27]); ce(k[0], k[1]); ce(k[30], k[31]);
  ce(k[4], k[8]); ce(k[5], k[9]); ce(k[6], k[10]); ce(k[7], k[11]); ce(k[12], k[16]); ce(k[13], k[17]); ce(k[14], k[18]); ce(k[15], k[19]); ce(k[20], k[24]); ce(k[21], k[25]); ce(k[22], k[26]); ce(k[23], k[27]);
  ce(k[4], k[6]); ce(k[5], k[7]); ce(k[8], k[10]); ce(k[9], k[11]); ce(k[12], k[14]); ce(k[13], k[15]); ce(k[16], k[18]); ce(k[17], k[19]); ce(k[20], k[22]); ce(k[21], k[23]); ce(k[24], k[26]); ce(k[25], k[27]);
  ce(k[2], k[16]); ce(k[3], k[17]); ce(k[6], k[20]); ce(k[7], k[21]); ce(k[10], k[24]); ce(k[11], k[25]); ce(k[14], k[28]); ce(k[15], k[29]);
  ce(k[2], k[8]); ce(k[3], k[9]); ce(k[6], k[12]); ce(k[7], k[13]); ce(k[10], k[16]); ce(k[11], k[17]); ce(k[14], k[20]); ce(k[15], k[21]); ce(k[18], k[24]); ce(k[19], k[25]); ce(k[22], k[28]); ce(k[23], k[29]);
  ce(k[2], k[4]); ce(k[3], k[5]); ce(k[6], k[8]); ce(k[7], k[9]); ce(k[10], k[12]); ce(k[11], k[13]); ce(k[14], k[16]); ce(k[15], k[17]); ce(k[18], k[20]); ce(k[19], k[21]); ce(k[22], k[24]); ce(k[23], k[25]); ce(k[26], k[28]); ce(k[27], k[29]);
  ce(k[2], k[3]); ce(k[4], k[5]); ce(k[6], k[7]); ce(k[8], k[9]); ce(k[10], k[11]); ce(k[12], k[13]); ce(k[14], k[15]); ce(k[16], k[17]); ce(k[18], k[19]); ce(k[20], k[21]); ce(k[22], k[23]); ce(k[24], k[25]); ce(k[26], k[27]); ce(k[28], k[29]);
  ce(k[1], k[16]); ce(k[3], k[18]); ce(k[5], k[20]); ce(k[7], k[22]); ce(k[9], k[24]); ce(k[11], k[26]); ce(k[13], k[28]); ce(k[15], k[30]);
  ce(k[1], k[8]); ce(k[3], k[10]); ce(k[5], k[12]); ce(k[7], k[14]); ce(k[9], k[16]); ce(k[11], k[18]); ce(k[13], k[20]); ce(k[15], k[22]); ce(k[17], k[24]); ce(k[19], k[26]); ce(k[21], k[28]); ce(k[23], k[30]);
  ce(k[1], k[4]); ce(k[3], k[6]); ce(k[5], k[8]); ce(k[7], k[10]); ce(k[9], k[12]); ce(k[11], k[14]); ce(k[13], k[16]); ce(k[15], k[18]); ce(k[17], k[20]); ce(k[19], k[22]); ce(k[21], k[24]); ce(k[23], k[26]); ce(k[25], k[28]); ce(k[27], k[30]);
  ce(k[1], k[2]); ce(k[3], k[4]); ce(k[5], k[6]); ce(k[7], k[8]); ce(k[9], k[10]); ce(k[11], k[12]); ce(k[13], k[14]); ce(k[15], k[16]); ce(k[17], k[18]); ce(k[19], k[20]); ce(k[21], k[22]); ce(k[23], k[24]); ce(k[25], k[26]); ce(k[27], k[28]); ce(k[29], k[30]);
}
struct FrCeU32 {
  __device__ __forceinline__ void operator()(uint32_t& lo, uint32_t& hi) const {
    const uint32_t a = min(lo, hi), b = max(lo, hi);
    lo = a, hi = b;
  }
};
struct FrCeF64 {  // (positive finite doubles as bit patterns: see fr_sortd_stage)
  __device__ __forceinline__ void operator()(double& lo, double& hi) const {
    double a, b;
    asm("v_min_f64 %0, %1, %2" : "=v"(a) : "v"(lo), "v"(hi));
    asm("v_max_f64 %0, %1, %2" : "=v"(b) : "v"(lo), "v"(hi));
    lo = a, hi = b;
  }
};

// Sorts a bucket of w entries by (label, arc), folds the runs of one label in arc order, sorts the
// folded keys (desc(float logp) | label); returns their number.  E(j) is the j-th entry.
template <class Entry>
__device__ __forceinline__ int fr_fold_bucket(int w, Entry E) {
  for (int i = 1; i < w; ++i) {
    const FKey v = E(i);
    int t = i - 1;
    while (t >= 0 && E(t).a > v.a) {
      E(t + 1) = E(t);
      --t;
    }
    E(t + 1) = v;
  }
  int m = 0;
  for (int j = 0; j < w;) {
    const FKey e = E(j);
    const uint32_t lab = (uint32_t)(e.a >> 32);
    double acc = __longlong_as_double((long long)e.b);
    for (++j; j < w && (uint32_t)(E(j).a >> 32) == lab; ++j) acc = fr_log_add(acc, __longlong_as_double((long long)E(j).b));
    const float lp = (float)acc;  // BaseFloat narrowing of :126
    FKey out;
    out.a = ((uint64_t)fr_desc_key(lp) << 32) | lab;
    out.b = 0;
    E(m++) = out;  // m <= the entries already read
  }
  for (int i = 1; i < m; ++i) {
    const FKey v = E(i);
    int t = i - 1;
    while (t >= 0 && E(t).a > v.a) {
      E(t + 1) = E(t);
      --t;
    }
    E(t + 1) = v;
  }
  return m;
}

__global__ __launch_bounds__(kFrThreads) void frame_kernel(FrameArgs a, int span_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fr_smem[];
  FKey* slab = reinterpret_cast<FKey*>(fr_smem);  // [kFrTile]: the cooperative sort's tile
  uint16_t* lbucket = reinterpret_cast<uint16_t*>(fr_smem);  // [kFrLdsEntries]: a tile's buckets as arc numbers (same memory)
  // [span_cap]: k0 | k1 << 16 of every arc -- the two passes over all arcs that every tile makes
  // (counts, scatter) read 4 bytes of LDS instead of a 16-byte row of L2 behind a dependent loop
  uint32_t* lspan = reinterpret_cast<uint32_t*>(fr_smem + sizeof(uint16_t) * kFrLdsEntries);
  __shared__ int tile_max, lab_min, lab_max, span_max, unsorted;
  // Only the frames at which some arc starts or ends (and the tile's first) differ from the frame
  // before them: those CHANGED frames get a bucket, the others repeat their predecessor's records.
  __shared__ uint16_t cidx[kFrTileFrames + 2];    // changed frames before frame k of the tile (cidx[tf] = their number)
  __shared__ uint16_t cframe[kFrTileFrames + 2];  // the c-th changed frame (cframe[nc] = tf)
  __shared__ int lboff[kFrTileFrames + 1];  // first expansion of every changed frame of the tile (tile-local)
  __shared__ int lcur[kFrTileFrames];       // scatter cursors; then -(folded entries) - 1 for a bucket the workgroup folded
  __shared__ int sh[2 * kFrThreads / 64];
  __shared__ int big_list[kFrBigList];
  __shared__ int big_n;
  // alone: every lattice.  Behind the lean kernel: the lattices it listed (the other workgroups leave at once)
  if (a.redo_list && (int)blockIdx.x >= *a.redo_n) return;
  const int u = a.redo_list ? a.redo_list[blockIdx.x] : (int)blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t a0 = a.arc_off[u];
  const int nA = (int)(a.arc_off[u + 1] - a0);
  const int64_t s0 = a.state_off[u];
  const int64_t pb = padded_state_base(s0, u);
  const double tot = a.total[u];
  const int T = a.num_frames[u];
  FKey* keys = reinterpret_cast<FKey*>(a.key_scratch) + a.scratch_off[u];
  FKey* recs = reinterpret_cast<FKey*>(a.rec_scratch) + a.scratch_off[u];
  uint64_t* rec8 = reinterpret_cast<uint64_t*>(recs);  // the records, 8 bytes each, from the front of the lattice's rows
  const int64_t fbase = a.frame_base[u];
  FrameArc* ai = a.arc_info + a0;
  const ScaleParams sp = a.sp;
  // the last frame_sort_slots(nA) record rows of the lattice are the cooperative sort's scratch
  FKey* tmp = recs + ((a.scratch_off[u + 1] - a.scratch_off[u]) - frame_sort_slots(nA));

  if (nA == 0) {  // nothing crosses any frame (and the unconditional gathers of the frames phase read the lattice's first arc)
    for (int f = tid; f < T; f += kFrThreads) a.frame_ref[fbase + f] = make_int2(0, 0);
    if (tid == 0) a.num_records[u] = 0, a.num_stored[u] = 0, a.rec_ptr[u] = rec8, a.rec_fmt[u] = 0;
    return;
  }
  const bool spans_lds = nA <= span_cap && T <= 65535;  // workgroup-uniform
  long long clk[6] = {0, 0, 0, 0, 0, 0};  // spans, counts, scatter, crowded, frames, (unused)
  long long t_prev = (long long)__builtin_amdgcn_s_memtime();
#define FR_LAP(k)                                                      \
  do {                                                                 \
    if (a.phase_clock) {                                               \
      const long long t_now = (long long)__builtin_amdgcn_s_memtime(); \
      clk[k] += t_now - t_prev;                                        \
      t_prev = t_now;                                                  \
    }                                                                  \
  } while (0)
  // ---- 0. clipped span and x of every arc; the label range of the lattice ----
  if (tid == 0) lab_min = 0x7FFFFFFF, lab_max = (int)0x80000000, span_max = 0, unsorted = 0;
  __syncthreads();
  int lmin_t = 0x7FFFFFFF, lmax_t = (int)0x80000000, span_max_t = 0;
  for (int base = 0; base < nA; base += kFrSpanBatch * kFrThreads) {  // eight arcs per thread at a time: loads batched, no branch around them
    int il[kFrSpanBatch], t0[kFrSpanBatch], t1[kFrSpanBatch];
    float gr[kFrSpanBatch], ac4[kFrSpanBatch];
    double lp[kFrSpanBatch];
#pragma unroll
    for (int q = 0; q < kFrSpanBatch; ++q) {
      const int i = min(base + q * kFrThreads + tid, nA - 1);
      il[q] = a.ilabel[a0 + i];
      t0[q] = a.src[a0 + i];
      t1[q] = a.dst[a0 + i];
      gr[q] = a.graph[a0 + i];
      ac4[q] = a.acoustic[a0 + i];
    }
#pragma unroll
    for (int q = 0; q < kFrSpanBatch; ++q) {
      // fw[u] + bw[nextstate] - cost - total (latbin/lattice-to-word-frame-post.cc:102-104,122), formed
      // here from alpha / beta: no posterior pass has to run before this kernel
      const double through = a.alpha[pb + t0[q]] + a.beta[pb + t1[q]] - arc_cost(gr[q], ac4[q], il[q] != 0, sp);
      lp[q] = (through > -INFINITY) ? through - tot : -INFINITY;
      t0[q] = a.state_time[s0 + t0[q]];
      t1[q] = a.state_time[s0 + t1[q]];
    }
#pragma unroll
    for (int q = 0; q < kFrSpanBatch; ++q) {
      const int i = base + q * kFrThreads + tid;
      if (i >= nA) continue;
      FrameArc r;
      r.k0 = r.k1 = 0;
      r.x = 0.0;
      const int lo = max(t0[q], 0), hi = min(t1[q], T);
      if (il[q] != 0 && hi > lo) {
        r.k0 = lo;
        r.k1 = hi;
        lmin_t = min(lmin_t, il[q]), lmax_t = max(lmax_t, il[q]);
        // lp holds alpha - (double)g - (double)a + beta - total; the reference forms the arc
        // cost here as a float sum, so swap the double sum for the float one
        float g, ac;
        arc_cost_parts(gr[q], ac4[q], true, sp, g, ac);
        // (an arc of infinite cost is log-zero either way; the correction would be inf - inf)
        r.x = (lp[q] > -INFINITY) ? lp[q] + ((double)g + (double)ac) - (double)(g + ac) : lp[q];
      }
      ai[i] = r;
      if (spans_lds) {
        // (an empty span is kept as [t, t) at its source's time: in a time-sorted lattice the first
        // frames of the arcs then ascend with the arc number, which the tiles below make use of)
        const int t = min(max(t0[q], 0), 65535);
        lspan[i] = r.k1 > r.k0 ? ((uint32_t)r.k0 | ((uint32_t)r.k1 << 16)) : ((uint32_t)t | ((uint32_t)t << 16));
        span_max_t = max(span_max_t, r.k1 - r.k0);
      }
    }
  }
  for (int d = 32; d > 0; d >>= 1) lmin_t = min(lmin_t, __shfl_xor(lmin_t, d, 64)), lmax_t = max(lmax_t, __shfl_xor(lmax_t, d, 64));
  for (int d = 32; d > 0; d >>= 1) span_max_t = max(span_max_t, __shfl_xor(span_max_t, d, 64));
  if ((tid & 63) == 0) atomicMin(&lab_min, lmin_t), atomicMax(&lab_max, lmax_t), atomicMax(&span_max, span_max_t);
  __syncthreads();
  // do the arcs' first frames ascend with the arc number?  Then a tile's arcs are one range of arcs
  // (two binary searches over the spans in LDS) and its passes need not visit the others.
  if (spans_lds) {
    bool desc = false;
    for (int i = tid + 1; i < nA; i += kFrThreads) desc |= (lspan[i] & 0xFFFFu) < (lspan[i - 1] & 0xFFFFu);
    if (__any(desc) && (tid & 63) == 0) atomicOr(&unsorted, 1);
  }
  __syncthreads();
  const bool ranged = spans_lds && unsorted == 0;  // workgroup-uniform
  const int longest = span_max;
  // (label - lab_min) and the arc number fit one 32-bit sort key?  (0xFFFFFFFF stays free for padding)
  int abits = 1;
  while ((1 << abits) < nA) ++abits;
  const bool key32 = nA <= 65535 && lab_max >= lab_min &&
                     (uint64_t)((int64_t)lab_max - (int64_t)lab_min) < ((1ull << (32 - abits)) - 1ull);
  const int lab_base = lab_min;
  const bool dsort = key32 && (int64_t)lab_max - (int64_t)lab_min < ((int64_t)1 << 30);  // folded keys sortable as doubles

  FR_LAP(0);
  int nbase = 0;  // expansions of the tiles before this one
  int mbase = 0;  // records written so far
  int dense = 0;  // (frame, word) entries so far: every frame counts its records, shared or not
  for (int f0 = 0; f0 < T; f0 += kFrTileFrames) {
    const int tf = min(kFrTileFrames, T - f0);
    // ---- 1. entries per frame from a difference array over the spans, the changed frames of the
    // tile, and the bucket offsets of the changed frames: one pass over the arcs, three scans ----
    constexpr int kCh = (kFrTileFrames + 1 + kFrThreads - 1) / kFrThreads;
    for (int k = tid; k <= tf; k += kFrThreads) lboff[k] = 0, cidx[k] = (k == 0) ? 1 : 0;
    if (tid == 0) big_n = 0;
    __syncthreads();
    // arcs that can cover a frame of the tile: first frame below the tile's end, and (no arc is
    // longer than `longest`) above its start less that
    int arc_lo = 0, arc_hi = nA;
    if (ranged) {
      auto first_at_least = [&](int frame) {  // first arc whose first frame is >= frame
        int l = 0, h = nA;
        while (l < h) {
          const int mid = (l + h) >> 1;
          if ((int)(lspan[mid] & 0xFFFFu) < frame) l = mid + 1;
          else h = mid;
        }
        return l;
      };
      arc_lo = first_at_least(f0 - longest + 1);
      arc_hi = first_at_least(f0 + tf);
    }
    auto span_of = [&](int i, int& lo, int& hi) {
      const uint32_t sp32 = spans_lds ? lspan[i] : 0u;
      const int k0 = spans_lds ? (int)(sp32 & 0xFFFFu) : ai[i].k0, k1 = spans_lds ? (int)(sp32 >> 16) : ai[i].k1;
      lo = max(k0, f0) - f0, hi = min(k1, f0 + tf) - f0;
    };
    for (int i = arc_lo + tid; i < arc_hi; i += kFrThreads) {
      int lo, hi;
      span_of(i, lo, hi);
      if (hi > lo) {
        atomicAdd(&lboff[lo], 1);
        atomicAdd(&lboff[hi], -1);
        cidx[lo] = 1;  // (plain stores of the same value)
        if (hi < tf) cidx[hi] = 1;
      }
    }
    __syncthreads();
    // every thread owns kCh consecutive frames
    int n = 0, cmax = 0, nc = 0;
    {
      int dl[kCh], fl[kCh], sum_d = 0, sum_f = 0;
#pragma unroll
      for (int q = 0; q < kCh; ++q) {
        const int k = tid * kCh + q;
        dl[q] = (k <= tf) ? lboff[k] : 0;
        fl[q] = (k < tf) ? (int)cidx[k] : 0;
        sum_d += dl[q];
        sum_f += fl[q];
      }
      int tot_d;
      const int2 pre = fr_block_scan2(sum_d, sum_f, &tot_d, &nc, sh);  // (all reads of the two arrays are done)
      int run = pre.x, sum_c = 0;
#pragma unroll
      for (int q = 0; q < kCh; ++q) {
        run += dl[q];
        dl[q] = fl[q] ? run : 0;  // entries of a changed frame
        cmax = max(cmax, dl[q]);
        sum_c += dl[q];
      }
      int off = fr_block_scan(sum_c, &n, sh);
      int c = pre.y;
#pragma unroll
      for (int q = 0; q < kCh; ++q) {
        const int k = tid * kCh + q;
        if (k <= tf) cidx[k] = (uint16_t)c;
        if (fl[q]) {
          cframe[c] = (uint16_t)k;
          lboff[c] = off;
          lcur[c] = off;
          off += dl[q];
          ++c;
        }
      }
      if (tid == 0) cframe[nc] = (uint16_t)tf, lboff[nc] = n;
    }
    if (tid == 0) tile_max = 0;
    __syncthreads();
    for (int d = 32; d > 0; d >>= 1) cmax = max(cmax, __shfl_xor(cmax, d, 64));
    if ((tid & 63) == 0) atomicMax(&tile_max, cmax);
    __syncthreads();
    const bool in_lds = nA <= 65535 && n <= kFrLdsEntries && tile_max <= kFrSerialMax;  // workgroup-uniform

    FR_LAP(1);
    // ---- 2. every arc drops itself into the buckets of its frames ----
    if (in_lds) {
      for (int i = arc_lo + tid; i < arc_hi; i += kFrThreads) {
        int lo, hi;
        span_of(i, lo, hi);
        if (hi <= lo) continue;
        const int c1 = cidx[hi];
        for (int c = cidx[lo]; c < c1; c += 4) {  // four cursors at a time: the atomics' round trips overlap
          int at[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) at[q] = (c + q < c1) ? atomicAdd(&lcur[c + q], 1) : -1;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (at[q] >= 0) lbucket[at[q]] = (uint16_t)i;
        }
      }
    } else {
      for (int i = tid; i < nA; i += kFrThreads) {
        int lo, hi;
        span_of(i, lo, hi);
        if (hi <= lo) continue;
        FKey e;
        e.a = ((uint64_t)fr_bias(a.ilabel[a0 + i]) << 32) | (uint32_t)i;
        e.b = (uint64_t)__double_as_longlong(ai[i].x);
        const int c1 = cidx[hi];
        for (int c = cidx[lo]; c < c1; ++c) keys[nbase + atomicAdd(&lcur[c], 1)] = e;
      }
    }
    __syncthreads();

    FR_LAP(2);
    // ---- 3a. buckets above kFrSerialMax entries, kFrBigList at a time: sorted through the LDS
    // tile (which overlays the slabs), folded by the heads of the runs, sorted again; the folded
    // keys stay at the bucket's start ----
    for (; !in_lds;) {  // (a tile whose buckets sit in LDS has no such frame)
      for (int k = tid; k < nc; k += kFrThreads)
        if (lboff[k + 1] - lboff[k] > kFrSerialMax && lcur[k] >= 0) {
          const int slot = atomicAdd(&big_n, 1);
          if (slot < kFrBigList) big_list[slot] = k;
        }
      __syncthreads();
      const int nbig = min(big_n, kFrBigList);
      if (nbig == 0) break;
      for (int bi = 0; bi < nbig; ++bi) {
        const int k = big_list[bi];
        const int o = nbase + lboff[k], w = lboff[k + 1] - lboff[k];
        int np = 1;
        while (np < w) np <<= 1;
        for (int i = tid; i < np; i += kFrThreads) {
          FKey e;
          e.a = e.b = ~0ull;
          if (i < w) e = keys[o + i];
          tmp[i] = e;
        }
        __syncthreads();
        fr_bitonic_sort(tmp, np, slab, FKeyLess());
        int m = 0;
        for (int base = 0; base < w; base += kFrThreads) {
          const int i = base + tid;
          bool head = false;
          FKey out;
          out.a = out.b = 0;
          if (i < w) {
            const uint32_t lab = (uint32_t)(tmp[i].a >> 32);
            head = (i == 0) || ((uint32_t)(tmp[i - 1].a >> 32) != lab);
            if (head) {
              double acc = __longlong_as_double((long long)tmp[i].b);
              for (int j = i + 1; j < w && (uint32_t)(tmp[j].a >> 32) == lab; ++j)
                acc = fr_log_add(acc, __longlong_as_double((long long)tmp[j].b));
              out.a = ((uint64_t)fr_desc_key((float)acc) << 32) | lab;
            }
          }
          int tot;
          const int off = fr_block_scan(head ? 1 : 0, &tot, sh);
          if (head) keys[o + m + off] = out;
          m += tot;
        }
        __syncthreads();
        int mp = 1;
        while (mp < m) mp <<= 1;
        for (int i = tid; i < mp; i += kFrThreads) {
          FKey e;
          e.a = e.b = ~0ull;
          if (i < m) e = keys[o + i];
          tmp[i] = e;
        }
        __syncthreads();
        fr_bitonic_sort(tmp, mp, slab, FKeyLess());
        for (int i = tid; i < m; i += kFrThreads) keys[o + i] = tmp[i];
        if (tid == 0) lcur[k] = -m - 1;
        __syncthreads();
      }
      if (tid == 0) big_n = 0;
      __syncthreads();
    }

    FR_LAP(3);
    // ---- 3b. one thread per changed frame, kFrThreads at a time: sort, fold, sort, place (once
    // for the frame and once more for every unchanged frame that follows it) ----
    for (int r0 = 0; r0 < nc; r0 += kFrThreads) {
      const int k = r0 + tid;
      int o = 0, w = 0, mk = 0;
      bool in_regs = false;
      if (k < nc) {
        o = nbase + lboff[k];
        w = lboff[k + 1] - lboff[k];
        if (lcur[k] < 0) {
          mk = -lcur[k] - 1;  // folded by the workgroup
        } else if (w > kFrRegs) {
          if (in_lds)  // a mid-sized bucket: its entries take their scratch rows after all
            for (int j = 0; j < w; ++j) {
              const int i = (int)lbucket[lboff[k] + j];
              FKey e;
              e.a = ((uint64_t)fr_bias(a.ilabel[a0 + i]) << 32) | (uint32_t)i;
              e.b = (uint64_t)__double_as_longlong(ai[i].x);
              keys[o + j] = e;
            }
          mk = fr_fold_bucket(w, [&](int j) -> FKey& { return keys[o + j]; });
        } else {
          in_regs = w > 0;
        }
      }
      uint64_t ek[kFrRegs];
      double ex[kFrRegs];
      bool narrow;
      if (in_lds && key32) {
        // sort keys (label - lab_base | arc number) in 32 bits: a compare-exchange is two instructions
        // and no payload moves; x is fetched by arc number after the sort
        // (every load here is unconditional -- a load inside a branch waits for its data before the branch closes, and
        // sixteen gathers would take sixteen trips to L2 one after the other; the slots beyond the bucket read the
        // lattice's first arc.  nA > 0: the tile has a bucket entry)
        uint32_t k32[kFrRegs];
        const int lo = (k < nc) ? lboff[k] : 0;
        const int wr = in_regs ? w : 0;
#pragma unroll
        for (int j = 0; j < kFrRegs; ++j) k32[j] = (uint32_t)lbucket[min(lo + j, kFrLdsEntries - 1)];
#pragma unroll
        for (int j = 0; j < kFrRegs; ++j) asm volatile("" : "+v"(k32[j]));  // (the reads stay outside the selects)
#pragma unroll
        for (int j = 0; j < kFrRegs; ++j) k32[j] = (j < wr) ? k32[j] : 0u;
        {
          int32_t lab[kFrRegs];
#pragma unroll
          for (int j = 0; j < kFrRegs; ++j) lab[j] = a.ilabel[a0 + (int)k32[j]];  // gathers served by L2
#pragma unroll
          for (int j = 0; j < kFrRegs; ++j) k32[j] = (j < wr) ? (k32[j] | ((uint32_t)(lab[j] - lab_base) << abits)) : 0xFFFFFFFFu;
        }
        narrow = __all(!in_regs || w <= 8);  // wave-uniform: the 8-wide network is enough
        if (narrow)
          fr_sort32_sizes<8, 2>(k32);
        else
          fr_sort32_sizes<kFrRegs, 2>(k32);
        const uint32_t amask = (1u << abits) - 1u;
#pragma unroll
        for (int j = 0; j < kFrRegs; ++j) ex[j] = ai[(k32[j] != 0xFFFFFFFFu) ? (int)(k32[j] & amask) : 0].x;
#pragma unroll
        for (int j = 0; j < kFrRegs; ++j) {
          const int arc = (int)(k32[j] & amask);
          const bool real = k32[j] != 0xFFFFFFFFu;
          ek[j] = real ? (((uint64_t)fr_bias((int32_t)(k32[j] >> abits) + lab_base) << 32) | (uint32_t)arc) : ~0ull;
          ex[j] = real ? ex[j] : 0.0;
        }
      } else {
        if (in_lds) {
          int arc[kFrRegs];
          const int lo = (k < nc) ? lboff[k] : 0;
#pragma unroll
          for (int j = 0; j < kFrRegs; ++j) arc[j] = (in_regs && j < w) ? (int)lbucket[lo + j] : -1;
#pragma unroll
          for (int j = 0; j < kFrRegs; ++j) {  // (label, x) of the arcs: gathers served by L2, all in flight together
            ek[j] = ~0ull;
            ex[j] = 0.0;
            if (arc[j] >= 0) {
              ek[j] = ((uint64_t)fr_bias(a.ilabel[a0 + arc[j]]) << 32) | (uint32_t)arc[j];
              ex[j] = ai[arc[j]].x;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < kFrRegs; ++j) {
            ek[j] = ~0ull;
            ex[j] = 0.0;
            if (in_regs && j < w) {
              const FKey e = keys[o + j];
              ek[j] = e.a;
              ex[j] = __longlong_as_double((long long)e.b);
            }
          }
        }
        narrow = __all(!in_regs || w <= 8);  // wave-uniform: the 8-wide network is enough
        if (narrow)
          fr_sort_regs<8, true>(ek, ex);
        else
          fr_sort_regs<kFrRegs, true>(ek, ex);
      }
      // fold: the key of a run is emitted at its last entry, the other entries turn into padding
      double acc = ex[0];
#pragma unroll
      for (int j = 1; j <= kFrRegs; ++j) {
        const uint32_t lab = (uint32_t)(ek[j - 1] >> 32);
        const bool cont = j < kFrRegs && j < w && (uint32_t)(ek[j < kFrRegs ? j : 0] >> 32) == lab;
        const bool emit = in_regs && !cont && j - 1 < w;
        ek[j - 1] = emit ? (((uint64_t)fr_desc_key((float)acc) << 32) | lab) : ~0ull;  // BaseFloat narrowing of :126
        mk += emit ? 1 : 0;
        if (j < kFrRegs) {
          // (LogAdd is ~200 instructions of double exp / log1p: skipped unless some frame of the wave
          // has two arcs of one word here)
          double folded = ex[j];
          if (__any(in_regs && cont)) folded = fr_log_add(acc, ex[j]);
          acc = cont ? folded : ex[j];
        }
      }
      if (dsort) {
        // (the label range fits 30 bits here: desc(float logp) << 30 | label - lab_base is a positive
        // finite double, sorted by min / max; the records keep their format)
        double dk[kFrRegs];
#pragma unroll
        for (int j = 0; j < kFrRegs; ++j) {
          const uint64_t key = ((ek[j] >> 32) << 30) | (uint64_t)(uint32_t)(fr_unbias((uint32_t)ek[j]) - lab_base);
          dk[j] = __longlong_as_double((long long)(ek[j] == ~0ull ? 0x7FEFFFFFFFFFFFFFull : key));
        }
        if (narrow)
          fr_sortd_sizes<8, 2>(dk);
        else
          fr_sortd_sizes<kFrRegs, 2>(dk);
#pragma unroll
        for (int j = 0; j < kFrRegs; ++j) {
          const uint64_t key = (uint64_t)__double_as_longlong(dk[j]);
          ek[j] = ((key >> 30) << 32) | fr_bias((int32_t)(uint32_t)(key & 0x3FFFFFFFull) + lab_base);
        }
      } else if (narrow) {
        fr_sort_regs<8, false>(ek, ex);
      } else {
        fr_sort_regs<kFrRegs, false>(ek, ex);
      }
      const int frame = (k < nc) ? (int)cframe[k] : 0, run = (k < nc) ? (int)cframe[k + 1] - frame : 0;
      int tot, tot_dense;
      const int dst = mbase + fr_block_scan2(mk, mk * run, &tot, &tot_dense, sh).x;
      // 8 bytes per record -- desc(float logp) | biased label, the sort key itself -- written ONCE per
      // changed frame; every frame names its records by (first record, count), so the unchanged frames
      // that follow share them, and the gather kernel expands the runs into the dense output
      for (int r = 0; r < run; ++r) {
        a.frame_ref[fbase + f0 + frame + r] = make_int2(mk, dst);
      }
      if (in_regs) {
#pragma unroll
        for (int j = 0; j < kFrRegs; ++j)
          if (j < mk) rec8[dst + j] = ek[j];
      } else {
        for (int j = 0; j < mk; ++j) rec8[dst + j] = keys[o + j].a;
      }
      dense += tot_dense;
      mbase += tot;
    }
    nbase += n;
    __syncthreads();
    FR_LAP(4);
  }
  if (tid == 0) a.num_records[u] = dense, a.num_stored[u] = mbase, a.rec_ptr[u] = rec8, a.rec_fmt[u] = 0;
  if (a.phase_clock && tid == 0)
    for (int k = 0; k < 5; ++k) a.phase_clock[(size_t)u * 8 + k] = clk[k];
#undef FR_LAP
}

// A rank of the lean kernel with more than kFrRegs (and at most 64) entries, by one wave: one lane per entry, both
// sorts as bitonic networks across the lanes, the fold entry by entry in arc order; returns the records written.
// (Rare, and kept out of line: inlined, its registers would be the lean kernel's.)
// The value of lane (l ^ STRIDE), without a trip through the LDS crossbar (ds_bpermute: ~100 cycles each, and a bitonic
// network over a wave is 21 of them one after the other): quad permutes for 1 and 2, a pair of row shifts under
// complementary bank masks for 4 and 8 (a bank is four lanes: the banks whose lanes have the stride's bit clear read
// upwards, the others downwards), and gfx950's v_permlane16_swap / v_permlane32_swap for 16 and 32 (both operands the
// value itself: the two results hold every row's / half's neighbour).
template <int STRIDE>
__device__ __forceinline__ uint32_t fr_lane_xor(uint32_t v, int lane) {
  static_assert(STRIDE == 1 || STRIDE == 2 || STRIDE == 4 || STRIDE == 8 || STRIDE == 16 || STRIDE == 32, "a power of two below the wave");
  if constexpr (STRIDE == 1) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm:[1,0,3,2]
  } else if constexpr (STRIDE == 2) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm:[2,3,0,1]
  } else if constexpr (STRIDE == 4) {
    const int up = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);   // row_shl:4 into banks 0, 2
    return (uint32_t)__builtin_amdgcn_update_dpp(up, (int)v, 0x114, 0xF, 0xA, false);  // row_shr:4 into banks 1, 3
  } else if constexpr (STRIDE == 8) {
    const int up = __builtin_amdgcn_update_dpp(0, (int)v, 0x108, 0xF, 0x3, false);   // row_shl:8 into banks 0, 1
    return (uint32_t)__builtin_amdgcn_update_dpp(up, (int)v, 0x118, 0xF, 0xC, false);  // row_shr:8 into banks 2, 3
  } else if constexpr (STRIDE == 16) {
    const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (lane & 16) ? p[0] : p[1];
  } else {
    const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? p[0] : p[1];
  }
}
template <int STRIDE>
__device__ __forceinline__ double fr_lane_xor(double v, int lane) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = fr_lane_xor<STRIDE>((uint32_t)b, lane), hi = fr_lane_xor<STRIDE>((uint32_t)(b >> 32), lane);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// Bitonic network over 64 * NR keys, NR a lane (key[r] of lane l is element r * 64 + l), ascending; pick(a, b, take_min)
// is the comparator.  Strides below 64 pair lanes of one register, those of 64 and more a lane's own registers.
template <int NR, int SIZE, int STRIDE, class T, class Pick>
__device__ __forceinline__ void fr_wave_sort_stage(T (&key)[NR], int lane, Pick pick) {
  if constexpr (STRIDE >= 64) {  // a lane's own registers r and r + STRIDE / 64
    constexpr int RS = STRIDE / 64;
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if ((r & RS) == 0) {
        const bool up = ((r << 6) & SIZE) == 0;
        const T lo = pick(key[r], key[r | RS], true), hi = pick(key[r], key[r | RS], false);
        key[r] = up ? lo : hi, key[r | RS] = up ? hi : lo;
      }
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const T other = fr_lane_xor<STRIDE>(key[r], lane);
      const bool up = ((lane | (r << 6)) & SIZE) == 0, lower = (lane & STRIDE) == 0;
      key[r] = pick(key[r], other, up == lower);  // the lower lane of an ascending pair keeps the smaller key (padding sorts last)
    }
  }
  if constexpr (STRIDE > 1) fr_wave_sort_stage<NR, SIZE, STRIDE / 2>(key, lane, pick);
}
template <int NR, int SIZE, class T, class Pick>
__device__ __forceinline__ void fr_wave_sort_sizes(T (&key)[NR], int lane, Pick pick) {
  fr_wave_sort_stage<NR, SIZE, SIZE / 2>(key, lane, pick);
  if constexpr (SIZE < 64 * NR) fr_wave_sort_sizes<NR, SIZE * 2>(key, lane, pick);
}

// A rank of more than R and at most 64 * NR entries, by a whole wave: NR entries a lane (entry r * 64 + l in register r of
// lane l), both sorts as networks across the lanes.
template <int NR>
__device__ __noinline__ int fr_lean_wide_rank(const uint16_t* bucket /* LDS */, int w, const int32_t* labels, const double* xs,
                                          int lab_base, int abits, const LnTabEntry* lntab /* LDS */, uint64_t* out) {
  const int lane = threadIdx.x & 63;
  uint32_t key[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    key[r] = 0xFFFFFFFFu;
    if (r * 64 + lane < w) {
      const int arc = (int)bucket[r * 64 + lane];
      key[r] = (uint32_t)arc | ((uint32_t)(labels[arc] - lab_base) << abits);
    }
  }
  // (the keys alone go through the network, x is fetched by arc number afterwards, as the thread-per-rank path does)
  fr_wave_sort_sizes<NR, 2>(key, lane, [](uint32_t a, uint32_t b, bool take_min) { return take_min ? min(a, b) : max(a, b); });
  double x[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) x[r] = r * 64 + lane < w ? xs[key[r] & ((1u << abits) - 1u)] : 0.0;
  // The runs of one word, folded in arc order with every lane at work: an entry knows where its run begins (a prefix
  // maximum over the entries that open one) and hence its place in it; step s folds the s-th entry of every run into the sum
  // its left neighbour holds -- as many steps as the longest run is long (two or three as a rule: walking the rank entry by
  // entry, one wave-uniform trip per entry, was most of this function's time).  The last entry of a run emits the record's
  // key, the others padding: the sort below brings the records to the front.  (Entry 64's left neighbour is lane 63's
  // first register.)
  uint32_t lab[NR];
  bool opens[NR];
  int start[NR], pos[NR], longest = 0, carry = -1;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    lab[r] = key[r] >> abits;
    uint32_t lab_left = (uint32_t)__shfl_up((int)lab[r], 1, 64);
    if (r > 0) {
      const uint32_t across = (uint32_t)__shfl((int)lab[r - 1], 63, 64);
      lab_left = lane == 0 ? across : lab_left;
    }
    const int e = r * 64 + lane;
    opens[r] = e < w && (e == 0 || lab_left != lab[r]);
    start[r] = opens[r] ? e : -1;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(start[r], d, 64);
      if (lane >= d) start[r] = max(start[r], o);
    }
    start[r] = max(start[r], carry);
    carry = __shfl(start[r], 63, 64);
    pos[r] = e < w ? e - start[r] : 0;
    longest = max(longest, pos[r]);
  }
  for (int d = 32; d > 0; d >>= 1) longest = max(longest, __shfl_xor(longest, d, 64));
  double acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = x[r];
  for (int s = 1; s <= longest; ++s) {  // (wave-uniform trips)
    double left[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      left[r] = __shfl_up(acc[r], 1, 64);
      if (r > 0) {
        const double across = __shfl(acc[r - 1], 63, 64);
        left[r] = lane == 0 ? across : left[r];
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const double folded = fr_log_add_fast(left[r], x[r], lntab);
      if (pos[r] == s) acc[r] = folded;
    }
  }
  int m = 0;
  double outk[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    bool opens_right = __shfl_down((int)opens[r], 1, 64) != 0;
    if (r + 1 < NR) {
      const bool across = __shfl((int)opens[r + 1 < NR ? r + 1 : r], 0, 64) != 0;
      opens_right = lane == 63 ? across : opens_right;
    }
    const int e = r * 64 + lane;
    const bool closes = e < w && (e == w - 1 || opens_right);
    m += __popcll(__ballot(closes));
    outk[r] = __longlong_as_double(0x7FEFFFFFFFFFFFFFll);
    if (closes) outk[r] = __longlong_as_double((long long)(((uint64_t)fr_desc_key((float)acc[r]) << 30) | (uint64_t)lab[r]));  // BaseFloat narrowing of :126
  }
  fr_wave_sort_sizes<NR, 2>(outk, lane, [](double a, double b, bool take_min) { return take_min ? fmin(a, b) : fmax(a, b); });
#pragma unroll
  for (int r = 0; r < NR; ++r)
    if (r * 64 + lane < m) out[r * 64 + lane] = (uint64_t)__double_as_longlong(outk[r]);
  return m;
}

// ---- the lean kernel ----
// Lattices with at most kLnStates states, 65 535 arcs and 65 535 frames, sort keys of 32 bits, and at most
// kLnWideMax word arcs across any frame: on the synthetic and the shipped lattices, all of them.  (Up to kFrLdsSpans
// arcs and kLnStatesLds states the arcs' rank spans stay in LDS, beyond that in the lattice's scratch rows.)  Twice the threads and half the registers of the general
// kernel (four waves per SIMD), and the work laid out over the lattice's distinct state times instead of its
// frames: a frame's posteriors differ from its predecessor's only where some arc starts or ends, i.e. at a state's
// time, so the r-th distinct time below T (its RANK r) stands for the frames from there to the next one.  An arc
// covers the ranks [rank(src), rank(dst)); a tile is kLnTile ranks -- two full rounds of one thread per rank --
// whatever the frames between them.  A lattice that does not fit is listed for the general kernel, which is
// launched behind this one over the list.
static constexpr int kLnStates = 8192;     // states of a lattice at most: their ranks lie under the buckets during the arc pass
static constexpr int kLnStatesLds = 4096;  // ... and of a lattice whose spans stay in LDS: the rank times end where the spans begin
static constexpr int kLnBlock = 128;       // ranks per entry of the table of first arcs (lattices whose spans lie in L2)
static constexpr int kLnTile = 1024;    // ranks per tile
static constexpr int kLnBatch = 6;      // arcs per thread and round of the arc pass
static constexpr int kLnWideMax = 256;  // a rank with more than kFrRegs entries and at most this many is folded by a whole wave (one, two or four entries a lane)

// BIG: the instantiation for the lattices the first one lists (spans in L2, up to kLnStates states; whatever else did
// not fit is tried once more and listed for the general kernel); the first one keeps the shorter code of a lattice
// whose spans stay in LDS.
// SC: the size class -- threads, bucket entries, states and ranks per tile are 1 / SC of the full-size kernel's, so a
// lattice of a few hundred states runs in 128 threads with 16 KiB of LDS and eight such workgroups share a CU (80 000
// lattices of 125 states took 3.25 ms in 512-thread workgroups, two a CU).  The host lists every class's lattices.
// PIECES: the second instantiation over the pieces of long lattices (FrameItem, klat_accum.h) instead of a list of lattices.
// R: bucket entries a rank's thread sorts in registers.  Sixteen serve lattices whose frames hold ten arcs or so
// (BASELINE's fanout 3); a lattice with more than twelve entries a rank on average is LISTED (a.dense; a piece: marked in
// a.piece_dense) and left to the instantiation with thirty-two (DENSE), which
// pays for its registers with half the waves -- a wave per rank of more than sixteen entries, the only other way this
// kernel has, costs forty times a thread per rank.
template <bool BIG, bool CLOCK, int SC, bool PIECES = false, int R = kFrRegs, int DENSE = 0>
__global__ __launch_bounds__(kFrLeanThreads / SC / (DENSE == 2 ? 2 : 1), (BIG || CLOCK) && R == kFrRegs ? 4 : 1) void frame_lean_kernel(FrameArgs a, int span_cap) {
  static_assert(!DENSE || (BIG && R == 2 * kFrRegs), "the dense pass is the second instantiation with thirty-two registers a rank");
  static_assert(!BIG || SC == 1, "the second instantiation is full size");
  static_assert(!PIECES || BIG, "pieces take the second instantiation");
  // (the dense pass: thirty-two registers a rank leave two waves a SIMD, one workgroup of 512 a CU.  DENSE == 2: two
  // workgroups of 256 instead, two lattices whose phases overlap -- a fifth faster where the pass has lattices for four
  // rounds of that and more, slower where it has few: the host chooses)
  constexpr int NT = kFrLeanThreads / SC / (DENSE == 2 ? 2 : 1);
  constexpr int kEntries = kFrLdsEntries / SC;    // bucket entries of a tile in LDS
  constexpr int kStatesLds = kLnStatesLds / SC;   // states of a lattice whose spans stay in LDS
  constexpr int kTile = kLnTile / SC;             // ranks per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char fr_smem[];
  uint16_t* lbucket = reinterpret_cast<uint16_t*>(fr_smem);  // [kEntries]: a tile's buckets as arc numbers
  uint16_t* lrank = lbucket;                                 // [S]: rank of every state, during the arc pass only
  // [kStatesLds + 2], or into the spans' place for a lattice whose spans lie in L2: the time of rank r; rtime[nr] = T
  uint16_t* rtime = reinterpret_cast<uint16_t*>(fr_smem + sizeof(uint16_t) * kEntries);
  uint32_t* lspan = reinterpret_cast<uint32_t*>(fr_smem + sizeof(uint16_t) * (kEntries + kStatesLds + 2));  // [span_cap]: rank(src) | rank(dst) << 16
  __shared__ int lboff[kTile + 1];  // first bucket entry of every rank of the tile
  __shared__ int lcur[kTile];       // scatter cursors
  __shared__ int sh[3 * NT / 64];
  __shared__ int tile_max, lab_min, lab_max, span_max, unfit, num_ranks, entries, wide_n, sum_stored, sum_dense, descending;
  // the tile's ranks of more than kFrRegs entries: every one of them where a lattice's frames hold twenty arcs and more
  // (fanout four, six: a wave per rank is forty times the work of a thread per rank, and a tenth of the general kernel's)
  __shared__ uint16_t wide_rank[kTile];
  // the tile's ranks in the order the threads take them: those of at most kW0 entries first, those of at most kW1 next,
  // the others last (each in ascending order) -- a wave sorts as many registers as its largest bucket needs, and on
  // lattices whose frames hold ten arcs or so (one rank in four has eight entries or fewer, all but one in thirty twelve
  // or fewer) a wave of consecutive ranks nearly always held one of more than twelve
  __shared__ uint16_t order[kTile];
  constexpr int kW0 = R == kFrRegs ? 8 : 16, kW1 = R == kFrRegs ? 12 : 24;
  __shared__ int blk_first[kLnStates / kLnBlock + 2];  // first arc whose first rank lies in or behind block b of kLnBlock ranks
  __shared__ unsigned long long rec_at;
  __shared__ LnTabEntry lntab[kLnTabSize];
  if (BIG && !PIECES && !DENSE && (int)blockIdx.x >= *a.redo_n) return;
  FrameItem item = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (PIECES) {
    item = a.piece_items[blockIdx.x];
    if (item.u < 0) return;  // (a slot the slicing left empty)
    if (DENSE && !a.piece_dense[blockIdx.x]) return;
  }
  if (DENSE && !PIECES && (int)blockIdx.x >= *a.dense_n) return;
  const int u = PIECES ? item.u : DENSE ? a.dense[blockIdx.x] : BIG ? a.redo_list[blockIdx.x] : (a.lattice_list ? a.lattice_list[blockIdx.x] : (int)blockIdx.x);
  const int tid = threadIdx.x;
  if (tid < kLnTabSize) lntab[tid] = a.ln_table[tid];  // (barriers follow before its first use)
  // a piece: its arcs, its states (numbered from s_lo), its frames (counted from f0)
  const int s_lo = PIECES ? item.s_lo : 0, f_lo = PIECES ? item.f0 : 0;
  const int64_t a0 = a.arc_off[u] + (PIECES ? item.a_lo : 0);
  const int nA = PIECES ? item.a_hi - item.a_lo : (int)(a.arc_off[u + 1] - a0);
  const int64_t s0 = a.state_off[u] + s_lo;
  const int S = PIECES ? item.s_hi - item.s_lo : (int)(a.state_off[u + 1] - s0);
  const int64_t pb = padded_state_base(a.state_off[u], u) + s_lo;
  const double tot = a.total[u];
  const int T = PIECES ? item.f1 - item.f0 : a.num_frames[u];
  // x of every arc (the front half of the lattice's rows; a piece has rows of its own: pieces share the arcs that cross a cut)
  double* xs = reinterpret_cast<double*>(PIECES ? a.piece_scratch + item.scratch : a.arc_info + a0);
  const int64_t fbase = a.frame_base[u] + f_lo;
  const ScaleParams sp = a.sp;
  auto leave = [&]() {  // (workgroup-uniform; a lattice is listed once, whichever of its pieces leaves first)
    if (tid == 0 && (atomicOr(&a.listed[u], BIG ? 2 : 1) & (BIG ? 2 : 1)) == 0) {
      if (BIG) a.redo2_list[atomicAdd(a.redo2_n, 1)] = u;
      else a.redo_list[atomicAdd(a.redo_n, 1)] = u;
    }
  };
  if (nA > 65535 || S > kLnStates || T > 65535 || (!BIG && (nA > span_cap || S > kStatesLds))) {
    leave();
    return;
  }
  // the spans of the arcs in LDS (two 16-bit ranks each), or -- more arcs or states than fit -- in the back half of
  // the lattice's scratch rows (L2): the tiles then read them a batch per thread, and the rank times take their place
  const bool spans_lds = BIG ? (nA <= span_cap && S <= kStatesLds) : true;  // workgroup-uniform
  uint32_t* gspan = reinterpret_cast<uint32_t*>(xs + nA);
  if (nA == 0) {  // nothing crosses any frame (and the unconditional loads below read the lattice's first arc)
    for (int f = tid; f < T; f += NT) a.frame_ref[fbase + f] = make_int2(0, 0);
    if (tid == 0 && !PIECES) a.num_records[u] = 0, a.num_stored[u] = 0, a.rec_ptr[u] = a.pool, a.rec_fmt[u] = 0;
    return;
  }
  long long clk[6] = {0, 0, 0, 0, 0, 0};  // spans, counts, scatter, (unused), frames, (unused)
  long long t_prev = CLOCK ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define FR_LAP(k)                                                      \
  do {                                                                 \
    if (CLOCK && a.phase_clock) {                                      \
      const long long t_now = (long long)__builtin_amdgcn_s_memtime(); \
      clk[k] += t_now - t_prev;                                        \
      t_prev = t_now;                                                  \
    }                                                                  \
  } while (0)
  // ---- 0a. ranks: every thread owns kSt consecutive states ----
  if (tid == 0) lab_min = 0x7FFFFFFF, lab_max = (int)0x80000000, span_max = 0, unfit = 0, entries = 0, sum_stored = 0, sum_dense = 0, descending = 0;
  if (tid < kLnStates / kLnBlock + 2) blk_first[tid] = nA;
  auto rank_pass = [&](auto per_thread) {  // (two sizes: a thread of a small lattice should not walk sixteen states while three in four idle)
    constexpr int kSt = decltype(per_thread)::value;
    int tm[kSt], fl[kSt], mine = 0;
    const int sfirst = tid * kSt;
    int prev = (sfirst > 0 && sfirst <= S) ? min(max(a.state_time[s0 + sfirst - 1] - f_lo, 0), T) : -1;
    bool descends = false;
#pragma unroll
    for (int q = 0; q < kSt; ++q) {
      const int s = sfirst + q;
      tm[q] = (s < S) ? min(max(a.state_time[s0 + s] - f_lo, 0), T) : T;
      descends |= s < S && tm[q] < prev;
      fl[q] = (s < S && tm[q] > prev && tm[q] < T) ? 1 : 0;  // the first state of a time below T: a rank
      prev = (s < S) ? tm[q] : prev;
      mine += fl[q];
    }
    int nr;
    int r = fr_block_scan<NT, false>(mine, &nr, sh);  // (behind its barrier lab_min ... are set)
    if (__any(descends) && (tid & 63) == 0) atomicOr(&descending, 1);
    // times that do not descend with the state number (the synthetic lattices; a top-sorted lattice need not be
    // like that): a state opens a rank where its time exceeds its predecessor's
#pragma unroll
    for (int q = 0; q < kSt; ++q) {
      const int s = sfirst + q;
      if (fl[q]) rtime[r] = (uint16_t)tm[q];
      r += fl[q];
      if (s < S) lrank[s] = (uint16_t)(r - (tm[q] < T ? 1 : 0));  // ranks whose time is below this state's
    }
    if (tid == 0) rtime[nr] = (uint16_t)T, num_ranks = nr;
    __syncthreads();
    if (SC == 1 && descending) {  // (workgroup-uniform; a smaller class has no room for the frame bits and leaves the lattice to the second instantiation)
      // any order of the times: a bit per frame that some state sits at, ranks by counting the bits below
      uint32_t* bits = reinterpret_cast<uint32_t*>(lrank + kLnStates);   // [2048]: frames 0 .. 65535 (behind the ranks, under the buckets)
      uint16_t* wpre = reinterpret_cast<uint16_t*>(bits + 2048);         // [2048]: bits set in the words before this one
      constexpr int kW = 2048 / NT;
      for (int w = tid; w < 2048; w += NT) bits[w] = 0;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kSt; ++q)
        if (sfirst + q < S && tm[q] < T) atomicOr(&bits[tm[q] >> 5], 1u << (tm[q] & 31));
      __syncthreads();
      uint32_t wd[kW];
      int cnt = 0;
#pragma unroll
      for (int q = 0; q < kW; ++q) wd[q] = bits[tid * kW + q], cnt += __popc(wd[q]);
      int r2 = fr_block_scan<NT, false>(cnt, &nr, sh + NT / 64);
#pragma unroll
      for (int q = 0; q < kW; ++q) {
        wpre[tid * kW + q] = (uint16_t)r2;
        for (uint32_t m = wd[q]; m; m &= m - 1) rtime[r2++] = (uint16_t)((tid * kW + q) * 32 + __ffs((int)m) - 1);
      }
      if (tid == 0) rtime[nr] = (uint16_t)T, num_ranks = nr;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kSt; ++q) {
        const int s = sfirst + q, t = tm[q];
        if (s < S) lrank[s] = (uint16_t)(t < T ? (int)wpre[t >> 5] + __popc(bits[t >> 5] & ((1u << (t & 31)) - 1u)) : nr);
      }
      __syncthreads();
    }
  };
  if (!BIG || S <= kStatesLds)  // (workgroup-uniform)
    rank_pass(std::integral_constant<int, kStatesLds / NT>());
  else
    rank_pass(std::integral_constant<int, kLnStates / NT>());
  if (SC > 1 && descending) {  // (workgroup-uniform)
    leave();
    return;
  }
  // descending times: the arcs' first ranks need not ascend with the arc number, every tile looks at all arcs
  const bool ranged = descending == 0;
  const int nr = num_ranks;
  FR_LAP(3);
  // ---- 0b. rank span and x of every arc; the label range of the lattice ----
  // Two rounds of kLnBatch arcs per thread at a time (one trip of the loop for a lattice whose spans stay in LDS): the
  // arrays of both rounds are asked for before anything waits, the gathers of alpha / beta follow round by round, and
  // nothing is stored before the last load has landed (a store among pending loads makes every later wait a wait
  // for all of them).
  for (int base = 0; base < (BIG ? nA : 1); base += 2 * kLnBatch * NT) {
    int lmin_t = 0x7FFFFFFF, lmax_t = (int)0x80000000, span_max_t = 0, entries_t = 0;  // (reduced at the end of the trip: nothing of a thread's lives across trips)
    static_assert(BIG || 2 * kLnBatch * NT >= kFrLdsSpans / SC, "one trip covers the arcs of a lattice whose spans stay in LDS");
    int il[2][kLnBatch], t0[2][kLnBatch], t1[2][kLnBatch];
    float gr[2][kLnBatch], ac4[2][kLnBatch];
    double xq[2][kLnBatch];
    const int last = nA - 1;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < kLnBatch; ++q) {
        const int i = min(base + (h * kLnBatch + q) * NT + tid, last);
        il[h][q] = a.ilabel[a0 + i];
        t0[h][q] = a.src[a0 + i] - s_lo;
        t1[h][q] = a.dst[a0 + i] - s_lo;
        gr[h][q] = a.graph[a0 + i];
        ac4[h][q] = a.acoustic[a0 + i];
      }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double al[kLnBatch], be[kLnBatch];
#pragma unroll
      for (int q = 0; q < kLnBatch; ++q) {
        al[q] = a.alpha[pb + t0[h][q]];
        be[q] = a.beta[pb + t1[h][q]];
        t0[h][q] = (int)lrank[t0[h][q]];
        t1[h][q] = (int)lrank[t1[h][q]];
      }
#pragma unroll
      for (int q = 0; q < kLnBatch; ++q) {
        // fw[u] + bw[nextstate] - cost - total (latbin/lattice-to-word-frame-post.cc:102-104,122), formed
        // here from alpha / beta: no posterior pass has to run before this kernel
        const double through = al[q] + be[q] - arc_cost(gr[h][q], ac4[h][q], il[h][q] != 0, sp);
        const double lp = (through > -INFINITY) ? through - tot : -INFINITY;
        double x = 0.0;
        int r1 = t0[h][q];  // (an empty span is kept as [r, r) at its source's rank: the first ranks of the arcs then ascend with the arc number)
        if (base + (h * kLnBatch + q) * NT + tid < nA && il[h][q] != 0 && t1[h][q] > t0[h][q]) {
          r1 = t1[h][q];
          lmin_t = min(lmin_t, il[h][q]), lmax_t = max(lmax_t, il[h][q]);
          // lp holds alpha - (double)g - (double)a + beta - total; the reference forms the arc
          // cost here as a float sum, so swap the double sum for the float one
          float g, ac;
          arc_cost_parts(gr[h][q], ac4[h][q], true, sp, g, ac);
          // (an arc of infinite cost is log-zero either way; the correction would be inf - inf)
          x = (lp > -INFINITY) ? lp + ((double)g + (double)ac) - (double)(g + ac) : lp;
          span_max_t = max(span_max_t, r1 - t0[h][q]);
          entries_t += r1 - t0[h][q];
        }
        xq[h][q] = x;
        t1[h][q] = r1;
      }
    }
    if (spans_lds) {  // (workgroup-uniform)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < kLnBatch; ++q) {
          const int i = base + (h * kLnBatch + q) * NT + tid;
          if (i < nA) {
            xs[i] = xq[h][q];
            lspan[i] = (uint32_t)t0[h][q] | ((uint32_t)t1[h][q] << 16);
          }
        }
    } else {
      // spans to L2, and the table of first arcs per block of ranks: the lanes of a wave hold consecutive arcs, the
      // first lane of every run of one block speaks for it (where the first ranks ascend with the arc number)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < kLnBatch; ++q) {
          const int i = base + (h * kLnBatch + q) * NT + tid;
          const int blk = t0[h][q] / kLnBlock;
          const int before = __shfl_up(blk, 1, 64);
          if (i < nA) {
            xs[i] = xq[h][q];
            gspan[i] = (uint32_t)t0[h][q] | ((uint32_t)t1[h][q] << 16);
            if ((tid & 63) == 0 || before != blk) atomicMin(&blk_first[blk], i);
          }
        }
    }
    for (int d = 32; d > 0; d >>= 1) lmin_t = min(lmin_t, __shfl_xor(lmin_t, d, 64)), lmax_t = max(lmax_t, __shfl_xor(lmax_t, d, 64));
    for (int d = 32; d > 0; d >>= 1) span_max_t = max(span_max_t, __shfl_xor(span_max_t, d, 64)), entries_t += __shfl_xor(entries_t, d, 64);
    if ((tid & 63) == 0) atomicMin(&lab_min, lmin_t), atomicMax(&lab_max, lmax_t), atomicMax(&span_max, span_max_t), atomicAdd(&entries, entries_t);
  }
  __syncthreads();
  // Where the times do not descend with the state number, the arcs' first ranks ascend with the arc number -- a tile's
  // arcs are one range of arcs -- because klat_batch_create takes only arcs grouped by ascending source
  // (csrc/klat_pack.cc:378, csrc/klat_pack_device.hip:158).  (The ranks of the states are not read beyond this barrier: the buckets
  // take their place.)
  // The lattice's records: at most one per (arc, rank) entry, taken from the batch's pool in one piece (so that
  // nothing has to count the entries before this kernel); a pool that is short leaves the lattice to the general
  // kernel, and the caller learns from pool_used what the next call needs.  Asked for here, looked at behind the
  // first tile's counts: the atomic's trip to memory is not waited for.
  if (R == kFrRegs && num_ranks > 0 && (long long)entries > 12LL * num_ranks) {  // (workgroup-uniform) crowded frames: the dense pass's
    if (tid == 0) {
      if (PIECES) a.piece_dense[blockIdx.x] = 1;
      else a.dense[atomicAdd(a.dense_n, 1)] = u;
    }
    return;
  }
  unsigned long long my_at = 0;
  if (tid == 0) my_at = atomicAdd(a.pool_used, (unsigned long long)entries);
  if (!spans_lds) {  // (workgroup-uniform) a block in which no arc begins takes the next block's first arc
    if (tid == 0)
      for (int k = kLnStates / kLnBlock; k >= 0; --k) blk_first[k] = min(blk_first[k], blk_first[k + 1]);
    __syncthreads();
  }
  const int longest = span_max;
  // (label - lab_min) and the arc number fit one 32-bit sort key (0xFFFFFFFF stays free for padding), and the
  // label range 30 bits (the folded keys sort as doubles)
  int abits = 1;
  while ((1 << abits) < nA) ++abits;
  const int lab_base = PIECES ? item.lab_base : lab_min;  // (a lattice's pieces share the lattice's: their records are decoded alike)
  const bool keys_fit = lab_max < lab_min ||
                        ((uint64_t)((int64_t)lab_max - (int64_t)lab_base) < ((1ull << (32 - abits)) - 1ull) &&
                         (int64_t)lab_max - (int64_t)lab_base < ((int64_t)1 << 30));
  if (unfit || !keys_fit) {
    leave();
    return;
  }
  const uint32_t amask = (1u << abits) - 1u;
  // the frames before the first rank hold nothing
  for (int f = tid, first = (nr > 0) ? (int)rtime[0] : T; f < first; f += NT) a.frame_ref[fbase + f] = make_int2(0, 0);
  FR_LAP(0);
  int nbase = 0;   // bucket entries of the tiles before this one
  int stored = 0;  // records this thread wrote
  int dense = 0;   // (frame, word) entries of its ranks: every frame counts its records, shared or not
  // ranks per tile: kTile, or -- spans in L2, a tile's arcs staged in LDS -- as many blocks of kLnBlock ranks as keep
  // the arcs of a tile (with a block and the longest span before it) inside the stage
  const int stage_cap = (int)((sizeof(uint16_t) * (kStatesLds + 2) + sizeof(uint32_t) * (size_t)span_cap) / sizeof(uint32_t)) -
                        (int)(sizeof(uint16_t) * (kLnStates + 2) / sizeof(uint32_t));
  int tile_ranks = kTile;
  if (!spans_lds && ranged && nr > 0) {
    const long long fit = (long long)max(stage_cap, 0) * nr / nA - kLnBlock - longest;  // ranks whose arcs fill the stage, less the slack
    tile_ranks = (int)max((long long)kLnBlock, min((long long)kTile, fit / kLnBlock * kLnBlock));
  }
  // ... and as many as keep a tile's bucket entries inside their LDS: a lattice whose frames hold thirty arcs each (fanout
  // six, arcs a few states long) fills the buckets with a quarter of a full tile's ranks -- tiles of the average rank's
  // entries and a third; a tile that overflows all the same leaves the lattice to the next kernel
  if (nr > 0 && kTile > kLnBlock) {
    const long long per_rank = ((long long)entries + nr - 1) / nr;
    const long long fit = (long long)kEntries * 3 / 4 / max(per_rank, 1LL);
    tile_ranks = (int)max(32LL, min((long long)tile_ranks, fit >= kLnBlock ? fit / kLnBlock * kLnBlock : fit / 32 * 32));  // (frames of a hundred arcs: tiles of 96 ranks, 64, 32)
  }
  for (int c0 = 0; c0 < nr; c0 += tile_ranks) {
    const int tc = min(tile_ranks, nr - c0);
    // ---- 1. entries per rank from a difference array over the spans, and the buckets' offsets ----
    for (int k = tid; k <= tc; k += NT) lboff[k] = 0;
    if (tid == 0) tile_max = 0, wide_n = 0;
    // first arc whose first rank is >= rank (spans in LDS): the wave's lanes probe 64 places of the range at once -- three
    // trips to LDS for 6 000 arcs where a binary search by every thread for itself took thirteen (every wave finds the same)
    auto first_at_least = [&](int rank) {
      const int l64 = tid & 63;
      int l = 0, h = nA;
      while (h - l > 64) {  // (wave-uniform)
        const int step = (h - l + 63) >> 6;
        const int at = min(l + (l64 + 1) * step - 1, h - 1);
        const int below = __popcll(__ballot((int)(lspan[at] & 0xFFFFu) < rank));  // the probes are ascending: the first `below` lanes
        const int nl = below ? min(l + below * step, h) : l;                     // behind the last probe below rank
        const int nh = below < 64 ? min(l + (below + 1) * step, h) : h;          // up to and with the first probe not below
        l = nl, h = nh;
      }
      const int below = __popcll(__ballot(l + l64 < h && (int)(lspan[min(l + l64, h - 1)] & 0xFFFFu) < rank));
      return l + below;
    };
    // arcs that can cover a rank of the tile: first rank below the tile's end, and (no arc is longer than
    // `longest`) above its start less that -- by binary search over the spans in LDS, or to the block of ranks
    // from the table of first arcs; all arcs where the first ranks do not ascend
    int arc_lo = 0, arc_hi = nA;
    if (ranged && spans_lds) arc_lo = first_at_least(c0 - longest + 1), arc_hi = first_at_least(c0 + tc);
    if (ranged && !spans_lds) arc_lo = blk_first[max(c0 - longest + 1, 0) / kLnBlock], arc_hi = blk_first[(c0 + tc + kLnBlock - 1) / kLnBlock];
    if (c0 == 0 && tid == 0) {
      rec_at = my_at;
      if (my_at + (unsigned long long)entries > (unsigned long long)a.pool_cap) {
        atomicOr(&unfit, 1);
        atomicAdd(a.pool_used + 1, (unsigned long long)entries);  // asked for in vain (the host does not count it twice)
      }
    }
    // the spans of the tile's arcs in LDS: where they are anyway, or -- a lattice whose spans lie in L2 -- staged behind
    // the rank times, in the place the spans of a smaller lattice would have: one coalesced trip per tile as a rule, one
    // per stage-full of arcs where a tile has more (a larger lattice whose first ranks do not ascend looks at all arcs)
    uint32_t* stage = reinterpret_cast<uint32_t*>(fr_smem + sizeof(uint16_t) * (kEntries + kLnStates + 2));
    if (!spans_lds && stage_cap < NT) {  // (workgroup-uniform: no stage to speak of)
      leave();
      return;
    }
    const int chunk = spans_lds ? max(arc_hi - arc_lo, 1) : stage_cap;  // arcs per trip
    const bool one_trip = arc_hi - arc_lo <= chunk;
    for (int cb = arc_lo; cb < arc_hi || cb == arc_lo; cb += chunk) {
      const int ce = min(cb + chunk, arc_hi);
      if (!spans_lds)
        for (int i = cb + tid; i < ce; i += NT) stage[i - cb] = gspan[i];
      __syncthreads();
      const uint32_t* tspan = spans_lds ? lspan : stage - cb;  // tspan[i]: the span of arc i, cb <= i < ce
      for (int i = cb + tid; i < ce; i += NT) {
        const uint32_t sp32 = tspan[i];
        const int lo = max((int)(sp32 & 0xFFFFu), c0) - c0, hi = min((int)(sp32 >> 16), c0 + tc) - c0;
        if (hi > lo) {
          atomicAdd(&lboff[lo], 1);
          atomicAdd(&lboff[hi], -1);
        }
      }
      __syncthreads();
    }
    int n = 0;
    {
      constexpr int kCh = kTile / NT;  // every thread owns kCh consecutive ranks
      int dl[kCh], sum_d = 0;
#pragma unroll
      for (int q = 0; q < kCh; ++q) {
        const int k = tid * kCh + q;
        dl[q] = (k < tc) ? lboff[k] : 0;
        sum_d += dl[q];
      }
      int tot_d;
      int run = fr_block_scan<NT, false>(sum_d, &tot_d, sh);  // (all reads of the array are done)
      int sum_c = 0, cmax = 0;
#pragma unroll
      for (int q = 0; q < kCh; ++q) {
        run += dl[q];
        dl[q] = (tid * kCh + q < tc) ? run : 0;  // entries of the rank
        cmax = max(cmax, dl[q]);
        sum_c += dl[q];
      }
      int narrow_c = 0;  // its ranks of at most kW0 entries | those of kW0 + 1 ... kW1 << 16
#pragma unroll
      for (int q = 0; q < kCh; ++q)
        if (tid * kCh + q < tc) narrow_c += dl[q] <= kW0 ? 1 : dl[q] <= kW1 ? 0x10000 : 0;
      int narrow_n;
      const int2 before = fr_block_scan2<NT, false>(sum_c, narrow_c, &n, &narrow_n, sh + NT / 64);
      int off = before.x;
      int at0 = before.y & 0xFFFF, at1 = (narrow_n & 0xFFFF) + (before.y >> 16);
      int at2 = (narrow_n & 0xFFFF) + (narrow_n >> 16) + min(tid * kCh, tc) - (before.y & 0xFFFF) - (before.y >> 16);
#pragma unroll
      for (int q = 0; q < kCh; ++q) {
        const int k = tid * kCh + q;
        if (k < tc) {
          lboff[k] = off, lcur[k] = off;
          order[dl[q] <= kW0 ? at0++ : dl[q] <= kW1 ? at1++ : at2++] = (uint16_t)k;
        }
        off += dl[q];
      }
      if (tid == 0) lboff[tc] = n;
      for (int d = 32; d > 0; d >>= 1) cmax = max(cmax, __shfl_xor(cmax, d, 64));
      if ((tid & 63) == 0) atomicMax(&tile_max, cmax);
    }
    __syncthreads();
    if (tile_max > kLnWideMax || n > kEntries || unfit) {  // workgroup-uniform
      leave();
      return;
    }
    FR_LAP(1);
    // ---- 2. every arc drops its number into the buckets of its ranks ----
    for (int cb = arc_lo; cb < arc_hi; cb += chunk) {
      const int ce = min(cb + chunk, arc_hi);
      if (!one_trip) {  // (workgroup-uniform; else the stage still holds the tile's arcs)
        __syncthreads();
        for (int i = cb + tid; i < ce; i += NT) stage[i - cb] = gspan[i];
        __syncthreads();
      }
      const uint32_t* tspan = spans_lds ? lspan : stage - cb;
      for (int i = cb + tid; i < ce; i += NT) {
        const uint32_t sp32 = tspan[i];
        const int lo = max((int)(sp32 & 0xFFFFu), c0) - c0, hi = min((int)(sp32 >> 16), c0 + tc) - c0;
        for (int c = lo; c < hi; c += 4) {  // four cursors at a time: the atomics' round trips overlap
          int at[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) at[q] = (c + q < hi) ? atomicAdd(&lcur[c + q], 1) : -1;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (at[q] >= 0) lbucket[at[q]] = (uint16_t)i;
        }
      }
    }
    __syncthreads();
    FR_LAP(2);
    // ---- 3. one thread per rank, NT at a time: sort by (word, arc), fold, sort by (logp desc, word), place ----
    for (int r0 = 0; r0 < tc; r0 += NT) {
      const int k = (r0 + tid < tc) ? (int)order[r0 + tid] : tc;
      const int lo = (k < tc) ? lboff[k] : 0;
      const int w_all = (k < tc) ? lboff[k + 1] - lo : 0;
      const bool wide = w_all > R;  // (rare: left to a whole wave below; its slots are reserved here)
      const int w = wide ? 0 : w_all;
      // W: the registers this wave sorts -- no bucket of its ranks holds more entries
      auto rank_round = [&](auto width) {
        constexpr int W = decltype(width)::value;
        int mk = 0;
        // (every load below is unconditional: a load inside a branch waits for its data before the branch
        // closes, and sixteen gathers would take sixteen trips to L2 one after the other.  The slots beyond
        // the bucket all read the lattice's first arc: one more request per instruction, not one per lane)
        uint32_t k32[W];
  #pragma unroll
        for (int j = 0; j < W; ++j) k32[j] = (uint32_t)lbucket[lo + j];  // (beyond the buckets lies rtime: read, not used)
  #pragma unroll
        for (int j = 0; j < W; ++j) asm volatile("" : "+v"(k32[j]));  // (the reads stay outside the selects)
  #pragma unroll
        for (int j = 0; j < W; ++j) k32[j] = (j < w) ? k32[j] : 0u;
        // (gathers served by L2; 32-bit byte offsets from the lattice's first arc: one address register each)
        const char* lab_row = reinterpret_cast<const char*>(a.ilabel + a0);
        const char* x_row = reinterpret_cast<const char*>(xs);
        {
          int32_t lab[W];
  #pragma unroll
          for (int j = 0; j < W; ++j) lab[j] = *reinterpret_cast<const int32_t*>(lab_row + (k32[j] << 2));
          // sort keys (label - lab_base | arc number) in 32 bits: a compare-exchange is two instructions and no
          // payload moves; x is fetched by arc number after the sort
  #pragma unroll
          for (int j = 0; j < W; ++j) k32[j] = (j < w) ? (k32[j] | ((uint32_t)(lab[j] - lab_base) << abits)) : 0xFFFFFFFFu;
        }
        if constexpr (W == 8) fr_sort8_net(k32, FrCeU32());
        else if constexpr (W == 12) fr_sort12_net(k32, FrCeU32());
        else if constexpr (W == 16) fr_sort16_net(k32, FrCeU32());
        else if constexpr (W == 24) fr_sort24_net(k32, FrCeU32());
        else if constexpr (W == 32) fr_sort32_net(k32, FrCeU32());
        else fr_sort32_sizes<W, 2, W>(k32);
        double ex[W];
  #pragma unroll
        for (int j = 0; j < W; ++j) ex[j] = *reinterpret_cast<const double*>(x_row + ((j < w ? k32[j] & amask : 0u) << 3));
        // fold on the sort keys themselves, the runs of one word in arc order: the key of a run is emitted at its
        // last entry as the positive finite double desc(float logp) << 30 | label - lab_base (sorted by v_min_f64 /
        // v_max_f64), the other entries turn into padding
        double dk[W];
        double acc = ex[0];
  #pragma unroll
        for (int j = 1; j <= W; ++j) {
          const uint32_t lab = k32[j - 1] >> abits;
          const bool cont = j < W && j < w && (k32[j < W ? j : 0] >> abits) == lab;
          const bool emit = !cont && j - 1 < w;
          const uint64_t key = ((uint64_t)fr_desc_key((float)acc) << 30) | (uint64_t)lab;  // BaseFloat narrowing of :126
          dk[j - 1] = __longlong_as_double((long long)(emit ? key : 0x7FEFFFFFFFFFFFFFull));
          mk += emit ? 1 : 0;
          if (j < W) {
            double next = ex[j];
            if (__any(cont)) {  // (skipped unless some frame of the wave has two arcs of one word here)
              const double folded = fr_log_add_fast(acc, ex[j], lntab);
              next = cont ? folded : next;
            }
            acc = next;
          }
        }
        if constexpr (W == 8) fr_sort8_net(dk, FrCeF64());
        else if constexpr (W == 12) fr_sort12_net(dk, FrCeF64());
        else if constexpr (W == 16) fr_sort16_net(dk, FrCeF64());
        else if constexpr (W == 24) fr_sort24_net(dk, FrCeF64());
        else if constexpr (W == 32) fr_sort32_net(dk, FrCeF64());
        else fr_sortd_sizes<W, 2, W>(dk);
        const int frame = (k < tc) ? (int)rtime[c0 + k] : 0, run = (k < tc) ? (int)rtime[c0 + k + 1] - frame : 0;
        // the records of a rank go where its bucket's entries would (the lattice's piece of the pool has a slot per
        // entry, and a rank has at most as many records as entries): no scan places them, the waves do not meet here
        const int dst = (int)rec_at + nbase + lo;  // (counted from the pool's start: a lattice's pieces take their records where they find room)
        // 8 bytes per record -- the sort key itself, desc(float logp) << 30 | label - lab_base (rec_fmt tells the gather
        // kernel) -- written ONCE per rank; every frame names its records by (first record, count), so the frames up
        // to the next rank share them, and the gather kernel expands the runs into the dense output.
        // What the stores of this pass cost is the NUMBER of their requests, one per lane and instruction where the lanes'
        // addresses lie in different lines: a rank's records go two to a store (190 M records one by one on C2), and the
        // frames' words are written behind the tile's ranks, a lane per frame (lcur, free since the scatter, keeps every
        // rank's number of records until then)
        if (k < tc && !wide) lcur[k] = mk;
        struct __attribute__((packed, aligned(8))) RecordPair {
          uint64_t first, second;
        };
  #pragma unroll
        for (int j = 0; j < W; j += 2) {
          if (j + 1 < mk)
            *reinterpret_cast<RecordPair*>(a.pool + dst + j) = RecordPair{(uint64_t)__double_as_longlong(dk[j]), (uint64_t)__double_as_longlong(dk[j + 1])};
          else if (j < mk)
            a.pool[dst + j] = (uint64_t)__double_as_longlong(dk[j]);
        }
        dense += mk * run;
        stored += mk;
      };
      if (__all(w <= kW0)) rank_round(std::integral_constant<int, kW0>());  // (wave-uniform)
      else if (kW1 < R && __all(w <= kW1)) rank_round(std::integral_constant<int, kW1>());
      else rank_round(std::integral_constant<int, R>());
      if (wide) {
        const int slot = atomicAdd(&wide_n, 1);
        wide_rank[slot] = (uint16_t)k;
      }
    }
    __syncthreads();
    // ---- 3w. the ranks with more than R entries, a wave each: one lane per entry, both sorts as bitonic
    // networks across the lanes, the fold entry by entry in arc order ----
    for (int wi = tid >> 6; wi < wide_n; wi += NT / 64) {  // (wave-uniform)
      const int lane = tid & 63;
      const int k = wide_rank[wi];
      const int lo = lboff[k], w = lboff[k + 1] - lo;  // R < w <= kLnWideMax
      const int dst = (int)rec_at + nbase + lo;
      const int m = w <= 64    ? fr_lean_wide_rank<1>(lbucket + lo, w, a.ilabel + a0, xs, lab_base, abits, lntab, a.pool + dst)  // (wave-uniform)
                    : w <= 128 ? fr_lean_wide_rank<2>(lbucket + lo, w, a.ilabel + a0, xs, lab_base, abits, lntab, a.pool + dst)
                               : fr_lean_wide_rank<4>(lbucket + lo, w, a.ilabel + a0, xs, lab_base, abits, lntab, a.pool + dst);
      const int run = (int)rtime[c0 + k + 1] - (int)rtime[c0 + k];
      if (lane == 0) lcur[k] = m, dense += m * run, stored += m;
    }
    if (wide_n > 0) __syncthreads();  // (workgroup-uniform; the thread path's counts lie behind the barrier above)
    // ---- 4. the frames of the tile's ranks, one run of frames: a thread per frame finds the frame's rank among the rank
    // times and writes its two words (records, first record) -- whole lines, where a thread per rank wrote every frame
    // of its rank one by one ----
    for (int f = (int)rtime[c0] + tid, f_end = (int)rtime[c0 + tc]; f < f_end; f += NT) {
      int j = 0;
#pragma unroll
      for (int step = kTile / 2; step > 0; step >>= 1)
        if (j + step < tc && (int)rtime[c0 + j + step] <= f) j += step;
      a.frame_ref[fbase + f] = make_int2(lcur[j], (int)rec_at + nbase + lboff[j]);
    }
    nbase += n;
    __syncthreads();
    FR_LAP(4);
  }
  for (int d = 32; d > 0; d >>= 1) stored += __shfl_xor(stored, d, 64), dense += __shfl_xor(dense, d, 64);
  if ((tid & 63) == 0) atomicAdd(&sum_stored, stored), atomicAdd(&sum_dense, dense);
  __syncthreads();
  if (tid == 0) {
    if (PIECES) atomicAdd(&a.num_records[u], sum_dense), atomicAdd(&a.num_stored[u], sum_stored);  // (zeroed by the slicing)
    else a.num_records[u] = sum_dense, a.num_stored[u] = sum_stored;
    a.rec_ptr[u] = a.pool;
    a.rec_fmt[u] = ((int64_t)1 << 32) | (int64_t)(uint32_t)lab_base;  // the records are the sort keys themselves
  }
  if (CLOCK && a.phase_clock && tid == 0)
    for (int k = 0; k < 5; ++k) a.phase_clock[(size_t)u * 8 + k] = clk[k];
#undef FR_LAP
}

// Dense output: entries of lattice u at entry_off[u].., frame offsets at frame_base[u]...  A wave
// takes 64 frames at a time: the frames' (first record, count) pairs, a prefix sum of the counts, and
// then one lane per output entry -- its frame found by a binary search over the wave's prefix sums
// (shuffles) -- so the expansion of shared records reads and writes full lines.
__global__ __launch_bounds__(256) void frame_gather_kernel(FrameArgs a, const int64_t* entry_off,
                                                           const int64_t* frame_base, int64_t* frame_off,
                                                           int32_t* labels, float* logps) {
  __shared__ int sh[2 * kFrThreads / 64];
  const int u = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const uint64_t* rec8 = a.rec_ptr[u];
  const int64_t fmt = a.rec_fmt[u];
  const int64_t o = entry_off[u];
  const int64_t fb = frame_base[u];
  const int T = a.num_frames[u];
  int carry = 0;
  for (int base = 0; base < T; base += 256) {
    const int k = base + (int)threadIdx.x;
    const int2 ref = k < T ? a.frame_ref[fb + k] : make_int2(0, 0);
    const int c = ref.x, src = ref.y;
    int tot;
    const int ex = fr_block_scan(c, &tot, sh);  // entries of this block's frames before frame k
    if (k < T) frame_off[fb + k] = o + carry + ex;
    const int first = __shfl(ex, 0, 64);                  // of the wave's 64 frames
    const int count = __shfl(ex + c, 63, 64) - first;     // entries of the wave's frames
    const int64_t out = o + carry + first;
    for (int e0 = 0; e0 < count; e0 += 64) {  // (wave-uniform trips: the shuffles read every lane)
      const int e = e0 + lane;
      int g = 0;
#pragma unroll
      for (int step = 32; step > 0; step >>= 1) {
        const int v = __shfl(ex, min(g + step, 63), 64) - first;
        if (v <= e) g += step;  // the LAST frame that starts at or before e: empty frames before it share its start
      }
      const int at = __shfl(src, g, 64) + (e - (__shfl(ex, g, 64) - first));
      if (e < count) {
        const uint64_t r = rec8[at];
        if (fmt >> 32) {  // the lean kernel's records: desc(float logp) << 30 | label - lab_base
          labels[out + e] = (int32_t)(uint32_t)(r & 0x3FFFFFFFull) + (int32_t)(uint32_t)fmt;
          logps[out + e] = fr_desc_unkey((uint32_t)(r >> 30));
        } else {
          labels[out + e] = fr_unbias((uint32_t)r);
          logps[out + e] = fr_desc_unkey((uint32_t)(r >> 32));
        }
      }
    }
    carry += tot;
  }
}

hipError_t launch_frame_count(const FrameArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(frame_count_kernel, dim3(num_lattices), dim3(256), 0, stream, a);
  return hipGetLastError();
}
static int frame_span_cap(int max_arcs) { return max_arcs < kFrLdsSpans ? (max_arcs > 0 ? max_arcs : 0) : kFrLdsSpans; }
// the general kernel: every lattice, or (a.redo_list set) the lattices the lean kernel listed
hipError_t launch_frame(const FrameArgs& a, int num_lattices, int max_arcs, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  static_assert(sizeof(FKey) * kFrTile <= sizeof(uint16_t) * kFrLdsEntries, "the sort tile lies under the buckets");
  const int span_cap = frame_span_cap(max_arcs);
  const size_t lds = sizeof(uint16_t) * kFrLdsEntries + sizeof(uint32_t) * (size_t)span_cap;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&frame_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(frame_kernel, dim3(num_lattices), dim3(kFrThreads), lds, stream, a, span_cap);
  return hipGetLastError();
}
// the lean kernel: needs neither scratch_off nor the scratch rows, takes its records from a.pool.  big: its second
// instantiation over the lattices the first one listed (a.redo_list); it lists what it leaves in a.redo2_list
// Pieces of a long lattice (FrameItem, klat_accum.h): one workgroup per lattice.
__global__ __launch_bounds__(1024) void frame_slice_kernel(FrameArgs a, const int32_t* long_list) {
  constexpr int kMaxPieces = 1024;
  __shared__ int bad, longest, lab_min, own_lo[kMaxPieces + 1], own_t[kMaxPieces + 1];  // a piece's first own state, its window's first frame
  __shared__ int32_t before[kSliceBlocks + 1], from[kSliceBlocks + 1];
  const int u = long_list[blockIdx.x], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t a0 = a.arc_off[u], s0 = a.state_off[u];
  const int nA = (int)(a.arc_off[u + 1] - a0), S = (int)(a.state_off[u + 1] - s0);
  const int T = a.num_frames[u];
  const int32_t* src = a.src + a0;
  const int32_t* dst = a.dst + a0;
  const int32_t* tm = a.state_time + s0;
  const int first = a.piece_first[u], pieces = (S + kFramePieceStates - 1) / kFramePieceStates;
  FrameItem* items = a.piece_items + first;
  if (tid == 0) bad = (pieces > kMaxPieces || S > 64 * kSliceBlocks) ? 1 : 0, longest = 0, lab_min = 0x7FFFFFFF;
  __syncthreads();
  // a cut between two states separates the frames where every state before it is earlier than every state behind it
  // (klat_slice.h: times that descend here and there still leave such places)
  auto time_of = [&](int s) { return tm[s]; };
  int span = 0, lmin = 0x7FFFFFFF;
  for (int i = tid; i < nA; i += 1024) {
    span = max(span, dst[i] - src[i]);
    const int l = a.ilabel[a0 + i];
    if (l != 0) lmin = min(lmin, l);
  }
  for (int d = 32; d > 0; d >>= 1) span = max(span, __shfl_xor(span, d, 64)), lmin = min(lmin, __shfl_xor(lmin, d, 64));
  if (lane == 0) atomicMax(&longest, span), atomicMin(&lab_min, lmin);
  if (!bad) slice_prepare(S, time_of, before, from);  // (workgroup-uniform)
  // the first state of piece k: the first good cut at or behind k * kFramePieceStates
  for (int k = 1 + wave; !bad && k < pieces; k += 16) {  // (wave-uniform)
    const int target = k * kFramePieceStates;  // < S
    int32_t earliest = 0;  // the earliest time of a state from the cut on: where the piece's window begins
    const int cut = slice_find(S, time_of, before, from, target, min(target + kFramePieceWindow - 1, S - 1), false, &earliest);
    if (cut < 0) bad = 1;
    if (lane == 0) own_lo[k] = cut, own_t[k] = earliest;
  }
  if (tid == 0) own_lo[0] = 0, own_lo[pieces <= kMaxPieces ? pieces : 0] = S;
  __syncthreads();
  const int L = longest;
  if (L > kFramePieceWindow) bad = 1;
  __syncthreads();
  auto first_arc_from = [&](int state) {  // first arc whose source is `state` or later (the arcs are grouped by ascending source)
    int lo = 0, hi = nA;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (src[mid] < state) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  FrameItem mine = {-1, 0, 0, 0, 0, 0, 0, 0, 0};  // (at most kMaxPieces = 1 024 pieces: one a thread)
  if (!bad && tid < pieces) {
    const int k = tid;
    FrameItem it;
    const int lo = own_lo[k], hi = own_lo[k + 1];
    it.u = u;
    it.s_lo = max(lo - L, 0), it.s_hi = min(hi + L, S);
    it.a_lo = first_arc_from(it.s_lo), it.a_hi = first_arc_from(hi);
    it.f0 = k == 0 ? 0 : min(max(own_t[k], 0), T);
    it.f1 = hi >= S ? T : min(max(own_t[k + 1], 0), T);
    it.lab_base = lab_min;
    const int arcs = it.a_hi - it.a_lo;
    if (arcs > 65535 || it.s_hi - it.s_lo > kLnStates || it.f1 - it.f0 > 65535) bad = 1;
    const unsigned long long at = atomicAdd(a.piece_scratch_used, (unsigned long long)arcs);
    if (at + (unsigned long long)arcs > (unsigned long long)a.piece_scratch_cap) bad = 1;
    it.scratch = (int64_t)at;
    mine = it;
  }
  __syncthreads();
  const bool ok = bad == 0;  // (set by plain stores of 1 from any thread: read behind the barrier)
  if (tid < pieces) {
    if (!ok) mine.u = -1;
    items[tid] = mine;
  }
  if (tid == 0) {
    if (ok) {
      a.num_records[u] = 0, a.num_stored[u] = 0;
    } else if ((atomicOr(&a.listed[u], 2) & 2) == 0) {  // the general kernel's
      a.redo2_list[atomicAdd(a.redo2_n, 1)] = u;
    }
  }
}

hipError_t launch_frame_slice(const FrameArgs& a, const int32_t* long_list, int count, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(frame_slice_kernel, dim3(count), dim3(1024), 0, stream, a, long_list);
  return hipGetLastError();
}

// The dense pass (thirty-two registers a rank) over the listed lattices (at most num_lattices of them) and the marked pieces.
hipError_t launch_frame_lean_dense(const FrameArgs& a, int num_lattices, int num_slots, hipStream_t stream) {
  const int span_cap = kFrLdsSpans;
  const size_t lds = std::max(sizeof(uint16_t) * kFrLdsEntries + sizeof(uint16_t) * (kLnStatesLds + 2) + sizeof(uint32_t) * (size_t)span_cap,
                              sizeof(uint16_t) * kFrLdsEntries + sizeof(uint16_t) * (kLnStates + 2));
  if (num_lattices >= 2048) {  // (eight lattices a CU and more)
    auto lean = a.phase_clock ? &frame_lean_kernel<true, true, 1, false, 2 * kFrRegs, 2> : &frame_lean_kernel<true, false, 1, false, 2 * kFrRegs, 2>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lean), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lean, dim3(num_lattices), dim3(kFrLeanThreads / 2), lds, stream, a, span_cap);
  } else if (num_lattices > 0) {
    auto lean = a.phase_clock ? &frame_lean_kernel<true, true, 1, false, 2 * kFrRegs, 1> : &frame_lean_kernel<true, false, 1, false, 2 * kFrRegs, 1>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lean), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lean, dim3(num_lattices), dim3(kFrLeanThreads), lds, stream, a, span_cap);
  }
  if (num_slots > 0) {  // (how many of the pieces are dense only the device knows: workgroups of 512)
    auto lean = a.phase_clock ? &frame_lean_kernel<true, true, 1, true, 2 * kFrRegs, 1> : &frame_lean_kernel<true, false, 1, true, 2 * kFrRegs, 1>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lean), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lean, dim3(num_slots), dim3(kFrLeanThreads), lds, stream, a, span_cap);
  }
  return hipGetLastError();
}

hipError_t launch_frame_lean_pieces(const FrameArgs& a, int num_slots, hipStream_t stream) {
  if (num_slots <= 0) return hipSuccess;
  const int span_cap = kFrLdsSpans;
  const size_t lds = std::max(sizeof(uint16_t) * kFrLdsEntries + sizeof(uint16_t) * (kLnStatesLds + 2) + sizeof(uint32_t) * (size_t)span_cap,
                              sizeof(uint16_t) * kFrLdsEntries + sizeof(uint16_t) * (kLnStates + 2));
  auto lean = a.phase_clock ? &frame_lean_kernel<true, true, 1, true> : &frame_lean_kernel<true, false, 1, true>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lean), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lean, dim3(num_slots), dim3(kFrLeanThreads), lds, stream, a, span_cap);
  return hipGetLastError();
}

template <bool BIG, int SC>
static hipError_t launch_frame_lean_class(const FrameArgs& a, int num_lattices, int max_arcs, hipStream_t stream) {
  const int span_cap = std::min(frame_span_cap(max_arcs), kFrLdsSpans / SC);
  // buckets, rank times, spans; a lattice whose spans lie in L2 has up to kLnStates + 1 rank times where the spans would be
  size_t lds = sizeof(uint16_t) * (kFrLdsEntries / SC) + sizeof(uint16_t) * (kLnStatesLds / SC + 2) + sizeof(uint32_t) * (size_t)span_cap;
  if (BIG) lds = std::max(lds, sizeof(uint16_t) * kFrLdsEntries + sizeof(uint16_t) * (kLnStates + 2));
  auto lean = a.phase_clock ? &frame_lean_kernel<BIG, true, SC> : &frame_lean_kernel<BIG, false, SC>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lean), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lean, dim3(num_lattices), dim3(kFrLeanThreads / SC), lds, stream, a, span_cap);
  return hipGetLastError();
}
hipError_t launch_frame_lean(const FrameArgs& a, int num_lattices, int max_arcs, bool big, int size_class, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  static_assert(sizeof(uint16_t) * kLnStates + 2048 * 6 <= sizeof(uint16_t) * kFrLdsEntries, "the states' ranks and the frame bits lie under the buckets");
  if (big) return launch_frame_lean_class<true, 1>(a, num_lattices, max_arcs, stream);
  switch (size_class) {  // kFrameClassArcs / kFrameClassStates (klat_accum.h)
    case 0: return launch_frame_lean_class<false, 4>(a, num_lattices, max_arcs, stream);
    case 1: return launch_frame_lean_class<false, 2>(a, num_lattices, max_arcs, stream);
    default: return launch_frame_lean_class<false, 1>(a, num_lattices, max_arcs, stream);
  }
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
