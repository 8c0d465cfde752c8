// klat_pack_device.hip -- the lockstep arc streams built on the device from the CSR batch already
// in HBM.  Same streams, bit for bit, as the host packer (klat_pack.cc), which stays as the
// reference of the stream format for the CPU-side tests; klat_batch_create uses this one.
//
//   pack_count_kernel  one workgroup per lattice, the lattice's tables in LDS: validates the
//                      topology, counts in- and out-degrees (LDS atomics), sorts the arcs by
//                      destination (counting sort; sources ascending inside a destination), counts
//                      the rounds its forward and its backward stream would take with 1, 2 ... 64
//                      lanes, and leaves three tables in HBM as full lines: first in-arc per state,
//                      first out-arc per state, arcs by destination.  The round counts go back to
//                      the host, which picks every lattice's lanes and slices the lattices into jobs
//                      (plan_layout, build_jobs in klat_pack.cc).  A lattice whose tables exceed the
//                      LDS of a CU takes pack_count_global_kernel (same results, tables in L2).
//   pack_rounds_kernel one workgroup per lattice: prefix sums of the rounds every state takes in
//                      the forward and in the backward stream with the lanes the plan gave it.
//   pack_fill_kernel   one workgroup per (job, run of tiles).  A tile is 32 rounds of the job's
//                      stored lanes: one contiguous piece of each of the three stream arrays.  The
//                      workgroup assembles the tile in LDS -- every state that has a round in the
//                      tile writes its term words and costs (ring cells, BYPASS / ONE / FAR / WORD,
//                      FIRST / LAST / COMMIT) and the padding of its unused lanes -- and stores it
//                      as whole lines (dwordx4 per lane).  Written term by term from registers the
//                      same streams cost 5.4 GB of HBM writes for 2 GB of stream on C2.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_math.h"
#include "klat_pack_device.h"
#include "klat_pack_common.h"

namespace klat {

namespace {

constexpr int kT = 256;       // pack_rounds_kernel, pack_fill_kernel
constexpr int kTG = 1024;     // pack_count_global_kernel: its passes are chains of L2 atomics and gathers, one per thread and turn;
                              // with 256 threads the few lattices above the LDS classes took 4.4 ms of the C5/8 shard's count pass
constexpr int kTC = 512;      // pack_count_kernel


__device__ __forceinline__ int lower_bound_src(const int32_t* src, int n, int s) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (src[mid] < s)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// exclusive prefix sums of three ints per thread with one pair of barriers
template <int T>
__device__ void block_scan3(const int v[3], int ex[3], int tot[3], int* sh /* [3 * T/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl[3] = {v[0], v[1], v[2]};
  for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int t = __shfl_up(incl[c], d, 64);
      if (lane >= d) incl[c] += t;
    }
  }
  __syncthreads();  // sh may still be read from a previous call
  if (lane == 63) {
#pragma unroll
    for (int c = 0; c < 3; ++c) sh[c * (T / 64) + wave] = incl[c];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int off = 0, t = 0;
    for (int w = 0; w < T / 64; ++w) {
      const int x = sh[c * (T / 64) + w];
      if (w < wave) off += x;
      t += x;
    }
    tot[c] = t;
    ex[c] = off + incl[c] - v[c];
  }
}

// The lattice's tables in LDS; arc numbers as 16-bit words (a lattice of 65 536 arcs or more takes
// the global kernel: pack_count_lds_bytes puts it above every class).  The passes over the arcs
// keep kCountUnroll loads per thread in flight (one after the other, twelve dependent round trips
// per pass were most of the kernel).
typedef uint16_t IA;
#ifndef KLAT_COUNT_UNROLL
#define KLAT_COUNT_UNROLL 4
#endif
constexpr int kCountUnroll = KLAT_COUNT_UNROLL;
// (six waves a SIMD: 80 VGPRs with 14 spilled where the compiler took 92 -- the LDS classes allow three and four workgroups a
// CU, 92 registers allowed two: 0.69 -> 0.60 ms on C2)
__global__ __launch_bounds__(kTC, 6) void pack_count_kernel(PackDevArgs a, int lds_lo, int lds_hi) {
  extern __shared__ int32_t lds[];
  __shared__ int sh[3 * (kTC / 64)];
  __shared__ int sh_bad, sh_mr, sh_mtf, sh_mtb, sh_linked, sh_span;
  __shared__ int sh_rounds[2 * kNumLaneChoices];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
  const int need = pack_count_lds_bytes(S, A);
  // every lattice belongs to exactly one launch: the narrowest LDS class it fits
  if (need <= lds_lo || need > lds_hi || (a.only && !a.only[u])) return;
  if (tid < (int)(sizeof(LatticeRounds) / sizeof(int32_t))) reinterpret_cast<int32_t*>(&a.lat_rounds[u])[tid] = 0;
  if (S <= 0) {
    if (tid == 0 && (S < 0 || A != 0)) report_bad(a, 1 /* KLAT_ERR_INVALID_ARG */, u);
    return;
  }
  int32_t* cnt = lds;              // [S + 2] in-degrees, then their prefix
  int32_t* row = lds + (S + 2);    // [S + 2] out-degrees, then their prefix
  IA* in_arc = reinterpret_cast<IA*>(lds + 2 * (S + 2));  // [A]
  const int64_t sb = s0 + 2 * (int64_t)u;
  const int32_t* src = a.src + a0;
  const int32_t* dst = a.dst + a0;
  // diagnostic (KLAT_PACK_PROFILE=1): shader cycles of thread 0 per phase, summed over the workgroups
  unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = a.prof ? __builtin_readcyclecounter() : 0;
  auto stamp = [&](int phase) {
    if (a.prof) {
      const unsigned long long now = __builtin_readcyclecounter();
      pc[phase] += now - pt;
      pt = now;
    }
  };
  if (tid == 0) sh_bad = 0, sh_mr = 1, sh_mtf = 1, sh_mtb = 1, sh_linked = 0, sh_span = 1;
  if (tid < 2 * kNumLaneChoices) sh_rounds[tid] = 0;
  for (int s = tid; s < 2 * (S + 2); s += kTC) lds[s] = 0;
  __syncthreads();
  stamp(0);
  // degrees + validation (the host packer's checks: ids in range, arcs grouped by source, every
  // arc goes to a higher state)
  int mr = 1;
  for (int ib = tid; ib < A; ib += kTC * kCountUnroll) {
    int vs[kCountUnroll], vt[kCountUnroll], vp[kCountUnroll];
#pragma unroll
    for (int x = 0; x < kCountUnroll; ++x) {
      const int i = min(ib + x * kTC, A - 1);
      vs[x] = src[i], vt[x] = dst[i], vp[x] = i > 0 ? src[i - 1] : 0;
    }
#pragma unroll
    for (int x = 0; x < kCountUnroll; ++x) {
      if (ib + x * kTC >= A) continue;
      const int s = vs[x], t = vt[x];
      if (s < 0 || s >= S || s < vp[x]) {
        atomicMax(&sh_bad, 2);  // malformed CSR
      } else if (t <= s || t >= S) {
        atomicMax(&sh_bad, 1);  // not topologically sorted
      } else {
        atomicAdd(&cnt[t], 1);
        atomicAdd(&row[s], 1);
        mr = max(mr, t - s);
      }
    }
  }
  {  // the longest arc alone (the final states join mr below)
    int span = mr;
    for (int d = 32; d > 0; d >>= 1) span = max(span, __shfl_xor(span, d, 64));
    if ((tid & 63) == 0) atomicMax(&sh_span, span);
  }
  __syncthreads();
  stamp(1);
  if (sh_bad) {
    if (tid == 0) report_bad(a, sh_bad == 2 ? 1 /* INVALID_ARG */ : 2 /* BAD_TOPOLOGY */, u);
    return;
  }
  // per state: rank among the final states (the terms of the forward stream's last position), first
  // out-arc (row table), first in-arc; the rounds of every position for every number of lanes --
  // forward position p holds the in-arcs of state p (+ the constant-one term of the start state),
  // backward position S-1-s the out-arcs of s (+ its final weight)
  int fsum[kNumLaneChoices] = {0}, bsum[kNumLaneChoices] = {0};
  int carry[3] = {0, 0, 0}, mtf = 1, mtb = 1;
  for (int base = 0; base <= S; base += kTC) {
    const int s = base + tid;
    int v[3] = {0, 0, 0}, ex[3], tot[3];
    if (s < S) {
      const float fg = a.final_graph[s0 + s], fa = a.final_acoustic[s0 + s];
      v[0] = !(fg == INFINITY && fa == INFINITY) ? 1 : 0;
      v[1] = row[s];
      v[2] = cnt[s];
    }
    block_scan3<kTC>(v, ex, tot, sh);
    if (s < S) {
      if (v[0]) {
        a.flist[s0 + carry[0] + ex[0]] = s;  // the final states in state order
        mr = max(mr, S - s);
      }
      const int nb = v[1] + v[0], nf = v[2] + (s == 0 ? 1 : 0);
      mtb = max(mtb, nb);
      mtf = max(mtf, nf);
      for (int k = 0; k < kNumLaneChoices; ++k) {
        bsum[k] += (max(nb, 1) + (1 << k) - 1) >> k;
        fsum[k] += (max(nf, 1) + (1 << k) - 1) >> k;
      }
    }
    if (s <= S) {
      row[s] = carry[1] + ex[1];
      cnt[s] = carry[2] + ex[2];
      a.row[sb + s] = carry[1] + ex[1];
      a.cnt[sb + s] = carry[2] + ex[2];
    }
    carry[0] += tot[0], carry[1] += tot[1], carry[2] += tot[2];
  }
  const int nfinal = carry[0];
  if (tid == 0) {
    a.cnt[sb + S + 1] = A + nfinal;  // position S holds one term per final state: "in-arcs" A .. A + nfinal - 1
    mtf = max(mtf, nfinal);
    for (int k = 0; k < kNumLaneChoices; ++k) fsum[k] += (max(nfinal, 1) + (1 << k) - 1) >> k;
  }
  for (int d = 32; d > 0; d >>= 1) mr = max(mr, __shfl_xor(mr, d, 64));
  if ((tid & 63) == 0) atomicMax(&sh_mr, mr);
  __syncthreads();
  stamp(2);
  // stable counting sort of the arcs by destination: scatter (cnt[t] moves from the first entry of
  // bucket t to the first entry of bucket t + 1), then order every bucket by arc number
  for (int ib = tid; ib < A; ib += kTC * kCountUnroll) {
    int vt[kCountUnroll];
#pragma unroll
    for (int x = 0; x < kCountUnroll; ++x) vt[x] = dst[min(ib + x * kTC, A - 1)];
#pragma unroll
    for (int x = 0; x < kCountUnroll; ++x)
      if (ib + x * kTC < A) in_arc[atomicAdd(&cnt[vt[x]], 1)] = (IA)(ib + x * kTC);
  }
  __syncthreads();
  stamp(3);
  for (int s = tid; s < S; s += kTC) {
    const int b = s > 0 ? cnt[s - 1] : 0, e = cnt[s], n = e - b;
    if (n <= 1) continue;
    if (n <= 8) {  // in registers: a sorting network on up to eight arc numbers (up to four: one wave in two had a lane with more, and waited for its insertion sort)
      int v[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) v[x] = x < n ? (int)in_arc[b + x] : 0x10000 + x;
      auto cswap = [&](int x, int y) {
        const int lo = min(v[x], v[y]), hi = max(v[x], v[y]);
        v[x] = lo, v[y] = hi;
      };
      cswap(0, 1), cswap(2, 3), cswap(4, 5), cswap(6, 7);
      cswap(0, 2), cswap(1, 3), cswap(4, 6), cswap(5, 7);
      cswap(1, 2), cswap(5, 6);
      cswap(0, 4), cswap(1, 5), cswap(2, 6), cswap(3, 7);
      cswap(2, 4), cswap(3, 5);
      cswap(1, 2), cswap(3, 4), cswap(5, 6);
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (x < n) in_arc[b + x] = (IA)v[x];
      continue;
    }
    for (int i = b + 1; i < e; ++i) {
      const IA v = in_arc[i];
      int j = i - 1;
      while (j >= b && in_arc[j] > v) {
        in_arc[j + 1] = in_arc[j];
        --j;
      }
      in_arc[j + 1] = v;
    }
  }
  __syncthreads();
  stamp(4);
  for (int i = tid; i < A; i += kTC) a.in_arc[a0 + i] = (int32_t)in_arc[i];
  stamp(5);
  // the out-arcs of every state in stream order: farthest destination first, nearest last (the term
  // that depends on the most recently finished state enters the accumulator last), equal
  // destinations in arc order.  Up to four arcs in registers; more through the table in HBM.
  int linked = 0;  // states with an arc to their successor (the level planner's test for a time-ordered chain)
  for (int s = tid; s < S; s += kTC) {
    const int b = row[s], e = row[s + 1], deg = e - b;
    if (deg <= 0) continue;
    if (deg <= 8) {
      // one key an arc: (S - destination) << 3 | its place among the state's arcs -- ascending is destination descending,
      // arc ascending; the slots behind the last arc sort to the end
      uint32_t k[8];
      int nearest = S;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int dv = x < deg ? dst[b + x] : S;
        if (x < deg) nearest = min(nearest, dv);
        k[x] = x < deg ? ((uint32_t)(S - dv) << 3) | (uint32_t)x : 0xFFFFFFFFu;
      }
      auto cswap = [&](int x, int y) {
        const uint32_t lo = min(k[x], k[y]), hi = max(k[x], k[y]);
        k[x] = lo, k[y] = hi;
      };
      cswap(0, 1), cswap(2, 3), cswap(4, 5), cswap(6, 7);
      cswap(0, 2), cswap(1, 3), cswap(4, 6), cswap(5, 7);
      cswap(1, 2), cswap(5, 6);
      cswap(0, 4), cswap(1, 5), cswap(2, 6), cswap(3, 7);
      cswap(2, 4), cswap(3, 5);
      cswap(1, 2), cswap(3, 4), cswap(5, 6);
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (x < deg) a.out_arc[a0 + b + x] = b + (int32_t)(k[x] & 7u);
      linked += nearest == s + 1 ? 1 : 0;
    } else {
      int32_t* out_arc = a.out_arc + a0;
      int nearest = S;
      for (int i = b; i < e; ++i) {
        const int dv = dst[i];
        nearest = min(nearest, dv);
        int j = i - 1;
        while (j >= b && dst[out_arc[j]] < dv) {
          out_arc[j + 1] = out_arc[j];
          --j;
        }
        out_arc[j + 1] = i;
      }
      linked += nearest == s + 1 ? 1 : 0;
    }
  }
  stamp(6);
  // (reduced per wave first: 512 threads adding to the same 14 words one after the other were half the kernel)
  for (int d = 32; d > 0; d >>= 1) {
    mtf = max(mtf, __shfl_xor(mtf, d, 64));
    mtb = max(mtb, __shfl_xor(mtb, d, 64));
#pragma unroll
    for (int k = 0; k < kNumLaneChoices; ++k) fsum[k] += __shfl_xor(fsum[k], d, 64), bsum[k] += __shfl_xor(bsum[k], d, 64);
  }
  for (int d = 32; d > 0; d >>= 1) linked += __shfl_xor(linked, d, 64);
  if ((tid & 63) == 0) {
    atomicMax(&sh_mtf, mtf);
    atomicMax(&sh_mtb, mtb);
    atomicAdd(&sh_linked, linked);
    for (int k = 0; k < kNumLaneChoices; ++k) {
      atomicAdd(&sh_rounds[k], fsum[k]);
      atomicAdd(&sh_rounds[kNumLaneChoices + k], bsum[k]);
    }
  }
  __syncthreads();
  if (tid < kNumLaneChoices) {
    a.lat_rounds[u].fwd[tid] = sh_rounds[tid];
    a.lat_rounds[u].bwd[tid] = sh_rounds[kNumLaneChoices + tid];
  }
  stamp(7);
  if (a.prof && tid == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&a.prof[i], pc[i]);
  if (tid == 0) {
    a.row[sb + S + 1] = sh_linked;  // (the spare entry of the row table)
    atomicMax(&a.status[2], sh_mr);
    a.lat_rounds[u].max_rel = sh_mr;
    a.lat_rounds[u].max_terms = min(sh_mtf, 64) | (min(sh_mtb, 64) << 16);
    a.lat_rounds[u].max_span = sh_span;
  }
}

// The same pass with the lattice's tables in its scratch rows in HBM (L2 atomics): for the lattices
// above `lds_hi` bytes of tables.
// (Its counters -- in-degrees, then the counting sort's cursors -- in LDS all the same where 4 (S + 2) bytes fit, up to 38 000
// states: the two passes of L2 atomics were half of this kernel's time.)
__global__ __launch_bounds__(kTG) void pack_count_global_kernel(PackDevArgs a, int lds_hi, int lds_counters) {
  extern __shared__ int32_t lcnt[];  // [S + 2] where it fits
  __shared__ int sh[kTG / 64];
  __shared__ int sh_bad, sh_mr, sh_mtf, sh_mtb, sh_linked, sh_span;
  __shared__ int sh_rounds[2 * kNumLaneChoices];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0), A = (int)(a.arc_off[u + 1] - a0);
  if (pack_count_lds_bytes(S, A) <= lds_hi || (a.only && !a.only[u])) return;
  if (tid < (int)(sizeof(LatticeRounds) / sizeof(int32_t))) reinterpret_cast<int32_t*>(&a.lat_rounds[u])[tid] = 0;
  const int64_t sb = s0 + 2 * (int64_t)u;
  int32_t* cnt = a.cnt + sb;
  int32_t* row = a.row + sb;
  const bool in_lds = (int64_t)sizeof(int32_t) * (S + 2) <= (int64_t)lds_counters;  // (workgroup-uniform)
  int32_t* cur = in_lds ? lcnt : a.r0b + sb;  // (the counting sort's cursors: LDS, or scratch in L2)
  int32_t* deg = in_lds ? lcnt : cnt;         // the in-degrees while they are counted
  int32_t* flist = a.flist + s0;
  int32_t* in_arc = a.in_arc + a0;
  int32_t* out_arc = a.out_arc + a0;
  const int32_t* src = a.src + a0;
  const int32_t* dst = a.dst + a0;
  if (tid == 0) sh_bad = 0, sh_mr = 1, sh_mtf = 1, sh_mtb = 1, sh_linked = 0, sh_span = 1;
  if (tid < 2 * kNumLaneChoices) sh_rounds[tid] = 0;
  for (int s = tid; s < S + 2; s += kTG) deg[s] = 0;
  __syncthreads();
  int mr = 1;
  constexpr int kAhead = 8;  // arcs a thread has in flight (one after the other, a thread's hundred trips to L2 were the pass)
  for (int ib = tid; ib < A; ib += kTG * kAhead) {
    int vs[kAhead], vt[kAhead], vp[kAhead];
#pragma unroll
    for (int x = 0; x < kAhead; ++x) {
      const int i = min(ib + x * kTG, A - 1);
      vs[x] = src[i], vt[x] = dst[i], vp[x] = i > 0 ? src[i - 1] : 0;
    }
#pragma unroll
    for (int x = 0; x < kAhead; ++x) {
      const int i = ib + x * kTG;
      if (i >= A) continue;
      const int s = vs[x], t = vt[x];
      if (s < 0 || s >= S || s < vp[x]) {
        atomicMax(&sh_bad, 2);
      } else if (t <= s || t >= S) {
        atomicMax(&sh_bad, 1);
      } else {
        atomicAdd(&deg[t], 1);
        mr = max(mr, t - s);
        // the row table from the arcs themselves (grouped by ascending source): arc i is the first out-arc of every
        // state behind its predecessor's source up to its own -- no search per state.  (The predecessor may be the
        // malformed arc this pass exists to reject: its source below zero must not move the fill's start before the
        // lattice's own rows; one above S leaves nothing to fill.)
        for (int q = (i > 0 ? max(vp[x], -1) + 1 : 0); q <= s; ++q) row[q] = i;
      }
    }
  }
  atomicMax(&sh_span, mr);
  for (int s = tid; s < S; s += kTG)
    if (dev_is_final(a, s0 + s)) mr = max(mr, S - s);
  atomicMax(&sh_mr, mr);
  __syncthreads();
  if (sh_bad) {
    if (tid == 0) report_bad(a, sh_bad == 2 ? 1 : 2, u);
    return;
  }
  if (tid == 0) atomicMax(&a.status[2], sh_mr);
  int carry = 0;
  for (int base = 0; base < S; base += kTG) {
    const int s = base + tid;
    const int f = (s < S && dev_is_final(a, s0 + s)) ? 1 : 0;
    int tot;
    const int ex = block_scan<kTG>(f, &tot, sh);
    if (f) flist[carry + ex] = s;
    carry += tot;
  }
  const int nfinal = carry;
  int fsum[kNumLaneChoices] = {0}, bsum[kNumLaneChoices] = {0};
  int carry_c = 0, mtf = 1, mtb = 1;
  for (int base = 0; base <= S; base += kTG) {
    const int p = base + tid;
    const int c = (p < S) ? deg[p] : 0;
    const int nterm = (p < S) ? c + (p == 0 ? 1 : 0) : nfinal;
    if (p <= S) {
      mtf = max(mtf, nterm);
      for (int k = 0; k < kNumLaneChoices; ++k) fsum[k] += (max(nterm, 1) + (1 << k) - 1) >> k;
    }
    int tot_c;
    const int ex_c = block_scan<kTG>(c, &tot_c, sh);
    if (p <= S) {
      cnt[p] = carry_c + ex_c;
      if (in_lds) lcnt[p] = carry_c + ex_c;  // (the cursors start at the buckets' first entries)
    }
    carry_c += tot_c;
  }
  if (tid == 0) cnt[S + 1] = A + nfinal;
  __syncthreads();
  if (!in_lds)
    for (int s = tid; s < S; s += kTG) cur[s] = cnt[s];
  for (int s = (A > 0 ? src[A - 1] + 1 : 0) + tid; s <= S; s += kTG) row[s] = A;  // (the states behind the last arc's source)
  __syncthreads();
  for (int ib = tid; ib < A; ib += kTG * kAhead) {
    int vt[kAhead], at[kAhead];
#pragma unroll
    for (int x = 0; x < kAhead; ++x) vt[x] = dst[min(ib + x * kTG, A - 1)];
#pragma unroll
    for (int x = 0; x < kAhead; ++x) at[x] = ib + x * kTG < A ? atomicAdd(&cur[vt[x]], 1) : 0;
#pragma unroll
    for (int x = 0; x < kAhead; ++x)
      if (ib + x * kTG < A) in_arc[at[x]] = ib + x * kTG;
  }
  __syncthreads();
  // sixteen keys in registers, ascending (the 60-comparator network of klat_frame.hip): a state's in-arcs / out-arcs are
  // loaded together, ordered and written back -- insertion sorts in place were chains of dependent trips to L2, twenty
  // states a thread, and a wave waits for the lane with the longest
  auto sort16 = [](uint32_t (&k)[16]) {
    auto ce = [](uint32_t& x, uint32_t& y) {
      const uint32_t lo = min(x, y), hi = max(x, y);
      x = lo, y = hi;
    };
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
  };
  for (int s = tid; s < S; s += kTG) {
    const int b = cnt[s], e = cnt[s + 1];
    const int r0 = row[s], r1 = row[s + 1];
    if (e - b <= 16) {
      uint32_t k[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) k[j] = b + j < e ? (uint32_t)in_arc[b + j] : 0xFFFFFFFFu;
      sort16(k);
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (b + j < e) in_arc[b + j] = (int32_t)k[j];
    } else {
      for (int i = b + 1; i < e; ++i) {
        const int v = in_arc[i];
        int j = i - 1;
        while (j >= b && in_arc[j] > v) {
          in_arc[j + 1] = in_arc[j];
          --j;
        }
        in_arc[j + 1] = v;
      }
    }
    const int nb = r1 - r0 + (dev_is_final(a, s0 + s) ? 1 : 0);
    mtb = max(mtb, nb);
    for (int k = 0; k < kNumLaneChoices; ++k) bsum[k] += (max(nb, 1) + (1 << k) - 1) >> k;
    int nearest = S;
    if (r1 - r0 <= 16 && S < (1 << 27)) {  // out-arcs: farthest destination first, equal destinations in arc order
      uint32_t k[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int dv = r0 + j < r1 ? dst[r0 + j] : S;
        if (r0 + j < r1) nearest = min(nearest, dv);
        k[j] = r0 + j < r1 ? ((uint32_t)(S - dv) << 4) | (uint32_t)j : 0xFFFFFFFFu;
      }
      sort16(k);
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (r0 + j < r1) out_arc[r0 + j] = r0 + (int32_t)(k[j] & 15u);
    } else {
      for (int i = r0; i < r1; ++i) {
        const int dv = dst[i];
        nearest = min(nearest, dv);
        int j = i - 1;
        while (j >= r0 && dst[out_arc[j]] < dv) {
          out_arc[j + 1] = out_arc[j];
          --j;
        }
        out_arc[j + 1] = i;
      }
    }
    if (nearest == s + 1) atomicAdd(&sh_linked, 1);
  }
  atomicMax(&sh_mtf, mtf);
  atomicMax(&sh_mtb, mtb);
  for (int k = 0; k < kNumLaneChoices; ++k) {
    if (fsum[k]) atomicAdd(&sh_rounds[k], fsum[k]);
    if (bsum[k]) atomicAdd(&sh_rounds[kNumLaneChoices + k], bsum[k]);
  }
  __syncthreads();
  if (tid < kNumLaneChoices) {
    a.lat_rounds[u].fwd[tid] = sh_rounds[tid];
    a.lat_rounds[u].bwd[tid] = sh_rounds[kNumLaneChoices + tid];
  }
  if (tid == 0) {
    row[S + 1] = sh_linked;  // (the spare entry of the row table: the level planner's chain test)
    a.lat_rounds[u].max_rel = sh_mr;
    a.lat_rounds[u].max_terms = min(sh_mtf, 64) | (min(sh_mtb, 64) << 16);
    a.lat_rounds[u].max_span = sh_span;
  }
}

// Fused costs: the largest finite cost of the batch (the residuals' unit follows from it).  Positive doubles
// order like their bit patterns.
__global__ __launch_bounds__(kT) void pack_maxcost_kernel(PackDevArgs a, int64_t num_arcs, int64_t num_states) {
  float m = 0.0f;
  // four arcs a lane and load (the arrays start on 256-byte boundaries), the last few one by one
  const int64_t quads = num_arcs / 4;
  for (int64_t q = (int64_t)blockIdx.x * kT + threadIdx.x; q < quads; q += (int64_t)gridDim.x * kT) {
    const float4 g = reinterpret_cast<const float4*>(a.graph)[q], ac = reinterpret_cast<const float4*>(a.acoustic)[q];
    int4 w = {0, 0, 0, 0};
    if (a.fsp.penalized) w = reinterpret_cast<const int4*>(a.ilabel)[q];
    const float c0 = arc_cost_abs_f(g.x, ac.x, w.x != 0, a.fsp), c1 = arc_cost_abs_f(g.y, ac.y, w.y != 0, a.fsp);
    const float c2 = arc_cost_abs_f(g.z, ac.z, w.z != 0, a.fsp), c3 = arc_cost_abs_f(g.w, ac.w, w.w != 0, a.fsp);
    if (c0 < 1.0e30f) m = fmaxf(m, c0);
    if (c1 < 1.0e30f) m = fmaxf(m, c1);
    if (c2 < 1.0e30f) m = fmaxf(m, c2);
    if (c3 < 1.0e30f) m = fmaxf(m, c3);
  }
  for (int64_t i = quads * 4 + (int64_t)blockIdx.x * kT + threadIdx.x; i < num_arcs; i += (int64_t)gridDim.x * kT) {
    const float c = arc_cost_abs_f(a.graph[i], a.acoustic[i], a.fsp.penalized ? a.ilabel[i] != 0 : false, a.fsp);
    if (c < 1.0e30f) m = fmaxf(m, c);
  }
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < num_states; i += (int64_t)gridDim.x * kT) {
    const float c = arc_cost_abs_f(a.final_graph[i], a.final_acoustic[i], false, a.fsp);
    if (c < 1.0e30f) m = fmaxf(m, c);
  }
  for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(a.maxcost, (unsigned long long)__double_as_longlong((double)m));
}

__global__ __launch_bounds__(kT) void pack_rounds_kernel(PackDevArgs a) {
  __shared__ int sh[kT / 64];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u];
  const int S = (int)(a.state_off[u + 1] - s0);
  const int dirs = a.only ? a.only[u] : 3;  // bit 0: the forward rounds, bit 1: the backward rounds (a lattice's other direction may belong to the lean passes)
  if (S <= 0 || a.lat_g[u] < 0 || !dirs) return;
  const int k = a.lat_g[u];
  const int64_t sb = s0 + 2 * (int64_t)u;
  const int32_t* cnt = a.cnt + sb;
  const int32_t* row = a.row + sb;
  int32_t* r0f = a.r0f + sb;
  int32_t* r0b = a.r0b + sb;
  int carry_r = 0, carry_b = 0;
  for (int base = 0; base <= S; base += kT) {
    const int p = base + tid;
    const int nterm = (p <= S) ? cnt[p + 1] - cnt[p] + (p == 0 ? 1 : 0) : 0;  // (cnt[S + 1] - cnt[S] = the final states)
    const int nr = (p <= S) ? (max(nterm, 1) + (1 << k) - 1) >> k : 0;
    int nrb = 0, fin = 0;
    if (p < S) {
      const int s = S - 1 - p;
      fin = dev_is_final(a, s0 + s) ? 1 : 0;
      const int nb = row[s + 1] - row[s] + fin;
      nrb = (max(nb, 1) + (1 << k) - 1) >> k;
    }
    int tot_r, tot_b;
    const int ex_r = block_scan<kT>(nr, &tot_r, sh);
    const int ex_b = block_scan<kT>(nrb, &tot_b, sh);
    if (p <= S && (dirs & 1)) r0f[p] = carry_r + ex_r;
    if (p < S && (dirs & 2)) r0b[p] = (carry_b + ex_b) | (fin << 31);  // bit 31: the position's state is final (the fill pass reads nothing else for it)
    carry_r += tot_r;
    carry_b += tot_b;
  }
  if (tid == 0 && (dirs & 1)) r0f[S + 1] = carry_r;
  if (tid == 0 && (dirs & 2)) r0b[S] = carry_b;
}

// ---- fill ----
#ifndef KLAT_FILL_UNROLL
#define KLAT_FILL_UNROLL 6
#endif
constexpr int kTF = 256;
constexpr int kFillUnroll = KLAT_FILL_UNROLL;
constexpr int kTileRounds = kPackTileRounds;     // 32: 8 chunks, what a job's length is a multiple of
constexpr int kWin = kTileRounds + 2;            // window of positions a tile can hold, per lattice (+ one past the end)
// One workgroup per (job, run of tiles).  Per tile:
//   windows   first round (r0f / r0b; backward: bit 31 = final state) and first term (cnt / row) of
//             the positions that can have a round in the tile: one global round trip;
//   terms     the lattice in slot k has the threads [k * TPS, (k + 1) * TPS) to itself.  Its terms in
//             the tile are one run of in_arc (+ flist: position S) / out_arc: a round trip for the arc
//             numbers, one for the arcs' data, kFillUnroll terms per thread in flight together.  A
//             forward term sits at the position of its destination, a backward term at that of its
//             source, its number within the position is its offset from the position's first term:
//             nothing is searched;
//   positions one thread per position for what is not an arc: the constant-one terms (alpha[start],
//             final weights), NIL terms, padding lanes;
//   store     the tile leaves LDS as whole lines.
__global__ __launch_bounds__(kTF) void pack_fill_kernel(PackDevArgs a, int max_lpw) {
  __shared__ __attribute__((aligned(16))) uint32_t tw[kTileRounds * 64];
  __shared__ __attribute__((aligned(16))) float tg[kTileRounds * 64];
  __shared__ __attribute__((aligned(16))) float ta[kTileRounds * 64];
  extern __shared__ int32_t dyn[];
  __shared__ int64_t sh_s0[64], sh_a0[64];
  __shared__ int sh_plo[64], sh_next[64], sh_S[64], sh_lat[64], sh_n[64], sh_A[64];
  int32_t* win_r = dyn;                    // [LPW][kWin] first round of position plo + o
  int32_t* win_c = dyn + max_lpw * kWin;   // [LPW][kWin] forward: first in-arc of position plo + o; backward: end of its out-arcs
  const PackEntry e = a.entries[blockIdx.x];
  const SweepJob job = a.jobs[e.job];
  const int G = job_lanes(job.tag), W = job_ring(job.tag), AL = job_stored_lanes(job.tag);
  const int lg = 31 - __clz(G), LPW = 64 >> lg, gmask = G - 1;  // G is a power of two
  const int lg_tps = lg + (31 - __clz(kTF)) - 6;                 // log2 of the threads per lattice: kTF / LPW
  const int tid = threadIdx.x;
  const bool fwd = job.dir == 0;
  const int row_units = AL * LPW;                      // 16-byte units of one chunk row
  const int tile_slots = kTileRounds * row_units;      // stream slots of one tile
  int nfar = 0;
  // the lattices of the job and, for each, the first position with a round in this run of tiles
  if (tid < LPW) {
    const LaneInfo li = a.lanes[job.lane_tab + tid];
    int plo = 0, S = 0, A = 0;
    int64_t s0 = 0, a0 = 0;
    if (li.lattice >= 0) {
      S = li.num_states;
      s0 = li.state_base;
      a0 = a.arc_off[li.lattice];
      A = (int)(a.arc_off[li.lattice + 1] - a0);
      const int32_t* r0 = (fwd ? a.r0f : a.r0b) + s0 + 2 * (int64_t)li.lattice;
      const int P = fwd ? S + 1 : S;  // positions; r0[P] = the lattice's rounds
      const int R0 = e.tile_begin * kTileRounds;
      int lo = 0, hi = P;             // first p with r0[p + 1] > R0 (P if none)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((r0[mid + 1] & kM) > R0)
          hi = mid;
        else
          lo = mid + 1;
      }
      plo = lo;
    }
    sh_lat[tid] = li.lattice;
    sh_S[tid] = S;
    sh_A[tid] = A;
    sh_plo[tid] = plo;
    sh_s0[tid] = s0;
    sh_a0[tid] = a0;
  }
  __syncthreads();
  for (int t = 0; t < e.tile_count; ++t) {
    const int R0 = (e.tile_begin + t) * kTileRounds, R1 = R0 + kTileRounds;
    // ---- padding everywhere (zero-weight terms that park their unused result in the lattice's
    // dummy cell behind the ring) and the windows
    for (int i = tid; i < (tile_slots >> 2); i += kTF) {  // one 16-byte unit (4 rounds of a lane) per store
      const int slot = i & (LPW - 1);  // unit = lane * LPW + slot, a row is a whole number of lanes
      const uint32_t w = kNil | ((uint32_t)(W * LPW + slot) << kWrShift);
      reinterpret_cast<uint4*>(tw)[i] = make_uint4(w, w, w, w);
      reinterpret_cast<float4*>(tg)[i] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
      reinterpret_cast<float4*>(ta)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    for (int i = tid; i < LPW * kWin; i += kTF) {
      const int k = i / kWin, o = i - k * kWin;
      int vr = kM, vc = 0;
      const int u = sh_lat[k];
      if (u >= 0) {
        const int S = sh_S[k], P = fwd ? S + 1 : S;
        const int p = sh_plo[k] + o;
        const int64_t sb = sh_s0[k] + 2 * (int64_t)u;
        if (p <= P) vr = ((fwd ? a.r0f : a.r0b) + sb)[p];
        vc = fwd ? a.cnt[sb + min(p, S + 1)] : a.row[sb + max(S - p, 0)];
        if (o == 0) sh_next[k] = P;
      }
      if (o == 0) sh_n[k] = 0;
      win_r[i] = vr;
      win_c[i] = vc;
    }
    __syncthreads();
    // ---- which positions have a round in the tile; the first position of the next tile
    for (int i = tid; i < LPW * (kTileRounds + 1); i += kTF) {
      const int k = i / (kTileRounds + 1), o = i - k * (kTileRounds + 1);
      if (sh_lat[k] < 0) continue;
      const int P = fwd ? sh_S[k] + 1 : sh_S[k];
      const int p = sh_plo[k] + o;
      if (p >= P) continue;
      const int r = win_r[k * kWin + o] & kM, rn = win_r[k * kWin + o + 1] & kM;
      if (r <= R1 && rn > R1) sh_next[k] = p;
      if (r >= R1 || rn <= R0) continue;
      atomicMax(&sh_n[k], o + 1);
    }
    __syncthreads();
    auto put = [&](int k, int rr, int sub, uint32_t w, float g, float ac) {
      const int i = ((((rr - R0) >> 2) * row_units + sub * LPW + k) << 2) + ((rr - R0) & 3);
      tw[i] = w;
      if (a.fused) {
        // the cost as the reference forms it (ScaleLattice, AddWordInsPen, ConvertToCost), kept as a float and
        // a 16-bit residual in units of the batch's lo_unit
        const double c = arc_cost(g, ac, (w & kWord) != 0, a.fsp);
        const float hi = (float)c;
        int n = 0;
        if (c < 1.0e30 && c > -1.0e30) n = max(-32767, min(32767, (int)rint((c - (double)hi) * a.lo_inv_unit)));
        tg[i] = hi;
        ta[i] = __int_as_float(n);
      } else {
        tg[i] = g;
        ta[i] = ac;
      }
    };
    // ---- one thread per arc term
    {
      const int k = tid >> lg_tps, gl = tid & ((1 << lg_tps) - 1);
      const int n = sh_n[k];
      const int32_t* wc = win_c + k * kWin;
      const int32_t* wr = win_r + k * kWin;
      const int nterms = n > 0 ? (fwd ? wc[n] - wc[0] : wc[0] - wc[n]) : 0;
      const int first = n > 0 ? (fwd ? wc[0] : wc[n]) : 0;  // the lattice's first term here: in-arc number / out-arc number
      const int plo = sh_plo[k], S = sh_S[k], A = sh_A[k];
      const int64_t a0 = sh_a0[k], s0 = sh_s0[k];
      const int32_t* ia = (fwd ? a.in_arc : a.out_arc) + a0;
      const TermCtx tc = {lg, W, LPW, k};
      for (int tb = gl; tb < nterms; tb += kFillUnroll << lg_tps) {
        int tx[kFillUnroll], arc[kFillUnroll];
#pragma unroll
        for (int u = 0; u < kFillUnroll; ++u) {
          tx[u] = first + min(tb + (u << lg_tps), nterms - 1);  // (a lane without a term repeats the last one and drops it)
          arc[u] = tx[u] < A ? ia[tx[u]] : a.flist[s0 + tx[u] - A];  // (forward, position S: one term per final state)
        }
        int vs[kFillUnroll], vd[kFillUnroll], il[kFillUnroll];
        float gg[kFillUnroll], aa[kFillUnroll];
#pragma unroll
        for (int u = 0; u < kFillUnroll; ++u) {
          if (tx[u] < A) {
            const int64_t x = a0 + arc[u];
            vs[u] = a.src[x];
            vd[u] = a.dst[x];
            il[u] = a.ilabel[x];
            gg[u] = a.graph[x];
            aa[u] = a.acoustic[x];
          } else {
            vs[u] = arc[u], vd[u] = S, il[u] = 0;
            gg[u] = a.final_graph[s0 + arc[u]];
            aa[u] = a.final_acoustic[s0 + arc[u]];
          }
        }
#pragma unroll
        for (int u = 0; u < kFillUnroll; ++u) {
          const int p = fwd ? vd[u] : S - 1 - vs[u];
          const int o = p - plo;
          const int r = wr[o] & kM, nr = (wr[o + 1] & kM) - r;
          // the term's number within its position: its offset from the position's first term, behind the constant-one term
          const int j = fwd ? tx[u] - wc[o] + (p == 0 ? 1 : 0) : tx[u] - wc[o + 1] + (int)((uint32_t)wr[o] >> 31);
          const int rr = r + (j >> lg);
          if (tb + (u << lg_tps) >= nterms || rr < R0 || rr >= R1) continue;
          bool far;
          const uint32_t w = term_word(tc, p, r, nr, j, vd[u] - vs[u], il[u] != 0, false, &far);
          nfar += far ? 1 : 0;
          put(k, rr, j & gmask, w, gg[u], aa[u]);
        }
      }
    }
    // ---- one thread per position: what is not an arc
    for (int i = tid; i < LPW * (kTileRounds + 1); i += kTF) {
      const int k = i / (kTileRounds + 1), o = i - k * (kTileRounds + 1);
      if (o >= sh_n[k]) continue;
      const int S = sh_S[k];
      const int p = sh_plo[k] + o;
      const int r = win_r[k * kWin + o] & kM, nr = (win_r[k * kWin + o + 1] & kM) - r;
      const int jlo = max(0, (R0 - r) << lg), jhi = min(nr << lg, (R1 - r) << lg);
      const int deg = fwd ? win_c[k * kWin + o + 1] - win_c[k * kWin + o] : win_c[k * kWin + o] - win_c[k * kWin + o + 1];
      const bool lead = fwd ? p == 0 : ((uint32_t)win_r[k * kWin + o] >> 31) != 0;  // alpha[start] = 0; a final state's weight
      const int nterm = deg + (lead ? 1 : 0);
      const TermCtx tc = {lg, W, LPW, k};
      bool far;
      if (jlo == 0 && (lead || nterm == 0)) {
        float g = 0.0f, ac = 0.0f;
        if (lead && !fwd) g = a.final_graph[sh_s0[k] + S - 1 - p], ac = a.final_acoustic[sh_s0[k] + S - 1 - p];
        const uint32_t w = term_word(tc, p, r, nr, 0, 0, false, !lead, &far);
        put(k, r, 0, w, lead ? g : INFINITY, ac);
      }
      // lanes of the position's rounds that carry no term
      const uint32_t dummy = (uint32_t)(W * LPW + k) << kWrShift;
      for (int jj = max(max(nterm, 1), jlo); jj < jhi; ++jj) {
        if ((jj & gmask) >= AL) continue;
        const int rr = r + (jj >> lg);
        uint32_t w = kNil | dummy;
        if (rr == r) w |= kFirst;
        if (rr == r + nr - 1) w |= kLast;
        put(k, rr, jj & gmask, w, INFINITY, 0.0f);
      }
    }
    __syncthreads();
    // ---- the tile leaves as whole lines
    {
      const int64_t base = (job.chunk_off + (int64_t)(R0 >> 2) * row_units) << 2;  // first slot of the tile
      uint4* gw = reinterpret_cast<uint4*>(a.sw + base);
      float4* gg = reinterpret_cast<float4*>(a.sg + base);
      const uint4* lw = reinterpret_cast<const uint4*>(tw);
      const float4* lg4 = reinterpret_cast<const float4*>(tg);
      const float4* la = reinterpret_cast<const float4*>(ta);
      if (a.fused) {  // the residuals as four 16-bit words per unit
        int2* gl = reinterpret_cast<int2*>(a.sl + base);
        for (int i = tid; i < (tile_slots >> 2); i += kTF) {
          gw[i] = lw[i];
          gg[i] = lg4[i];
          const float4 r = la[i];
          const int n0 = __float_as_int(r.x), n1 = __float_as_int(r.y), n2 = __float_as_int(r.z), n3 = __float_as_int(r.w);
          gl[i] = make_int2((n0 & 0xFFFF) | (n1 << 16), (n2 & 0xFFFF) | (n3 << 16));
        }
      } else {
        float4* ga = reinterpret_cast<float4*>(a.sa + base);
        for (int i = tid; i < (tile_slots >> 2); i += kTF) {
          gw[i] = lw[i];
          gg[i] = lg4[i];
          ga[i] = la[i];
        }
      }
    }
    if (tid < LPW) sh_plo[tid] = sh_next[tid];
    __syncthreads();
  }
  for (int d = 32; d > 0; d >>= 1) nfar += __shfl_xor(nfar, d, 64);
  if ((tid & 63) == 0 && nfar) atomicAdd(&a.status[3], nfar);
}

}  // namespace

hipError_t launch_pack_count(const PackDevArgs& a, int num_lattices, const int class_need[kPackCountClasses], bool any_global,
                             hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  // LDS classes: a lattice runs in the launch of the narrowest class that holds its tables, so a
  // batch of small lattices keeps four workgroups on a CU whatever its largest lattice needs
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPackCountMaxLds);
    attr_done = true;
  }
  for (int c = 0; c < kPackCountClasses; ++c)
    if (class_need[c] > 0)
      hipLaunchKernelGGL(pack_count_kernel, dim3(num_lattices), dim3(kTC), (size_t)class_need[c], stream, a,
                         c == 0 ? -1 : kPackCountClassLds[c - 1], kPackCountClassLds[c]);
  if (any_global) {
    static bool global_attr_done = false;
    if (!global_attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_count_global_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPackCountMaxLds);
      global_attr_done = true;
    }
    hipLaunchKernelGGL(pack_count_global_kernel, dim3(num_lattices), dim3(kTG), (size_t)kPackCountMaxLds, stream, a, kPackCountMaxLds, kPackCountMaxLds);
  }
  return hipGetLastError();
}

hipError_t launch_pack_maxcost(const PackDevArgs& a, int64_t num_arcs, int64_t num_states, hipStream_t stream) {
  hipLaunchKernelGGL(pack_maxcost_kernel, dim3(2048), dim3(kT), 0, stream, a, num_arcs, num_states);
  return hipGetLastError();
}

hipError_t launch_pack_rounds(const PackDevArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_rounds_kernel, dim3(num_lattices), dim3(kT), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_pack_fill(const PackDevArgs& a, int num_entries, int max_lpw, hipStream_t stream) {
  if (num_entries <= 0) return hipSuccess;
  const size_t lds = (size_t)max_lpw * kWin * 2 * sizeof(int32_t);
  hipLaunchKernelGGL(pack_fill_kernel, dim3(num_entries), dim3(kTF), lds, stream, a, max_lpw);
  return hipGetLastError();
}

}  // namespace klat
