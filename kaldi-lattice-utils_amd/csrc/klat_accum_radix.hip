// klat_accum_radix.hip -- the segment / position index for lattices of up to 8192 arcs whose keys
// pack into 62 bits: same results as index_kernel (klat_accum.hip), with the two bitonic sorts
// replaced by block radix sorts (rocPRIM's block_radix_sort primitive inside our kernel).
//
//   1. every thread owns 8 consecutive arcs (blocked: position order = arc order); key =
//      (label - label_min | t0 | t1) or (label - label_min | pos), value = arc index; filtered-out
//      arcs get a key above all others.  Stable LSD radix sort on the bits in use.
//   2. runs of equal keys are folded in arc order (the sort is stable) with the reference's LogAdd
//      (kwsbin2/lattice-word-index-segment.cc:170, kwsbin2/lattice-word-index-position.cc:174-181).
//   3. the folded records, still in (label, t0, t1) order, are sorted by descending logp with a
//      second stable radix sort on all 64 bits of the order-preserving image of the double: ties
//      keep the (label, t0, t1) / (label, pos) order, which is the reference's comparator
//      (kwsbin2/lattice-word-index-segment.cc:98-126, -position.cc:102-127).
// One workgroup of 1024 threads per lattice; 8 items per thread (lattices of up to 8192 arcs, 64 KiB
// of LDS, two workgroups per CU) or 16 (up to 16384 arcs, 132 KiB, one workgroup per CU).  A launch
// takes the lattices whose arc count lies in (min_arcs, max_arcs]; the others belong to another
// instantiation or to the general kernel (klat_accum.hip).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include <rocprim/block/block_radix_sort.hpp>

#include "klat_accum.h"

namespace klat {

namespace {

constexpr int kThreads = 1024;
template <int ITEMS>
using SortT = rocprim::block_radix_sort<uint64_t, kThreads, ITEMS, uint32_t>;

__device__ __forceinline__ double radix_log_add(double x, double y) {
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
__device__ __forceinline__ uint32_t rbias(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
__device__ __forceinline__ uint64_t rdesc_key(double x) {  // descending double -> ascending uint64
  uint64_t b = (uint64_t)__double_as_longlong(x);
  b = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
  return ~b;
}
__device__ __forceinline__ bool rin_sorted(const int32_t* set, int n, int32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (set[mid] < v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < n && set[lo] == v;
}

struct Payload {  // 16 B, kept in the (otherwise unused) key scratch of the lattice
  uint64_t k1, k2;
};
struct Rec {  // 24 B: what index_gather_kernel reads
  uint64_t k0, k1, k2;
};

// exclusive prefix sum of one int per thread over the block; *total = block sum
__device__ int block_exclusive_sum(int v, int* total, int* sh /* [kThreads/64 + 1] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < kThreads / 64; ++w) {
    const int s = sh[w];
    if (w < wave) off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return off + incl - v;
}

template <bool POSITION, int kItems>
__global__ __launch_bounds__(kThreads) void index_radix_kernel(AccumArgs a, RadixPack pk) {
  using Sort = SortT<kItems>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int sh[kThreads / 64 + 1];
  typename Sort::storage_type& storage = *reinterpret_cast<typename Sort::storage_type*>(smem);
  uint64_t* skey = reinterpret_cast<uint64_t*>(smem);  // [kThreads * kItems], after a sort
  const int u = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t a0 = a.arc_off[u], a1 = a.arc_off[u + 1];
  const int64_t s0 = a.state_off[u];
  const int nA = (int)(a1 - a0);
  if (nA <= a.min_arcs || nA > a.max_arcs) return;  // another launch's lattice (workgroup-uniform)
  // the sorted values beside skey: in LDS behind the keys with 8 items per thread, in the lattice's
  // `cap` slots of arc_order (L2) with 16, where LDS has no room for them
  using ArcT = std::conditional_t<kItems == 8, uint16_t, uint32_t>;
  ArcT* sarc = kItems == 8 ? reinterpret_cast<ArcT*>(smem + sizeof(uint64_t) * kThreads * kItems)
                           : reinterpret_cast<ArcT*>(a.arc_order + a.scratch_off[u]);
  const int cap = kItems == 8 ? kThreads * kItems : (int)(a.scratch_off[u + 1] - a.scratch_off[u]);
  Payload* payload = reinterpret_cast<Payload*>(a.key_scratch) + a.scratch_off[u];
  Rec* recs = reinterpret_cast<Rec*>(a.rec_scratch) + a.scratch_off[u];
  const uint64_t invalid = 1ull << pk.total_bits;

  // ---- 1. keys in arc order, sort ----
  uint64_t keys[kItems];
  uint32_t vals[kItems];
  int nvalid = 0;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int i = tid * kItems + j;
    keys[j] = invalid;
    vals[j] = (uint32_t)i;
    if (i < nA) {
      const int32_t label = a.olabel[a0 + i];
      const bool valid = label != 0 && ((a.num_include > 0 && rin_sorted(a.include, a.num_include, label)) ||
                                        (a.num_include == 0 && !rin_sorted(a.exclude, a.num_exclude, label)));
      if (valid) {
        const uint64_t l = (uint64_t)(uint32_t)(label - pk.label_min);
        const int32_t s = a.src[a0 + i];
        if (POSITION) {
          keys[j] = (l << pk.time_bits) | (uint64_t)(uint32_t)a.state_len[s0 + s];
        } else {
          const int32_t d = a.dst[a0 + i];
          keys[j] = (((l << pk.time_bits) | (uint64_t)(uint32_t)a.state_time[s0 + s]) << pk.time_bits) |
                    (uint64_t)(uint32_t)a.state_time[s0 + d];
        }
        nvalid++;
      }
    }
  }
  int n;
  (void)block_exclusive_sum(nvalid, &n, sh);
  Sort().sort(keys, vals, storage, 0, pk.total_bits + 1);
  __syncthreads();  // the sort's storage becomes skey / sarc
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    skey[tid * kItems + j] = keys[j];
    if (tid * kItems + j < cap) sarc[tid * kItems + j] = (ArcT)vals[j];
  }
  __syncthreads();

  // ---- 2. fold runs of equal keys in arc order ----
  uint32_t head_mask = 0;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int i = tid * kItems + j;
    if (i < n && (i == 0 || skey[i - 1] != skey[i])) head_mask |= 1u << j;
  }
  int m;
  int id = block_exclusive_sum(__popc(head_mask), &m, sh);  // record ids follow the key order
  uint64_t keys2[kItems];
  uint32_t vals2[kItems];
  const uint64_t tmask = (1ull << pk.time_bits) - 1ull;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int i = tid * kItems + j;
    keys2[j] = ~0ull;
    vals2[j] = 0xFFFFFFFFu;
    if (head_mask & (1u << j)) {
      // (key and arc from skey / sarc, not from the sort's registers: those are dead by now, which
      // the 16-item instantiation needs to stay within its 128 registers)
      const uint64_t kj = skey[i];
      int arc = (int)sarc[i];
      double acc = a.logpost[a0 + arc];
      double best = acc;
      int best_arc = arc;
      for (int k = i + 1; k < n && skey[k] == kj; ++k) {
        arc = (int)sarc[k];
        const double x = a.logpost[a0 + arc];
        acc = radix_log_add(acc, x);
        if (x > best) {  // strict: the first maximum in arc order wins
          best = x;
          best_arc = arc;
        }
      }
      Payload q;
      if (POSITION) {
        const int32_t label = (int32_t)(uint32_t)(kj >> pk.time_bits) + pk.label_min;
        const int32_t pos = (int32_t)(uint32_t)(kj & tmask) + 1;  // 1-based on output
        q.k1 = ((uint64_t)rbias(label) << 32) | rbias(pos);
        const int32_t t0 = a.state_time[s0 + a.src[a0 + best_arc]];
        const int32_t t1 = a.state_time[s0 + a.dst[a0 + best_arc]];
        q.k2 = ((uint64_t)(uint32_t)t0 << 32) | (uint32_t)t1;
      } else {
        const int32_t label = (int32_t)(uint32_t)(kj >> (2 * pk.time_bits)) + pk.label_min;
        const int32_t t0 = (int32_t)(uint32_t)((kj >> pk.time_bits) & tmask);
        const int32_t t1 = (int32_t)(uint32_t)(kj & tmask);
        q.k1 = ((uint64_t)rbias(label) << 32) | rbias(t0);
        q.k2 = (uint64_t)rbias(t1) << 32;
      }
      payload[id] = q;
      keys2[j] = rdesc_key(acc);
      vals2[j] = (uint32_t)id;
      ++id;
    }
  }
  // the payload is read back by other threads of this workgroup below: workgroup scope is enough
  // (an agent-scope fence writes the L2 back on this multi-XCD part, ~90 us per workgroup)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();   // ... and skey / sarc are dead: the storage is the sort's again

  // ---- 3. order by descending logp (stable: ties keep the key order) ----
  // The order-preserving images of one lattice's values share their upper bits, and 64 bits of
  // radix passes are 8 passes.  Instead: subtract the smallest image (the rest needs `hb` bits),
  // run the passes over the upper 23 of those bits plus one bit that sends the padding to the end
  // (3 passes), and let the first thread of every run of records that agree in the sorted bits
  // (relative difference ~1e-6: a handful per lattice) order it on all bits with a stable
  // insertion sort in LDS.  A long unsorted run (> kLongRun records) takes the remaining passes
  // instead: lower bits, then the upper bits again -- stable passes over the current arrangement
  // give the same final order.
  constexpr int kLongRun = 128;
  constexpr int kUpperBits = 24;
  __shared__ unsigned long long sh_min, sh_max;
  if (tid == 0) sh_min = ~0ull, sh_max = 0ull, sh[kThreads / 64] = 0;
  __syncthreads();
  {
    unsigned long long lo = ~0ull, hi = 0ull;
#pragma unroll
    for (int j = 0; j < kItems; ++j)
      if (head_mask & (1u << j)) lo = min(lo, (unsigned long long)keys2[j]), hi = max(hi, (unsigned long long)keys2[j]);
    for (int d = 32; d > 0; d >>= 1) {
      lo = min(lo, (unsigned long long)__shfl_xor(lo, d, 64));
      hi = max(hi, (unsigned long long)__shfl_xor(hi, d, 64));
    }
    if ((tid & 63) == 0 && lo <= hi) atomicMin(&sh_min, lo), atomicMax(&sh_max, hi);
  }
  __syncthreads();
  const uint64_t kmin = m > 0 ? sh_min : 0ull;
  const uint64_t range = m > 0 ? sh_max - sh_min : 0ull;
  const int hb = range ? 64 - __clzll((long long)range) : 0;  // bits of the largest (image - kmin)
  const bool narrow = hb <= 62;  // else: values of both signs; sort the raw images' upper half
  const int sort_end = narrow ? hb + 1 : 64;
  const int sort_begin = narrow ? max(0, sort_end - kUpperBits) : 32;
  if (narrow) {
#pragma unroll
    for (int j = 0; j < kItems; ++j) keys2[j] = (head_mask & (1u << j)) ? keys2[j] - kmin : (1ull << hb);
  }
  Sort().sort(keys2, vals2, storage, sort_begin, sort_end);
  __syncthreads();  // the sort's storage becomes skey / sarc again
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    skey[tid * kItems + j] = keys2[j];
    if (tid * kItems + j < cap) sarc[tid * kItems + j] = (ArcT)vals2[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int p = tid * kItems + j;
    const uint64_t hi = keys2[j] >> sort_begin;
    // the sorted bits do not move during the fix-up (a run permutes records that agree in them)
    if (p + 1 < m && (skey[p + 1] >> sort_begin) == hi && (p == 0 || (skey[p - 1] >> sort_begin) != hi)) {
      int q = p + 1;
      bool sorted = true;
      while (q < m && (skey[q] >> sort_begin) == hi) {
        sorted = sorted && skey[q - 1] <= skey[q];
        ++q;
      }
      if (sorted) continue;
      if (q - p > kLongRun) {
        sh[kThreads / 64] = 1;
        continue;
      }
      for (int i = p + 1; i < q; ++i) {
        const uint64_t k = skey[i];
        const ArcT v = sarc[i];
        int t = i - 1;
        while (t >= p && skey[t] > k) {
          skey[t + 1] = skey[t];
          sarc[t + 1] = sarc[t];
          --t;
        }
        skey[t + 1] = k;
        sarc[t + 1] = v;
      }
    }
  }
  __syncthreads();
  const bool long_run = sh[kThreads / 64] != 0;  // workgroup-uniform
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    keys2[j] = skey[tid * kItems + j];
    vals2[j] = (tid * kItems + j < cap) ? sarc[tid * kItems + j] : 0xFFFFFFFFu;
  }
  if (long_run) {
    __syncthreads();  // skey / sarc are read: the storage is the sort's again
    Sort().sort(keys2, vals2, storage, 0, sort_begin);
    __syncthreads();
    Sort().sort(keys2, vals2, storage, sort_begin, sort_end);
  }
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    const int p = tid * kItems + j;
    if (p < m) {
      const Payload q = payload[vals2[j]];
      Rec r;
      r.k0 = narrow ? keys2[j] + kmin : keys2[j];
      r.k1 = q.k1;
      r.k2 = q.k2;
      recs[p] = r;
    }
  }
  if (tid == 0) a.num_records[u] = m;
}

template <bool POSITION, int ITEMS>
static hipError_t launch_one(const AccumArgs& a, const RadixPack& pk, int num_lattices, hipStream_t stream) {
  const size_t arrays = (sizeof(uint64_t) + (ITEMS == 8 ? sizeof(uint16_t) : 0)) * kThreads * ITEMS;  // skey (+ sarc)
  const size_t lds = std::max(sizeof(typename SortT<ITEMS>::storage_type), arrays);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&index_radix_kernel<POSITION, ITEMS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((index_radix_kernel<POSITION, ITEMS>), dim3(num_lattices), dim3(kThreads), lds, stream, a, pk);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_index_radix(const AccumArgs& a, const RadixPack& pk, int num_lattices, bool position, int items,
                              hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  if (items == 8) return position ? launch_one<true, 8>(a, pk, num_lattices, stream) : launch_one<false, 8>(a, pk, num_lattices, stream);
  if (items == 16) return position ? launch_one<true, 16>(a, pk, num_lattices, stream) : launch_one<false, 16>(a, pk, num_lattices, stream);
  return hipErrorInvalidValue;
}

}  // namespace klat
