// klat_accum.hip -- second stage: the reference's key accumulators over the arc posteriors.
//
//   segment index  kwsbin2/lattice-word-index-segment.cc:143-174 (acc[label][(t0,t1)] = LogAdd)
//                  and :98-126 (sort by logp desc, label, t0, t1)
//   position index kwsbin2/lattice-word-index-position.cc:144-187 (key (label,pos); LogAdd sum
//                  plus the (t0,t1) of the FIRST arc attaining the maximum, strict '>' :178)
//                  and :100-127 (pos 1-based, sort by logp desc, label, pos)
//
// The reference does this with two-level std::map inserts per arc.  Here one workgroup owns one
// lattice: (1) filter arcs and emit 16-byte keys (label,t0,t1 | label,pos ; arc index),
// (2) bitonic sort of the keys, (3) each run of equal keys is folded in arc order with the
// reference's LogAdd (cutoff log(DBL_EPSILON) included), (4) the folded records are
// bitonic-sorted by the reference's output comparator.  The sorts run in a dynamic LDS tile sized
// by the host for the largest lattice of the batch (up to 128 KiB: 8192 keys), so a C2-sized
// lattice is sorted with one load and one store; larger lattices sort tile by tile and merge with
// global passes.  Utterance-level parallelism (thousands of lattices) fills the chip; there is no
// inter-workgroup communication.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <vector>

#include "klat_accum.h"

namespace klat {

static constexpr int kAccThreads = 1024;

// LogAdd(double,double) of SURVEY.md A.2, as the reference's accumulators call it.
__device__ __forceinline__ double log_add(double x, double y) {
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

__device__ __forceinline__ uint32_t bias(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
__device__ __forceinline__ int32_t unbias(uint32_t v) { return (int32_t)(v ^ 0x80000000u); }

// descending double -> ascending uint64
__device__ __forceinline__ uint64_t desc_key(double x) {
  uint64_t b = (uint64_t)__double_as_longlong(x);
  b = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);  // ascending order key
  return ~b;                                                           // flip: descending
}
__device__ __forceinline__ double desc_unkey(uint64_t k) {
  uint64_t b = ~k;
  b = (b & 0x8000000000000000ull) ? (b & 0x7FFFFFFFFFFFFFFFull) : ~b;
  return __longlong_as_double((long long)b);
}

struct Key {  // 16 B
  uint64_t a;  // segment: label|t0     position: label|pos
  uint64_t b;  // segment: t1|arc       position: 0|arc
};
struct Rec {  // 24 B
  uint64_t k0;  // desc_key(logp)
  uint64_t k1;  // segment: label|t0    position: label|pos
  uint64_t k2;  // segment: t1|0        position: t0|t1 (payload, not compared)
};

struct KeyLess {
  __device__ bool operator()(const Key& x, const Key& y) const { return x.a < y.a || (x.a == y.a && x.b < y.b); }
};
template <bool USE_K2>
struct RecLess {
  __device__ bool operator()(const Rec& x, const Rec& y) const {
    if (x.k0 != y.k0) return x.k0 < y.k0;
    if (x.k1 != y.k1) return x.k1 < y.k1;
    return USE_K2 ? (x.k2 < y.k2) : false;
  }
};

// one compare-exchange pass of stride j inside an LDS tile of tl elements starting at global
// index `base`; k is the bitonic block size that fixes the direction
template <class T, class Less>
__device__ __forceinline__ void tile_pass(T* tile, int tl, int base, int k, int j, Less less) {
  for (int i = threadIdx.x; i < tl; i += kAccThreads) {
    const int l = i ^ j;
    if (l > i) {
      const T x = tile[i], y = tile[l];
      const bool up = ((base + i) & k) == 0;
      if (up ? less(y, x) : less(x, y)) {
        tile[i] = y;
        tile[l] = x;
      }
    }
  }
  __syncthreads();
}

// In-place bitonic sort of n (power of two) elements at `data` (global) by one workgroup, with an
// LDS tile of `tile_elems` (power of two) elements.
template <class T, class Less>
__device__ void block_bitonic_sort(T* data, int n, T* tile, int tile_elems, Less less) {
  const int tid = threadIdx.x;
  const int tl = n < tile_elems ? n : tile_elems;
  // phase 1: every tile is sorted completely in LDS (all k <= tl), directions as in the full network
  for (int base = 0; base < n; base += tl) {
    for (int i = tid; i < tl; i += kAccThreads) tile[i] = data[base + i];
    __syncthreads();
    for (int k = 2; k <= tl; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) tile_pass(tile, tl, base, k, j, less);
    for (int i = tid; i < tl; i += kAccThreads) data[base + i] = tile[i];
    __syncthreads();
  }
  // phase 2: merge levels above the tile size: global passes for strides >= tl, then one LDS
  // visit per tile for the strides below
  for (int k = tl << 1; k <= n; k <<= 1) {
    for (int j = k >> 1; j >= tl; j >>= 1) {
      for (int i = tid; i < n; i += kAccThreads) {
        const int l = i ^ j;
        if (l > i) {
          const T x = data[i], y = data[l];
          const bool up = (i & k) == 0;
          if (up ? less(y, x) : less(x, y)) {
            data[i] = y;
            data[l] = x;
          }
        }
      }
      __syncthreads();
    }
    for (int base = 0; base < n; base += tl) {
      for (int i = tid; i < tl; i += kAccThreads) tile[i] = data[base + i];
      __syncthreads();
      for (int j = tl >> 1; j > 0; j >>= 1) tile_pass(tile, tl, base, k, j, less);
      for (int i = tid; i < tl; i += kAccThreads) data[base + i] = tile[i];
      __syncthreads();
    }
  }
}

__device__ __forceinline__ bool in_sorted(const int32_t* set, int n, int32_t v) {
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

// block-wide exclusive prefix of a 0/1 flag; returns the thread's offset, *total the block sum
__device__ int block_compact_offset(bool flag, int* total, int* sh /* [kAccThreads/64] */) {
  const uint64_t ballot = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int within = __popcll(ballot & ((1ull << lane) - 1ull));
  if (lane == 0) sh[wave] = __popcll(ballot);
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < kAccThreads / 64; ++w) {
    if (w < wave) off += sh[w];
    tot += sh[w];
  }
  __syncthreads();
  *total = tot;
  return off + within;
}

template <bool POSITION>
__global__ __launch_bounds__(kAccThreads) void index_kernel(AccumArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int sh[kAccThreads / 64];
  const int u = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t a0 = a.arc_off[u], a1 = a.arc_off[u + 1];
  const int64_t s0 = a.state_off[u];
  const int64_t pb = padded_state_base(s0, u);
  const int nA = (int)(a1 - a0);
  if (nA <= a.min_arcs || nA > a.max_arcs) return;  // the hash kernel's lattice (workgroup-uniform)
  if (a.piece_count && a.piece_count[u] > 0) return;  // a long lattice the hash kernel took in pieces
  Key* keys = reinterpret_cast<Key*>(a.key_scratch) + a.scratch_off[u];
  Rec* recs = reinterpret_cast<Rec*>(a.rec_scratch) + a.scratch_off[u];
  const int cap = (int)(a.scratch_off[u + 1] - a.scratch_off[u]);  // power of two >= nA (or 0)
  // LDS tile capacities (powers of two) for the two element sizes
  int key_tile = 1, rec_tile = 1;
  while ((size_t)key_tile * 2 * sizeof(Key) <= (size_t)a.tile_bytes) key_tile <<= 1;
  while ((size_t)rec_tile * 2 * sizeof(Rec) <= (size_t)a.tile_bytes) rec_tile <<= 1;

  // ---- 1. filter + keys (compacted; order is fixed later by the arc index in the key) ----
  int n = 0;
  for (int base = 0; base < nA; base += kAccThreads) {
    const int i = base + tid;
    bool valid = false;
    Key key;
    key.a = key.b = 0;
    if (i < nA) {
      const int32_t label = a.olabel[a0 + i];
      valid = label != 0 && ((a.num_include > 0 && in_sorted(a.include, a.num_include, label)) ||
                             (a.num_include == 0 && !in_sorted(a.exclude, a.num_exclude, label)));
      if (valid) {
        const int32_t s = a.src[a0 + i], d = a.dst[a0 + i];
        if (POSITION) {
          key.a = ((uint64_t)bias(label) << 32) | bias(a.state_len[s0 + s]);
          key.b = (uint32_t)i;
        } else {
          key.a = ((uint64_t)bias(label) << 32) | bias(a.state_time[s0 + s]);
          key.b = ((uint64_t)bias(a.state_time[s0 + d]) << 32) | (uint32_t)i;
        }
      }
    }
    int tot;
    const int off = block_compact_offset(valid, &tot, sh);
    if (valid) keys[n + off] = key;
    n += tot;
  }
  int np = 1;
  while (np < n) np <<= 1;
  if (n == 0) np = 0;
  for (int i = n + tid; i < np; i += kAccThreads) {
    Key pad;
    pad.a = pad.b = ~0ull;
    keys[i] = pad;
  }
  __syncthreads();
  if (np > cap) {  // cannot happen: cap = next_pow2(arcs)
    if (tid == 0) a.num_records[u] = -1;
    return;
  }

  // ---- 2. sort keys ----
  if (np > 1) block_bitonic_sort<Key, KeyLess>(keys, np, reinterpret_cast<Key*>(smem), key_tile, KeyLess());

  // ---- 3. fold runs of equal keys in arc order ----
  const uint64_t group_mask_b = POSITION ? 0ull : 0xFFFFFFFF00000000ull;
  int m = 0;
  for (int base = 0; base < n; base += kAccThreads) {
    const int i = base + tid;
    bool head = false;
    Rec rec;
    rec.k0 = rec.k1 = rec.k2 = 0;
    if (i < n) {
      const Key ki = keys[i];
      head = (i == 0);
      if (!head) {
        const Key kp = keys[i - 1];
        head = (kp.a != ki.a) || ((kp.b & group_mask_b) != (ki.b & group_mask_b));
      }
      if (head) {
        int arc = (int)(uint32_t)ki.b;
        double acc = accum_through(a, a0, pb, arc);
        double best = acc;
        int best_arc = arc;
        for (int j = i + 1; j < n; ++j) {
          const Key kj = keys[j];
          if (kj.a != ki.a || ((kj.b & group_mask_b) != (ki.b & group_mask_b))) break;
          arc = (int)(uint32_t)kj.b;
          const double x = accum_through(a, a0, pb, arc);
          acc = log_add(acc, x);
          if (x > best) {  // strict: the first maximum in arc order wins
            best = x;
            best_arc = arc;
          }
        }
        rec.k0 = desc_key(accum_normalise(acc, a.total[u]));  // lkh - total_lkh_, kwsbin2/lattice-word-index-segment.cc:107
        if (POSITION) {
          const int32_t label = unbias((uint32_t)(ki.a >> 32));
          const int32_t pos = unbias((uint32_t)ki.a) + 1;  // 1-based on output
          rec.k1 = ((uint64_t)bias(label) << 32) | bias(pos);
          const int32_t t0 = a.state_time[s0 + a.src[a0 + best_arc]];
          const int32_t t1 = a.state_time[s0 + a.dst[a0 + best_arc]];
          rec.k2 = ((uint64_t)(uint32_t)t0 << 32) | (uint32_t)t1;
        } else {
          rec.k1 = ki.a;
          rec.k2 = ki.b & 0xFFFFFFFF00000000ull;
        }
      }
    }
    int tot;
    const int off = block_compact_offset(head, &tot, sh);
    if (head) recs[m + off] = rec;
    m += tot;
  }
  int mp = 1;
  while (mp < m) mp <<= 1;
  if (m == 0) mp = 0;
  for (int i = m + tid; i < mp; i += kAccThreads) {
    Rec pad;
    pad.k0 = pad.k1 = pad.k2 = ~0ull;
    recs[i] = pad;
  }
  __syncthreads();

  // ---- 4. sort by the reference's output comparator ----
  if (mp > 1) {
    if (POSITION)
      block_bitonic_sort<Rec, RecLess<false>>(recs, mp, reinterpret_cast<Rec*>(smem), rec_tile, RecLess<false>());
    else
      block_bitonic_sort<Rec, RecLess<true>>(recs, mp, reinterpret_cast<Rec*>(smem), rec_tile, RecLess<true>());
  }
  if (tid == 0) a.num_records[u] = m;
}

// Gathers the per-lattice sorted records into the dense output (record_off computed on the host
// from num_records).  Output record = {int32 f0,f1,f2,f3; double logp} (24 B):
//   segment : label, t0, t1, 0        position : label, pos, t0, t1
template <bool POSITION>
__device__ __forceinline__ OutRecord decode_record(const Rec& r) {
  OutRecord w;
  w.logp = desc_unkey(r.k0);
  w.f0 = unbias((uint32_t)(r.k1 >> 32));
  w.f1 = unbias((uint32_t)r.k1);
  if (POSITION) {
    w.f2 = (int32_t)(uint32_t)(r.k2 >> 32);
    w.f3 = (int32_t)(uint32_t)r.k2;
  } else {
    w.f2 = unbias((uint32_t)(r.k2 >> 32));
    w.f3 = 0;
  }
  return w;
}

// A lattice the hash kernel took in pieces left one sorted run per piece: a record's place in the lattice's order is
// its place in its own run plus, for every other run, the records there that come before it -- (k0, k1[, k2]) compared
// as the kernels' sorts compare them (the position index's k2 is payload; its keys differ in k1).  No key occurs in two
// pieces, so no two records compare equal.
//
// One workgroup per BAND of the lattice's output (a lattice of 120 000 records: thirty workgroups, not one): the records
// that lie between two boundary records -- every (n / bands)-th record of the lattice's longest run -- in ALL runs.  A
// band is one range of every run (a binary search of either boundary, 2 x runs of them a workgroup) and ONE range of the
// output, which the workgroup writes whole: a record's place is its band's first place plus its place in its own range
// plus the records before it in the other runs' ranges (searches over a few hundred values, which the workgroup keeps in LDS).
// (Chunks of the INPUT instead, every record searching the whole of every other run from samples in LDS, scattered 24-byte
// records over the lattice's whole output from thirty workgroups on eight L2s: lines reached memory in pieces, and the
// configs[4]-shaped mix's 36 M records took 2.2 ms; plain binary searches from one workgroup per lattice 20 ms.)
static constexpr int kGatherChunk = 4096, kGatherRuns = 32, kGatherBand = 6144;  // (a band's values in LDS: 48 KB)
struct GatherChunk {
  int32_t u, first;  // band first / kGatherChunk of lattice u (a lattice in one run: its records [first, first + kGatherChunk))
};

template <bool POSITION>
__device__ __forceinline__ bool rec_before(const Rec& x, const Rec& r) {
  if (x.k0 != r.k0) return x.k0 < r.k0;
  return x.k1 < r.k1 || (!POSITION && x.k1 == r.k1 && x.k2 < r.k2);
}
// first record of run[lo .. hi) that does not come before r
template <bool POSITION>
__device__ __forceinline__ int rec_lower_bound(const Rec* run, int lo, int hi, const Rec& r) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rec_before<POSITION>(run[mid], r)) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

template <bool POSITION>
__global__ __launch_bounds__(256) void index_gather_kernel(AccumArgs a, const int64_t* record_off, const GatherChunk* chunks, OutRecord* out) {
  __shared__ int run_begin[kGatherRuns], run_n[kGatherRuns], band_lo[kGatherRuns], band_hi[kGatherRuns], band_at[kGatherRuns + 1];
  __shared__ uint64_t band_k0[kGatherBand];
  const GatherChunk ch = chunks[blockIdx.x];
  const int u = ch.u, tid = threadIdx.x;
  const Rec* recs = reinterpret_cast<const Rec*>(a.rec_scratch) + a.scratch_off[u];
  const int64_t o = record_off[u];
  const int m = (int)(record_off[u + 1] - o);
  const int last = min(ch.first + kGatherChunk, m);
  const int pieces = a.piece_count ? a.piece_count[u] : 0;
  if (pieces == 0) {
    for (int i = ch.first + tid; i < last; i += 256) out[o + i] = decode_record<POSITION>(recs[i]);
    return;
  }
  const IndexItem* its = a.piece_items + a.piece_first[u];
  if (pieces > kGatherRuns) {  // (workgroup-uniform; a lattice of more than 400 000 arcs) a chunk of the input, plain binary searches
    for (int f = ch.first + tid; f < last; f += 256) {
      int p = 0, j = f;
      while (j >= a.item_records[its[p].slot]) j -= a.item_records[its[p].slot], ++p;
      const Rec r = recs[its[p].begin + j];
      int pos = j;
      for (int q = 0; q < pieces; ++q)
        if (q != p) pos += rec_lower_bound<POSITION>(recs + its[q].begin, 0, a.item_records[its[q].slot], r);
      out[o + pos] = decode_record<POSITION>(r);
    }
    return;
  }
  if (tid < pieces) run_begin[tid] = its[tid].begin, run_n[tid] = a.item_records[its[tid].slot];
  __syncthreads();
  // the band's two boundary records, from the longest run; its range in every run (threads 0 .. 2 * pieces - 1)
  const int bands = (m + kGatherChunk - 1) / kGatherChunk, c = ch.first / kGatherChunk;
  int longest = 0;
  for (int q = 1; q < pieces; ++q)
    if (run_n[q] > run_n[longest]) longest = q;
  if (tid < 2 * pieces) {
    const int q = tid >> 1, side = tid & 1, edge = c + side;  // edge 0: before everything; edge `bands`: behind everything
    int at = edge == 0 ? 0 : run_n[q];
    if (edge > 0 && edge < bands) {
      const int b = (int)((int64_t)edge * run_n[longest] / bands);
      at = q == longest ? b : rec_lower_bound<POSITION>(recs + run_begin[q], 0, run_n[q], recs[run_begin[longest] + b]);
    }
    if (side) band_hi[q] = at;
    else band_lo[q] = at;
  }
  __syncthreads();
  if (tid == 0) {
    int at = 0;
    for (int q = 0; q < pieces; ++q) band_at[q] = at, at += band_hi[q] - band_lo[q];
    band_at[pieces] = at;
  }
  __syncthreads();
  int first_place = 0;  // the band's first place in the output: the records before it in every run
  for (int q = 0; q < pieces; ++q) first_place += band_lo[q];
  const int count = band_at[pieces];
  // the values (k0) of the band's records in LDS, run after run: the searches read those; two records of equal value
  // (rare: different runs hold different keys) are compared in full, from memory
  const bool staged = count <= kGatherBand;  // (workgroup-uniform; a band far larger than the others: the searches go to memory)
  if (staged) {
    for (int f = tid; f < count; f += 256) {
      int p = 0;
      while (f >= band_at[p + 1]) ++p;
      band_k0[f] = recs[run_begin[p] + band_lo[p] + f - band_at[p]].k0;
    }
    __syncthreads();
  }
  for (int f = tid; f < count; f += 256) {
    int p = 0;
    while (f >= band_at[p + 1]) ++p;
    const int j = band_lo[p] + f - band_at[p];
    const Rec r = recs[run_begin[p] + j];
    int pos = first_place + (j - band_lo[p]);
    for (int q = 0; q < pieces; ++q) {
      if (q == p) continue;
      if (!staged) {
        pos += rec_lower_bound<POSITION>(recs + run_begin[q], band_lo[q], band_hi[q], r) - band_lo[q];
        continue;
      }
      int lo = band_at[q], hi = band_at[q + 1];  // first value of the run's range not below r's
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (band_k0[mid] < r.k0) lo = mid + 1;
        else hi = mid;
      }
      pos += lo - band_at[q];
      for (int e = lo; e < band_at[q + 1] && band_k0[e] == r.k0; ++e)  // (equal values: the keys decide)
        pos += rec_before<POSITION>(recs[run_begin[q] + band_lo[q] + e - band_at[q]], r) ? 1 : 0;
    }
    out[o + pos] = decode_record<POSITION>(r);
  }
}

// label / time / length ranges of the batch (out[] preset to {INT_MAX, INT_MIN, 0, 0})
__global__ __launch_bounds__(256) void range_kernel(RangeArgs a) {
  int32_t lmin = 0x7FFFFFFF, lmax = (int32_t)0x80000000, tmax = 0, nmax = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  // the labels four a lane and load, four loads a thread in flight (one dword a trip to memory, 457 trips a thread:
  // 0.19 ms on C2 for a third of a gigabyte; the arrays start on 16-byte boundaries, the last few labels one by one)
  const int64_t quads = a.num_arcs / 4;
  const int4* const l4 = reinterpret_cast<const int4*>(a.olabel);
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += 4 * stride) {
    int4 v[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) v[x] = l4[min(q + x * stride, quads - 1)];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      lmin = min(min(lmin, v[x].x), min(min(v[x].y, v[x].z), v[x].w));
      lmax = max(max(lmax, v[x].x), max(max(v[x].y, v[x].z), v[x].w));
    }
  }
  for (int64_t i = quads * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.num_arcs; i += stride) {
    const int32_t l = a.olabel[i];
    lmin = min(lmin, l);
    lmax = max(lmax, l);
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.num_states; i += 4 * stride) {
    int32_t t[4], n[4] = {0, 0, 0, 0};
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int64_t j = min(i + x * stride, a.num_states - 1);
      t[x] = a.state_time[j];
      if (a.state_len) n[x] = a.state_len[j];
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      tmax = max(tmax, t[x] < 0 ? 0x7FFFFFFF : t[x]);  // a negative time disables the radix path
      nmax = max(nmax, n[x] < 0 ? 0x7FFFFFFF : n[x]);
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    lmin = min(lmin, __shfl_xor(lmin, d, 64));
    lmax = max(lmax, __shfl_xor(lmax, d, 64));
    tmax = max(tmax, __shfl_xor(tmax, d, 64));
    nmax = max(nmax, __shfl_xor(nmax, d, 64));
  }
  // one set of atomics per workgroup: thousands of them on the same four words are what this
  // kernel's time went into
  __shared__ int32_t part[4][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) part[wave][0] = lmin, part[wave][1] = lmax, part[wave][2] = tmax, part[wave][3] = nmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      lmin = min(lmin, part[w][0]);
      lmax = max(lmax, part[w][1]);
      tmax = max(tmax, part[w][2]);
      nmax = max(nmax, part[w][3]);
    }
    atomicMin(&a.out[0], lmin);
    atomicMax(&a.out[1], lmax);
    atomicMax(&a.out[2], tmax);
    atomicMax(&a.out[3], nmax);
  }
}

hipError_t launch_range(const RangeArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(range_kernel, dim3(1024), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_index(const AccumArgs& a, int num_lattices, bool position, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipError_t e;
  if (position) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&index_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, a.tile_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(index_kernel<true>, dim3(num_lattices), dim3(kAccThreads), (size_t)a.tile_bytes, stream, a);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&index_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, a.tile_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(index_kernel<false>, dim3(num_lattices), dim3(kAccThreads), (size_t)a.tile_bytes, stream, a);
  }
  return hipGetLastError();
}

// record_off_host: the offsets on the host (the chunk list is built from them); chunk_buf: device room for the list, at
// least index_gather_chunks(record_off_host, num_lattices) entries of 8 bytes
int64_t index_gather_chunks(const int64_t* record_off_host, int num_lattices) {
  int64_t n = 0;
  for (int u = 0; u < num_lattices; ++u) n += (record_off_host[u + 1] - record_off_host[u] + kGatherChunk - 1) / kGatherChunk;
  return n;
}
hipError_t launch_index_gather(const AccumArgs& a, int num_lattices, bool position, const int64_t* record_off, const int64_t* record_off_host,
                               void* chunk_buf, OutRecord* out, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  std::vector<GatherChunk> chunks;
  for (int u = 0; u < num_lattices; ++u)
    for (int64_t f = 0; f < record_off_host[u + 1] - record_off_host[u]; f += kGatherChunk) chunks.push_back(GatherChunk{u, (int32_t)f});
  if (chunks.empty()) return hipSuccess;
  hipError_t e = hipMemcpyAsync(chunk_buf, chunks.data(), sizeof(GatherChunk) * chunks.size(), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(stream);  // (the list is a local)
  if (e != hipSuccess) return e;
  if (position)
    hipLaunchKernelGGL(index_gather_kernel<true>, dim3((unsigned)chunks.size()), dim3(256), 0, stream, a, record_off, static_cast<const GatherChunk*>(chunk_buf), out);
  else
    hipLaunchKernelGGL(index_gather_kernel<false>, dim3((unsigned)chunks.size()), dim3(256), 0, stream, a, record_off, static_cast<const GatherChunk*>(chunk_buf), out);
  return hipGetLastError();
}

}  // namespace klat
