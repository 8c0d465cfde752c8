// klat_accum_hash.hip -- the segment / position index (a10-a12) for lattices of up to 16 384 arcs
// whose keys pack into 62 bits, without a full sort of the arcs.
//
// What the reference does per lattice (kwsbin2/lattice-word-index-segment.cc:143-174,98-126;
// -position.cc:144-187,100-127): acc[key] = LogAdd over the arcs with that key, in arc order; then
// the records sorted by (logp desc, label, t0, t1) / (logp desc, label, pos).  One workgroup owns
// one lattice, everything below lives in LDS and registers:
//
//   1. GROUP.  Every thread owns 8 or 16 arcs (strided: coalesced loads) and packs their keys
//      (label - label_min | t0 | t1) or (label - label_min | pos).  The keys go into an open-
//      addressing hash table of 32-bit slots {18-bit fingerprint, arc index of the inserter}
//      (load <= 0.5, linear probing, one ds_cmpst per insert); a fingerprint match is verified
//      against the inserter's key.  An arc that finds its key present marks the slot in a bitmap.
//      Almost every key is unique (C2: 5 713 records from 5 713 word arcs), so almost every arc is
//      a finished record right here.
//   2. FOLD.  Only the arcs of marked slots are sorted -- 32-bit (slot | arc) words, bitonic in LDS
//      -- and every run is folded in arc order with the reference's LogAdd (cutoff included; the
//      position index keeps the (t0,t1) of the first arc attaining the maximum, strict '>' as at
//      kwsbin2/lattice-word-index-position.cc:178).
//   3. ORDER.  The records are ordered by descending logp with one bucket pass instead of radix
//      passes: bucket = floor(sqrt(max - logp) * scale) (correctly rounded sqrt: monotone), counted
//      and ranked with LDS atomics, scanned, scattered as 64-bit words {upper value bits | record
//      id}; then every bucket is put in the reference's exact order -- (logp desc, label, t0, t1) on
//      the full doubles and keys: a record of a small bucket (<= 32 entries) counts the entries
//      before it; a crowded bucket (clustered or tied values) is sorted by the whole workgroup
//      (bitonic network for arbitrary lengths) and searched.  No result depends on the order in which the atomics landed.
//
// Records leave as the 24-byte {desc_key(logp), k1, k2} rows index_gather_kernel reads
// (klat_accum.hip).  Workgroups of 1 024 threads.  LOGH = 14: up to 8 192 arcs, 8 per thread, 75 KiB
// of LDS; LOGH = 15: up to 16 384 arcs, 16 per thread, 150 KiB.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cstdlib>
#include <stdint.h>

#include "klat_accum.h"

namespace klat {

namespace {

constexpr int kSerialMax = 32;   // buckets up to this size are ordered by one thread
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

__device__ __forceinline__ double hlog_add(double x, double y) {
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
__device__ __forceinline__ uint32_t hbias(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
__device__ __forceinline__ uint64_t hdesc_key(double x) {  // descending double -> ascending uint64
  uint64_t b = (uint64_t)__double_as_longlong(x);
  b = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
  return ~b;
}
__device__ __forceinline__ bool hin_sorted(const int32_t* set, int n, int32_t v) {
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

struct Payload {  // 16 B per record id, in the lattice's key scratch
  uint64_t k1, k2;
};
struct Rec {  // 24 B: what index_gather_kernel reads
  uint64_t k0, k1, k2;
};

// exclusive prefix sum of one int per thread over the block; *total = block sum
__device__ __forceinline__ int block_excl_sum(int T, int v, int* total, int* sh /* [T/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();  // sh may still be read from the previous call
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < T / 64; ++w) {
    const int s = sh[w];
    if (w < wave) off += s;
    tot += s;
  }
  *total = tot;
  return off + incl - v;
}

template <bool POSITION>
__device__ __forceinline__ bool arc_key(const AccumArgs& a, const RadixPack& pk, int64_t a0, int64_t s0, int i, uint64_t* key) {
  const int32_t label = a.olabel[a0 + i];
  const bool valid = label != 0 && ((a.num_include > 0 && hin_sorted(a.include, a.num_include, label)) ||
                                    (a.num_include == 0 && !hin_sorted(a.exclude, a.num_exclude, label)));
  if (!valid) return false;
  const uint64_t l = (uint64_t)(uint32_t)(label - pk.label_min);
  const int32_t s = a.src[a0 + i];
  if (POSITION) {
    *key = (l << pk.time_bits) | (uint64_t)(uint32_t)a.state_len[s0 + s];
  } else {
    const int32_t d = a.dst[a0 + i];
    *key = (((l << pk.time_bits) | (uint64_t)(uint32_t)a.state_time[s0 + s]) << pk.time_bits) |
           (uint64_t)(uint32_t)a.state_time[s0 + d];
  }
  return true;
}

// the record fields of a key; for the position index (t0,t1) come from `seg_arc`
template <bool POSITION>
__device__ __forceinline__ Payload make_payload(const AccumArgs& a, const RadixPack& pk, int64_t a0, int64_t s0, uint64_t key, int seg_arc) {
  const uint64_t tmask = (1ull << pk.time_bits) - 1ull;
  Payload q;
  if (POSITION) {
    const int32_t label = (int32_t)(uint32_t)(key >> pk.time_bits) + pk.label_min;
    const int32_t pos = (int32_t)(uint32_t)(key & tmask) + 1;  // 1-based on output
    q.k1 = ((uint64_t)hbias(label) << 32) | hbias(pos);
    const int32_t t0 = a.state_time[s0 + a.src[a0 + seg_arc]];
    const int32_t t1 = a.state_time[s0 + a.dst[a0 + seg_arc]];
    q.k2 = ((uint64_t)(uint32_t)t0 << 32) | (uint32_t)t1;
  } else {
    const int32_t label = (int32_t)(uint32_t)(key >> (2 * pk.time_bits)) + pk.label_min;
    const int32_t t0 = (int32_t)(uint32_t)((key >> pk.time_bits) & tmask);
    const int32_t t1 = (int32_t)(uint32_t)(key & tmask);
    q.k1 = ((uint64_t)hbias(label) << 32) | hbias(t0);
    q.k2 = (uint64_t)hbias(t1) << 32;
  }
  return q;
}

// Sorting network for ANY length c (bitonic with mirrored first step: all comparators ascending, so
// skipping the comparators that reach past c equals padding with +infinity), by the whole workgroup.
template <class V, class Less>
__device__ __forceinline__ void block_sort_any(int T, V* v, int c, Less less) {
  int P = 1;
  while (P < c) P <<= 1;
  for (int k = 2; k <= P; k <<= 1) {
    for (int i = threadIdx.x; i < c; i += T) {  // flip: partner = mirror inside the block of k
      const int l = i ^ (k - 1);
      if (l > i && l < c) {
        const V x = v[i], y = v[l];
        if (less(y, x)) v[i] = y, v[l] = x;
      }
    }
    __syncthreads();
    for (int j = k >> 2; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < c; i += T) {
        const int l = i ^ j;
        if (l > i && l < c) {
          const V x = v[i], y = v[l];
          if (less(y, x)) v[i] = y, v[l] = x;
        }
      }
      __syncthreads();
    }
  }
}

// value (full double image) and key of the record that arc `i` stands for: `vals` holds every arc's
// normalised through-likelihood, the folded sum in place of a repeated key's first arc
template <bool POSITION>
__device__ __forceinline__ void record_of(const AccumArgs& a, const RadixPack& pk, int64_t a0, int64_t s0, const double* vals, int i,
                                          uint64_t* dkey, uint64_t* key) {
  *dkey = hdesc_key(vals[i]);
  (void)arc_key<POSITION>(a, pk, a0, s0, i, key);
}

// TWO: the instantiation for launches whose lattices fit 768 threads (C2: 6 000 arcs): at most 80 VGPRs and the
// arcs' values in the scratch row instead of LDS, so that TWO workgroups share a CU and one's barriers and LDS
// round trips hide behind the other's work (the phases are chains of dependent LDS / L2 round trips: one
// workgroup per CU left the CU waiting most of the time).
template <bool POSITION, int LOGH, bool TWO>
__global__ __launch_bounds__(TWO ? 768 : 1024) __attribute__((amdgpu_waves_per_eu(TWO ? 6 : 4, TWO ? 6 : 4)))
void index_hash_kernel(AccumArgs a, RadixPack pk) {
  // The phases are bound by LDS / L2 latency, not by arithmetic, and 16 waves per CU is what the
  // register budget allows either way: one workgroup of 16 waves measured the same as two of 8
  // (2.5 ms on C2) and keeps the per-thread arrays short.
  // The workgroup is as wide as the launch's largest lattice needs (512 .. 1 024 threads in whole waves, kPerThread
  // arcs each): C2's 6 000 arcs take 12 waves, not 16 -- a quarter of the instructions and of every barrier gone.
  const int T = (int)blockDim.x;
  constexpr int H = 1 << LOGH;
  constexpr int MAXN = H / 2;
  constexpr int kPerThread = MAXN / 1024;  // arcs (and records) per thread: 8 or 16
  constexpr int RB = LOGH - 1;          // bits of an arc index (= record id)
  constexpr int B = MAXN / 2;           // value buckets
  static_assert(MAXN * 2 == H, "load factor 0.5");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* table = reinterpret_cast<uint32_t*>(smem);           // [H]        phase 1
  uint32_t* dups = reinterpret_cast<uint32_t*>(smem);            // [<= MAXN]  phase 2 (over the table)
  uint64_t* order = reinterpret_cast<uint64_t*>(smem);           // [<= MAXN]  phase 3 (over both)
  uint32_t* marked = reinterpret_cast<uint32_t*>(smem + (size_t)H * 4);      // [H / 32] slots with a repeated key
  uint32_t* headbit = marked;                                    // [MAXN / 32] later: first arcs of repeated keys
  uint32_t* bcnt = marked + H / 32;                              // [B / 2] packed 16-bit counters -> inclusive scan
  // every arc's value alpha[src] - cost + beta[dst] - total, formed once while its ends are in registers
  // (phase 1) and read by the later phases: in LDS behind the other arrays where the CU has room for it
  // (up to 8 192 arcs: 64 KiB), else in the lattice's scratch row (L2)
  constexpr bool kValsInLds = LOGH <= 14 && !TWO;
  __shared__ int sh[1024 / 64];
  __shared__ unsigned long long sh_dmin, sh_dmax;
  __shared__ double sh_vmin, sh_vmax;
  __shared__ int sh_nbig, sh_big[MAXN / kSerialMax + 1];

  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t a0 = a.arc_off[u], s0 = a.state_off[u];
  const int64_t pb = padded_state_base(s0, u);
  const int n = (int)(a.arc_off[u + 1] - a0);
  if (n <= a.min_arcs || n > a.max_arcs) return;  // another launch's lattice (workgroup-uniform)
  const double tot = a.total[u];
  double* vals = kValsInLds ? reinterpret_cast<double*>(smem + (size_t)H * 4 + H / 8 + (size_t)B * 2) : a.val_scratch + a.scratch_off[u];
  Payload* payload = reinterpret_cast<Payload*>(a.key_scratch) + a.scratch_off[u];  // by first arc, repeated keys only
  Rec* recs = reinterpret_cast<Rec*>(a.rec_scratch) + a.scratch_off[u];
#define KLAT_STAMP(k)                                                                 \
  do {                                                                                \
    if (a.phase_clock && tid == 0) a.phase_clock[(size_t)u * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
  KLAT_STAMP(0);
  if (n == 0) {
    if (tid == 0) a.num_records[u] = 0;
    return;
  }

  // ---- 1. keys (loads batched, no branches around them), hash grouping ----
  for (int i = tid; i < H; i += T) table[i] = kEmpty;
  for (int i = tid; i < H / 32; i += T) marked[i] = 0;
  uint64_t key[kPerThread];
  uint32_t valid = 0;
  const bool filtered = (a.num_include | a.num_exclude) != 0;  // workgroup-uniform
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    int32_t lab[kPerThread / 2], t0[kPerThread / 2], t1[kPerThread / 2];
    float gr[kPerThread / 2], ac[kPerThread / 2];
    bool word[kPerThread / 2];
    double thr[kPerThread / 2];
#pragma unroll
    for (int q = 0; q < kPerThread / 2; ++q) {
      const int i = min((half * (kPerThread / 2) + q) * T + tid, n - 1);
      lab[q] = a.olabel[a0 + i];
      t0[q] = a.src[a0 + i];
      t1[q] = a.dst[a0 + i];
      gr[q] = a.graph[a0 + i];
      ac[q] = a.acoustic[a0 + i];
      word[q] = a.sp.penalized ? a.ilabel[a0 + i] != 0 : false;
    }
#pragma unroll
    for (int q = 0; q < kPerThread / 2; ++q) {
      // fw[src] + bw[dst] - cost (kwsbin2/lattice-word-index-segment.cc:160-163), the posterior pass's
      // expression: the accumulator needs no per-arc posteriors from anybody
      thr[q] = a.alpha[pb + t0[q]] + a.beta[pb + t1[q]] - arc_cost(gr[q], ac[q], word[q], a.sp);
      t0[q] = POSITION ? a.state_len[s0 + t0[q]] : a.state_time[s0 + t0[q]];
      if (!POSITION) t1[q] = a.state_time[s0 + t1[q]];
    }
#pragma unroll
    for (int q = 0; q < kPerThread / 2; ++q) {
      const int j = half * (kPerThread / 2) + q;
      const int i = j * T + tid;
      if (i < n) vals[i] = accum_normalise(thr[q], tot);
      bool ok = i < n && lab[q] != 0;
      if (filtered && ok)
        ok = a.num_include > 0 ? hin_sorted(a.include, a.num_include, lab[q]) : !hin_sorted(a.exclude, a.num_exclude, lab[q]);
      const uint64_t l = (uint64_t)(uint32_t)(lab[q] - pk.label_min);
      key[j] = POSITION ? ((l << pk.time_bits) | (uint64_t)(uint32_t)t0[q])
                        : ((((l << pk.time_bits) | (uint64_t)(uint32_t)t0[q]) << pk.time_bits) | (uint64_t)(uint32_t)t1[q]);
      if (ok) valid |= 1u << j;
    }
  }
  if (!kValsInLds) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (vals in the scratch row: read by other threads below)
  __syncthreads();
  if (!kValsInLds) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  KLAT_STAMP(1);
  // optimistic insert of every key at its home slot, all compare-and-swaps in flight at once
  uint32_t slot[kPerThread], seen[kPerThread];
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    const uint64_t hsh = key[j] * 0x9E3779B97F4A7C15ull;
    slot[j] = (uint32_t)(hsh >> (64 - LOGH));
    const uint32_t mine = ((uint32_t)(hsh >> (64 - LOGH - 18)) << 14) | (uint32_t)(j * T + tid);
    seen[j] = kEmpty;
    if (valid & (1u << j)) seen[j] = atomicCAS(&table[slot[j]], kEmpty, mine);
  }
  // three more batches for the arcs whose slot was taken by another key's fingerprint: the next slots
#pragma unroll
  for (int round = 0; round < 3; ++round) {
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
      const uint64_t hsh = key[j] * 0x9E3779B97F4A7C15ull;
      const uint32_t fp = (uint32_t)(hsh >> (64 - LOGH - 18)) & 0x3FFFFu;
      if (seen[j] != kEmpty && (seen[j] >> 14) != fp) {
        slot[j] = (slot[j] + 1) & (H - 1);
        seen[j] = atomicCAS(&table[slot[j]], kEmpty, (fp << 14) | (uint32_t)(j * T + tid));
      }
    }
  }
  // the few left: linear probing from there; a fingerprint match is checked against the inserter's
  // key (another arc of the same key marks the slot)
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    if (seen[j] == kEmpty) continue;  // inserted (or not a record)
    const uint64_t hsh = key[j] * 0x9E3779B97F4A7C15ull;
    const uint32_t fp = (uint32_t)(hsh >> (64 - LOGH - 18)) & 0x3FFFFu;
    const uint32_t mine = (fp << 14) | (uint32_t)(j * T + tid);
    uint32_t h = slot[j], v = seen[j];
    for (int probe = 0; probe < H; ++probe) {
      if (v == kEmpty) {
        v = atomicCAS(&table[h], kEmpty, mine);
        if (v == kEmpty) break;  // inserted: this arc represents its key
      }
      if ((v >> 14) == fp) {
        uint64_t other = 0;
        (void)arc_key<POSITION>(a, pk, a0, s0, (int)(v & 0x3FFFu), &other);
        if (other == key[j]) {
          atomicOr(&marked[h >> 5], 1u << (h & 31));
          break;
        }
      }
      h = (h + 1) & (H - 1);
      v = table[h];
    }
    slot[j] = h;
  }
  __syncthreads();
  KLAT_STAMP(2);
  uint32_t dupmask = 0;
  int ndup = 0;
#pragma unroll
  for (int j = 0; j < kPerThread; ++j)
    if ((valid & (1u << j)) && (marked[slot[j] >> 5] & (1u << (slot[j] & 31)))) dupmask |= 1u << j, ++ndup;
  int tot_dup;
  int dpos = block_excl_sum(T, ndup, &tot_dup, sh);
  __syncthreads();  // every thread has read `marked` and the table: the regions become `headbit` and `dups`
  for (int i = tid; i < MAXN / 32; i += T) headbit[i] = 0;
  uint32_t recmask = valid & ~dupmask;  // the arcs that stand for a record: unique keys, plus first arcs of repeated keys
  KLAT_STAMP(3);

  // ---- 2. the arcs of repeated keys: sort (slot | arc), fold every run in arc order ----
  if (tot_dup > 0) {  // workgroup-uniform
#pragma unroll
    for (int j = 0; j < kPerThread; ++j)
      if (dupmask & (1u << j)) dups[dpos++] = (slot[j] << 14) | (uint32_t)(j * T + tid);
    __syncthreads();
    block_sort_any(T, dups, tot_dup, [](uint32_t x, uint32_t y) { return x < y; });
    for (int p = tid; p < tot_dup; p += T) {
      const uint32_t w = dups[p];
      if (!(p == 0 || (dups[p - 1] >> 14) != (w >> 14))) continue;
      int arc = (int)(w & 0x3FFFu);
      const int first_arc = arc;
      double acc = vals[arc], best = acc;
      int best_arc = arc;
      for (int q = p + 1; q < tot_dup && (dups[q] >> 14) == (w >> 14); ++q) {
        arc = (int)(dups[q] & 0x3FFFu);
        const double x = vals[arc];
        acc = hlog_add(acc, x);
        if (x > best) {  // strict: the first maximum in arc order wins
          best = x;
          best_arc = arc;
        }
      }
      vals[first_arc] = acc;  // (runs are disjoint: nobody else reads this arc's own value any more)
      if (POSITION) {
        uint64_t k = 0;
        (void)arc_key<POSITION>(a, pk, a0, s0, first_arc, &k);
        payload[first_arc] = make_payload<POSITION>(a, pk, a0, s0, k, best_arc);
      }
      atomicOr(&headbit[first_arc >> 5], 1u << (first_arc & 31));
    }
    // the staged values are read back by other threads of this workgroup: workgroup scope is enough
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
      const int i = j * T + tid;
      if ((dupmask & (1u << j)) && (headbit[i >> 5] & (1u << (i & 31)))) recmask |= 1u << j;
    }
  }
  if (tid == 0) {
    sh_dmin = ~0ull, sh_dmax = 0ull, sh_nbig = 0;
    sh_vmin = INFINITY, sh_vmax = -INFINITY;
  }
  for (int i = tid; i < B / 2; i += T) bcnt[i] = 0;
  __syncthreads();  // `dups` is dead: the region becomes `order`
  KLAT_STAMP(4);

  // ---- 3. order by (logp desc, key) ----
  double val[kPerThread];
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) val[j] = vals[min(j * T + tid, n - 1)];  // (a repeated key's first arc holds the folded sum)
  int m;
  {
    unsigned long long dlo = ~0ull, dhi = 0ull;
    double vlo = INFINITY, vhi = -INFINITY;
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
      if (!(recmask & (1u << j))) continue;
      const unsigned long long d = hdesc_key(val[j]);
      dlo = min(dlo, d), dhi = max(dhi, d);
      if (val[j] > -INFINITY && val[j] < INFINITY) vlo = fmin(vlo, val[j]), vhi = fmax(vhi, val[j]);
    }
    for (int d = 32; d > 0; d >>= 1) {
      dlo = min(dlo, (unsigned long long)__shfl_xor(dlo, d, 64));
      dhi = max(dhi, (unsigned long long)__shfl_xor(dhi, d, 64));
      vlo = fmin(vlo, __shfl_xor(vlo, d, 64));
      vhi = fmax(vhi, __shfl_xor(vhi, d, 64));
    }
    if ((tid & 63) == 0) {
      if (dlo <= dhi) atomicMin(&sh_dmin, dlo), atomicMax(&sh_dmax, dhi);
      if (vlo <= vhi) {  // finite doubles through two small compare-and-swap loops
        unsigned long long* pmin = reinterpret_cast<unsigned long long*>(&sh_vmin);
        unsigned long long* pmax = reinterpret_cast<unsigned long long*>(&sh_vmax);
        unsigned long long old = *pmin;
        while (__longlong_as_double((long long)old) > vlo) {
          const unsigned long long got = atomicCAS(pmin, old, (unsigned long long)__double_as_longlong(vlo));
          if (got == old) break;
          old = got;
        }
        old = *pmax;
        while (__longlong_as_double((long long)old) < vhi) {
          const unsigned long long got = atomicCAS(pmax, old, (unsigned long long)__double_as_longlong(vhi));
          if (got == old) break;
          old = got;
        }
      }
    }
    (void)block_excl_sum(T, __popc(recmask), &m, sh);  // (its barriers also publish the minima / maxima)
  }
  __syncthreads();
  const uint64_t dmin = m > 0 ? sh_dmin : 0ull;
  const uint64_t drange = m > 0 ? sh_dmax - sh_dmin : 0ull;
  const int hb = drange ? 64 - __clzll((long long)drange) : 0;
  const int shift = hb > 64 - RB ? hb - (64 - RB) : 0;  // value bits that do not fit beside the arc index
  const double vmax = sh_vmax, vspan = sh_vmax - sh_vmin;
  const float scale = (vspan > 0.0 && vspan < INFINITY) ? (float)(B - 1) / __fsqrt_rn((float)vspan) : 0.0f;
  uint32_t where[kPerThread];  // bucket << 16 | arrival number inside it
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    where[j] = 0;
    if (recmask & (1u << j)) {
      int b = B - 1;  // -inf, NaN: behind every finite value (their images order them inside the bucket)
      if (val[j] > -INFINITY && val[j] < INFINITY) {
        // monotone in -logp: double -> float rounding, correctly rounded float sqrt, multiply, truncate
        const float t = __fsqrt_rn((float)(vmax - val[j])) * scale;
        b = t >= (float)(B - 1) ? B - 1 : (int)t;
      } else if (val[j] == INFINITY) {
        b = 0;
      }
      const uint32_t old = atomicAdd(&bcnt[b >> 1], 1u << ((b & 1) * 16));
      where[j] = ((uint32_t)b << 16) | ((old >> ((b & 1) * 16)) & 0xFFFFu);
    }
  }
  __syncthreads();
  {  // in-place inclusive scan of the B 16-bit counters: PER consecutive ones per thread, by the first B / PER threads
    constexpr int PER = B / 512;  // 8 or 16 (a workgroup has at least 512 threads)
    static_assert(PER % 2 == 0, "");
    uint32_t w[PER / 2];
    int sum = 0;
    const bool scans = tid < B / PER;
#pragma unroll
    for (int q = 0; q < PER / 2; ++q) {
      w[q] = scans ? bcnt[tid * (PER / 2) + q] : 0u;
      sum += (int)(w[q] & 0xFFFFu) + (int)(w[q] >> 16);
    }
    int total;
    int run = block_excl_sum(T, sum, &total, sh);
#pragma unroll
    for (int q = 0; q < PER / 2; ++q) {
      const int lo = run + (int)(w[q] & 0xFFFFu);
      const int hi = lo + (int)(w[q] >> 16);
      run = hi;
      if (scans) bcnt[tid * (PER / 2) + q] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
  }
  __syncthreads();
  const uint16_t* bend = reinterpret_cast<const uint16_t*>(bcnt);  // bend[b] = end of bucket b
  const uint64_t idmask = (1ull << RB) - 1ull;
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    if (recmask & (1u << j)) {
      const int b = (int)(where[j] >> 16);
      const int pos = (b ? (int)bend[b - 1] : 0) + (int)(where[j] & 0xFFFFu);
      order[pos] = (((uint64_t)hdesc_key(val[j]) - dmin) >> shift) << RB | (uint64_t)(j * T + tid);
    }
  }
  __syncthreads();
  KLAT_STAMP(5);
  // Exact order: (logp desc, label, t0, t1) / (logp desc, label, pos) on the full doubles and keys.
  // Two words whose upper parts differ are ordered by them; equal upper parts (values within
  // 2^-(64-RB) of the lattice's range, or equal) go back to the records themselves.
  auto less = [&](uint64_t x, uint64_t y) -> bool {
    if ((x >> RB) != (y >> RB)) return x < y;
    uint64_t dx, kx, dy, ky;
    record_of<POSITION>(a, pk, a0, s0, vals, (int)(x & idmask), &dx, &kx);
    record_of<POSITION>(a, pk, a0, s0, vals, (int)(y & idmask), &dy, &ky);
    return dx != dy ? dx < dy : kx < ky;
  };
  // crowded buckets (clustered or tied values) are sorted by the whole workgroup, in place
  for (int b = tid; b < B; b += T)
    if ((int)bend[b] - (b ? (int)bend[b - 1] : 0) > kSerialMax) sh_big[atomicAdd(&sh_nbig, 1)] = b;
  __syncthreads();
  const int nbig = sh_nbig;  // workgroup-uniform
  for (int g = 0; g < nbig; ++g) {
    const int b = sh_big[g];
    const int lo = b ? (int)bend[b - 1] : 0, hi = (int)bend[b];
    block_sort_any(T, order + lo, hi - lo, less);
  }
  // every record finds its final position inside its bucket: by counting the smaller entries of a
  // small bucket (independent LDS reads, no dependent chain), by binary search in a sorted crowded one
  uint32_t finalpos[kPerThread / 2];  // two 16-bit positions per word
#pragma unroll
  for (int j = 0; j < kPerThread; ++j) {
    int p = 0;
    if (recmask & (1u << j)) {
      const int b = (int)(where[j] >> 16);
      const int lo = b ? (int)bend[b - 1] : 0, hi = (int)bend[b];
      const uint64_t me = (((uint64_t)hdesc_key(val[j]) - dmin) >> shift) << RB | (uint64_t)(j * T + tid);
      if (hi - lo > kSerialMax) {
        int l = lo, h = hi;  // first entry not less than me
        while (l < h) {
          const int mid = (l + h) >> 1;
          if (less(order[mid], me)) l = mid + 1;
          else h = mid;
        }
        p = l;
      } else {
        p = lo;
        for (int q = lo; q < hi; ++q) {
          const uint64_t f = order[q];
          p += ((f >> RB) != (me >> RB)) ? (f < me ? 1 : 0) : ((f != me && less(f, me)) ? 1 : 0);
        }
      }
    }
    if (j & 1) finalpos[j >> 1] |= (uint32_t)p << 16;
    else finalpos[j >> 1] = (uint32_t)p;
  }
  KLAT_STAMP(6);

  // ---- 4. the records at their final positions: through LDS (the sort's array is free now), so
  // that they leave as full lines -- as scattered 24-byte stores they cost 1.8 x their bytes in HBM
  // writes and a store instruction per record and thread ----
  __syncthreads();  // every thread is done with `order` (and with the counters behind it)
  {
    // (only LDS has to be ordered between the passes: __syncthreads() would also wait for the
    // record stores of the pass before, a full HBM write latency each time)
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    constexpr int kStageRecs = (H * 4 + H / 8 + B * 2) / (int)sizeof(Rec);  // the whole dynamic LDS of the kernel
    uint64_t* stage = reinterpret_cast<uint64_t*>(smem);  // [kStageRecs][3]
    uint64_t* out64 = reinterpret_cast<uint64_t*>(recs);
    for (int base = 0; base < m; base += kStageRecs) {  // (m is workgroup-uniform)
#pragma unroll
      for (int j = 0; j < kPerThread; ++j) {
        if (!(recmask & (1u << j))) continue;
        const int p = (int)((finalpos[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu) - base;
        if ((unsigned)p >= (unsigned)kStageRecs) continue;
        const int i = j * T + tid;
        const Payload q = (POSITION && (dupmask & (1u << j))) ? payload[i]  // (t0,t1) of the first arc attaining the maximum
                                                              : make_payload<POSITION>(a, pk, a0, s0, key[j], i);
        stage[p * 3 + 0] = hdesc_key(val[j]);
        stage[p * 3 + 1] = q.k1;
        stage[p * 3 + 2] = q.k2;
      }
      lds_barrier();
      const int cnt = min(kStageRecs, m - base) * 3;
      for (int i = tid; i < cnt; i += T) out64[(size_t)base * 3 + i] = stage[i];
      lds_barrier();
    }
  }
  if (tid == 0) a.num_records[u] = m;
  KLAT_STAMP(7);
#undef KLAT_STAMP
}

template <bool POSITION, int LOGH>
hipError_t launch_hash_one(const AccumArgs& a, const RadixPack& pk, int num_lattices, int max_arcs, hipStream_t stream) {
  constexpr int H = 1 << LOGH;
  constexpr int B = H / 4;
  constexpr int kPerThread = (H / 2) / 1024;
  // whole waves for the launch's largest lattice, 512 threads at least (the counters' scan), 1 024 at most
  int T = ((max_arcs + kPerThread - 1) / kPerThread + 63) / 64 * 64;
  T = T < 512 ? 512 : T > 1024 ? 1024 : T;
  const bool two = LOGH <= 14 && T <= 768 && std::getenv("KLAT_INDEX_ONE_PER_CU") == nullptr;
  const size_t lds = (size_t)H * 4 + H / 8 + (size_t)B * 2 + (LOGH <= 14 && !two ? (size_t)(H / 2) * sizeof(double) : 0);  // + the arcs' values
  if (two) {
    if constexpr (LOGH <= 14) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&index_hash_kernel<POSITION, LOGH, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((index_hash_kernel<POSITION, LOGH, true>), dim3(num_lattices), dim3(T), lds, stream, a, pk);
    }
    return hipGetLastError();
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&index_hash_kernel<POSITION, LOGH, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((index_hash_kernel<POSITION, LOGH, false>), dim3(num_lattices), dim3(T), lds, stream, a, pk);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_index_hash(const AccumArgs& a, const RadixPack& pk, int num_lattices, bool position, int log_slots, int max_arcs,
                             hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  if (log_slots == 14)
    return position ? launch_hash_one<true, 14>(a, pk, num_lattices, max_arcs, stream) : launch_hash_one<false, 14>(a, pk, num_lattices, max_arcs, stream);
  if (log_slots == 15)
    return position ? launch_hash_one<true, 15>(a, pk, num_lattices, max_arcs, stream) : launch_hash_one<false, 15>(a, pk, num_lattices, max_arcs, stream);
  return hipErrorInvalidValue;
}

}  // namespace klat
