// klat_accum_hash.hip -- the segment / position index (a10-a12) for lattices of up to 16 384 arcs whose keys pack
// into 62 bits, without a full sort of the arcs.
//
// What the reference does per lattice (kwsbin2/lattice-word-index-segment.cc:143-174,98-126;
// -position.cc:144-187,100-127): acc[key] = LogAdd over the arcs with that key, in arc order; then the records
// sorted by (logp desc, label, t0, t1) / (logp desc, label, pos).  One workgroup owns one lattice; per arc the
// registers hold the packed key (label - label_min | t0 | t1) or (label - label_min | pos) and the value
// alpha[src] - cost + beta[dst] - total, everything else lives in LDS:
//
//   1. GROUP.  The keys go into an open-addressing hash table of 32-bit slots {17-bit fingerprint, arc of the
//      inserter} at load <= 0.5: the hash computed once per arc (24-bit pieces, full-rate multiplies, a 32-bit
//      finaliser), three rounds of compare-and-swap with a round's swaps in flight together.  Everything that is
//      not "the slot was free" -- a slot taken by another key after three tries, a fingerprint that matches -- goes
//      to a QUEUE (the lattice's scratch row) that the workgroup drains afterwards in one compact loop: linear
//      probing, a fingerprint match verified against the inserter's key; an arc that finds its key present marks
//      itself and the inserter in a bitmap over the ARCS and puts (inserter | arc) on a list.  Almost every key of
//      the synthetic lattices is unique (C2: 5 713 records from 5 713 word arcs): those skip step 2 and its barriers.
//   2. FOLD.  The list is grouped by a counting sort over the inserters (three 16-bit arrays in the dead table);
//      the thread that meets an inserter's own word orders the key's handful of arcs and folds them in arc order
//      with the reference's LogAdd (cutoff included; the position index keeps the (t0, t1) of the first arc
//      attaining the maximum, strict '>' as at kwsbin2/lattice-word-index-position.cc:178).  A key of more than 64
//      arcs sends the lattice's list through the workgroup's bitonic network instead.  Only the members of
//      repeated keys put their values in the scratch row.
//   3. ORDER.  One bucket pass by value instead of radix passes: bucket = floor(sqrt(max - logp) * scale) (every
//      step monotone), counted and ranked with LDS atomics on packed 16-bit counters, scanned in place, scattered as
//      64-bit order words {image of the value, upper 50 bits | arc}.  A record finds its place in its bucket by
//      counting the words below every word with its upper part and those up to the last such word (independent LDS
//      reads); only if another record's value agrees in the upper 50 bits, or the bucket holds more than 32
//      records (clustered or tied values: sorted by the whole workgroup), the exact comparison on the full doubles
//      and keys runs, out of line, recomputing the other record's value from alpha / beta.  No result depends on
//      the order in which the atomics landed.
//   4. The 24-byte records {desc_key(logp), k1, k2} (what index_gather_kernel reads, klat_accum.hip) go to their
//      final rows through LDS and leave as full lines.
//
// Size classes (kHashClassArcs, klat_accum.h): a lattice runs in the smallest workgroup that holds its arcs --
// 128 threads x 12 arcs and 17 KiB of LDS (eight workgroups a CU) ... 512 x 12 and 67 KiB (two a CU: C2's 5 997
// arcs) ... 1 024 x 16 and 150 KiB -- each class one launch over its own list of lattices; the per-lattice table is
// the smallest power of two holding twice the lattice's arcs.
//
// Round 4 rewrote this kernel.  What the first form cost, measured (profiles/r03_accum_sq_counters.txt and its
// ISA): 3 200 vector instructions per wave for eight arcs a thread, 15 000 lines of code per instantiation (more
// than the instruction cache two CUs share: a fifth of the wave-cycles waited for instructions), 80 registers at
// two workgroups a CU with 43 of them spilled and every arc's value written to and read from a scratch row (2.5 x
// the algorithmic bytes), one 1 024-thread workgroup shape for every lattice.  The rare paths now exist once in
// the code (the queue's loop, noinline functions over a lattice view in LDS), not once per unrolled arc; 512
// threads x 12 arcs fit 128 registers without spills.  C2: 1.52 -> 0.90 ms; 20 000 lattices of 500 states: 1.28 ->
// 0.40 ms; configs[3]'s shard (12 500 x 14 997 arcs): 6.1 -> 4.3 ms; records bit-identical to the first form's
// (compared on six shapes before it was removed).  Dead ends, measured: 768 x 8 and 1 024 x 6 arcs (80 / 64
// registers: 70 - 135 spilled, 1.6 and 2.2 ms), every arc's ends loaded ahead of the rest (74 spilled, 1.29 ms),
// staggering the first workgroups' start (no change: they drift apart by themselves), 1 .. 4 inline rounds (3).
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <stdint.h>

#include "klat_accum.h"
#include "klat_slice.h"

namespace klat {

namespace {

constexpr int kSerialMax = 32;   // buckets up to this size are ordered by one thread

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


constexpr uint32_t kFree = 0xFFFFFFFFu;  // {fingerprint 17 bits | arc 15 bits}: arc 32 767 does not exist
#ifndef KLAT_IDX_ROUNDS
#define KLAT_IDX_ROUNDS 3
#endif
constexpr int kRounds = KLAT_IDX_ROUNDS;  // tries before an arc goes to the queue

__device__ __forceinline__ uint32_t key_hash32(uint64_t key) {
  const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  const uint32_t p0 = lo & 0xFFFFFFu, p1 = ((lo >> 24) | (hi << 8)) & 0xFFFFFFu, p2 = hi >> 16;
  uint32_t h = __umul24(p0, 0x9E3779u);
  h += __umul24(p1, 0x85EBCBu);
  h += __umul24(p2, 0xC2B2AFu);
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  return h;
}

// What the out-of-line paths need of the lattice: in LDS, written by the workgroup's first thread (they run for a
// handful of arcs; handed over by value the view cost thirty registers and a stack frame at every call site).
struct LatView {
  const int32_t* olabel;  // the lattice's first arc
  const int32_t* src;
  const int32_t* dst;
  const int32_t* ilabel;
  const float* graph;
  const float* acoustic;
  const int32_t* tm;    // the lattice's first state: state_len (position index) or state_time
  const int32_t* time;  // state_time
  const double* alpha;  // padded rows
  const double* beta;
  const double* vals;   // scratch row: the values of the members of repeated keys (a first arc: the folded sum)
  const int32_t* include;
  const int32_t* exclude;
  double tot;
  ScaleParams sp;
  int32_t label_min, time_bits;
  int32_t num_include, num_exclude;
  uint32_t dupbit_off;  // byte offset of the bitmap of repeated keys' arcs in the dynamic LDS
};
__shared__ LatView lat_view;
extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];

template <bool POSITION>
__device__ __noinline__ uint64_t slow_key(int arc) {
  const LatView& v = lat_view;
  const uint64_t l = (uint64_t)(uint32_t)(v.olabel[arc] - v.label_min);
  const int32_t s = v.src[arc];
  if (POSITION) return (l << v.time_bits) | (uint64_t)(uint32_t)v.tm[s];
  return (((l << v.time_bits) | (uint64_t)(uint32_t)v.tm[s]) << v.time_bits) | (uint64_t)(uint32_t)v.tm[v.dst[arc]];
}

// image (descending order) of the value record `arc` carries: the folded sum of a repeated key, else the arc's own
__device__ __noinline__ uint64_t slow_image(int arc) {
  const LatView& v = lat_view;
  const uint32_t* dupbit = reinterpret_cast<const uint32_t*>(smem2 + v.dupbit_off);
  if ((dupbit[arc >> 5] >> (arc & 31)) & 1u) return hdesc_key(v.vals[arc]);
  const int32_t s = v.src[arc], d = v.dst[arc];
  const bool word = v.sp.penalized ? v.ilabel[arc] != 0 : false;
  const double thr = v.alpha[s] + v.beta[d] - arc_cost(v.graph[arc], v.acoustic[arc], word, v.sp);
  return hdesc_key(accum_normalise(thr, v.tot));
}

__device__ __noinline__ bool filter_keeps(int32_t label) {
  const LatView& v = lat_view;
  return v.num_include > 0 ? hin_sorted(v.include, v.num_include, label) : !hin_sorted(v.exclude, v.num_exclude, label);
}

// exact comparison of two order words with equal upper parts: (logp desc, label, t0, t1) / (logp desc, label, pos)
// on the full doubles and keys, recomputed from the arcs
template <bool POSITION>
__device__ __noinline__ bool tie_less(int ax, int ay) {
  const uint64_t dx = slow_image(ax), dy = slow_image(ay);
  if (dx != dy) return dx < dy;
  return slow_key<POSITION>(ax) < slow_key<POSITION>(ay);
}

// position of order word `me` in its bucket [lo, hi) where the plain count does not do: a crowded bucket (sorted by
// the workgroup beforehand: binary search) or one that holds words whose upper parts equal mine
// (my_image, my_key: the calling record's own full image and key -- it has them in registers; only the other record of a
// tie is recomputed from memory)
template <bool POSITION, int RB>
__device__ __noinline__ int slow_position(const uint64_t* order, int lo, int hi, uint64_t me, bool plain, uint64_t my_image, uint64_t my_key) {
  constexpr uint64_t idmask = (1ull << RB) - 1ull;
  auto less = [&](uint64_t x, uint64_t y) -> bool {  // y == me.  (plain: the last bucket's words are the records' keys themselves)
    if (plain) return x < y;
    if ((x ^ y) & ~idmask) return x < y;
    if (x == y) return false;
    const uint64_t dx = slow_image((int)(x & idmask));
    return dx != my_image ? dx < my_image : slow_key<POSITION>((int)(x & idmask)) < my_key;
  };
  if (hi - lo > kSerialMax) {
    int l = lo, h = hi;  // first entry not less than me
    while (l < h) {
      const int mid = (l + h) >> 1;
      if (less(order[mid], me)) l = mid + 1;
      else h = mid;
    }
    return l;
  }
  int p = lo;
  for (int q = lo; q < hi; ++q) p += less(order[q], me) ? 1 : 0;
  return p;
}

template <bool POSITION>
__device__ __forceinline__ Payload payload_of(uint64_t key, int seg_arc, int32_t label_min, int time_bits) {
  const uint64_t tmask = (1ull << time_bits) - 1ull;
  Payload q;
  if (POSITION) {
    const LatView& v = lat_view;
    const int32_t label = (int32_t)(uint32_t)(key >> time_bits) + label_min;
    const int32_t pos = (int32_t)(uint32_t)(key & tmask) + 1;  // 1-based on output
    q.k1 = ((uint64_t)hbias(label) << 32) | hbias(pos);
    const int32_t t0 = v.time[v.src[seg_arc]];
    const int32_t t1 = v.time[v.dst[seg_arc]];
    q.k2 = ((uint64_t)(uint32_t)t0 << 32) | (uint32_t)t1;
  } else {
    const int32_t label = (int32_t)(uint32_t)(key >> (2 * time_bits)) + label_min;
    const int32_t t0 = (int32_t)(uint32_t)((key >> time_bits) & tmask);
    const int32_t t1 = (int32_t)(uint32_t)(key & tmask);
    q.k1 = ((uint64_t)hbias(label) << 32) | hbias(t0);
    q.k2 = (uint64_t)hbias(t1) << 32;
  }
  return q;
}

struct QEntry {  // 16 B, in the lattice's record row until the records go there
  uint32_t klo, khi, packed /* fingerprint << 15 | slot to look at next */, arc;
};

// T threads, P arcs a thread; RB: bits of an arc number.  Dynamic LDS: [0, 4H) table -> order words -> stage;
// [4H, 4H + H/8) the two arc bitmaps; the value counters at cnt_off (inside the first region where the order words
// leave room, else behind the bitmaps).
template <bool POSITION, int T, int P, int WPE>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void index_hash2_kernel(AccumArgs a, RadixPack pk, int log_h, int cnt_off, int log_b, int lds_bytes) {
  constexpr int CAP = T * P;
  constexpr int RB = CAP > 8192 ? 14 : 13;
  static_assert(CAP <= 16384, "arc numbers take 14 bits");
  unsigned char* const smem = smem2;
  const int H = 1 << log_h, B = 1 << log_b;
  uint32_t* table = reinterpret_cast<uint32_t*>(smem);
  uint64_t* order = reinterpret_cast<uint64_t*>(smem);
  uint32_t* dupbit = reinterpret_cast<uint32_t*>(smem + (size_t)H * 4);  // [CAP / 32] arcs of repeated keys
  uint32_t* headbit = dupbit + H / 64;                                   // [CAP / 32] their first arcs
  uint32_t* bcnt = reinterpret_cast<uint32_t*>(smem + cnt_off);          // [B / 2] packed 16-bit counters
  __shared__ int sh[T / 64];
  __shared__ float sh_lo[T / 64], sh_hi[T / 64];
  __shared__ int sh_qn, sh_ndl, sh_nbig, sh_big[CAP / kSerialMax + 1];

  const IndexItem item = a.items[blockIdx.x];
  if (item.u < 0) return;  // (a piece slot the slicing left empty)
  const int u = item.u, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t a0 = a.arc_off[u] + item.begin, s0 = a.state_off[u];
  const int n = item.end - item.begin;
  if (n > CAP) return;  // (never: the host lists a class's lattices, the pieces fit the largest classes)
#define KLAT_STAMP(k)                                                                 \
  do {                                                                                \
    if (a.phase_clock && tid == 0) a.phase_clock[(size_t)u * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
  KLAT_STAMP(0);
  if (n == 0) {  // (an empty lattice: pieces are never empty)
    if (tid == 0) a.num_records[u] = 0;
    return;
  }
  const int64_t pb = padded_state_base(s0, u);
  const int64_t row = a.scratch_off[u] + item.begin;  // (the scratch rows hold a slot per arc: a piece has its arcs' slots)
  double* vals = a.val_scratch + row;
  const double tot = a.total[u];
  const int32_t* const l_olabel = a.olabel + a0;
  const int32_t* const l_src = a.src + a0;
  const int32_t* const l_dst = a.dst + a0;
  const int32_t* const l_tm = (POSITION ? a.state_len : a.state_time) + s0;
  const double* const l_alpha = a.alpha + pb;
  const double* const l_beta = a.beta + pb;
  if (tid == 0) {
    LatView& v = lat_view;
    v.olabel = l_olabel, v.src = l_src, v.dst = l_dst, v.ilabel = a.ilabel + a0;
    v.graph = a.graph + a0, v.acoustic = a.acoustic + a0;
    v.tm = l_tm, v.time = a.state_time + s0;
    v.alpha = l_alpha, v.beta = l_beta;
    v.vals = vals;
    v.include = a.include, v.exclude = a.exclude, v.num_include = a.num_include, v.num_exclude = a.num_exclude;
    v.tot = tot;
    v.sp = a.sp;
    v.label_min = pk.label_min, v.time_bits = pk.time_bits;
    v.dupbit_off = (uint32_t)H * 4u;
  }
  // (24 B a slot: room for n entries.  A piece's row may start 8 bytes past a 16-byte boundary -- its cut is wherever the
  // times allow; the queue's 16-byte accesses are then two-dword aligned, which global memory serves, at the price of a
  // split line now and then on this rare path.  The copy-out below, which every record takes, does align itself.)
  QEntry* queue = reinterpret_cast<QEntry*>(reinterpret_cast<Rec*>(a.rec_scratch) + row);
  // the list of repeated keys' arcs: behind the queue in the record row (16 of a slot's 24 bytes are the queue's),
  // out of the way of the payloads the fold writes while it reads the list
  uint32_t* duplist = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(reinterpret_cast<Rec*>(a.rec_scratch) + row) +
                                                  16 * (size_t)n);
  Payload* payload = reinterpret_cast<Payload*>(a.key_scratch) + row;  // by first arc, repeated keys only (after the list is in LDS)
  Rec* recs = reinterpret_cast<Rec*>(a.rec_scratch) + row;
  const int jmax = (n + T - 1) / T;  // rounds of T arcs this lattice has (workgroup-uniform)
  // the lattice's table: the smallest power of two >= 2 n, 1 024 slots at least
  int lh = 10;
  while ((1 << lh) < 2 * n) ++lh;
  const uint32_t hmask = (1u << lh) - 1u;

  // ---- 1. keys and values (loads batched, no branches around them); table and bitmaps cleared ----
  // A whole lattice whose states' alpha / beta / times (20 bytes a state) fit the table's LDS has them STAGED there first:
  // the arcs' ends then look them up in LDS instead of gathering them from memory -- per group of arcs one trip to memory
  // instead of two, and no 64-address gathers (one 1 024-thread workgroup a CU has nothing to hide them behind: the
  // 14 997-arc class spent 74 of its 212 thousand ticks in this phase).  The table is cleared afterwards.
  const int S_all = (int)(a.state_off[u + 1] - s0);
  // (Only the 1 024-thread classes: with two and more workgroups a CU the neighbours hide the gathers, and the second
  // barrier and the rows' trip through LDS cost the smaller classes 2 - 7 %.)
  const bool staged = T == 1024 && item.slot < 0 && (size_t)S_all * 20 <= (size_t)H * 4;  // (workgroup-uniform; a piece's states are a range this kernel does not know)
  double* const st_alpha = reinterpret_cast<double*>(smem);
  double* const st_beta = st_alpha + S_all;
  int32_t* const st_tm = reinterpret_cast<int32_t*>(st_beta + S_all);
  auto clear_table = [&]() {
    const uint4 free4 = {kFree, kFree, kFree, kFree};
    for (int i = tid * 4; i < (1 << lh); i += T * 4) *reinterpret_cast<uint4*>(table + i) = free4;
  };
  {
    for (int i = tid; i < H / 32; i += T) dupbit[i] = 0;  // (both bitmaps)
    if (tid == 0) sh_qn = 0, sh_ndl = 0, sh_nbig = 0;
    if (staged) {
      for (int s = tid; s < S_all; s += T) st_alpha[s] = l_alpha[s], st_beta[s] = l_beta[s], st_tm[s] = l_tm[s];
      __syncthreads();
    } else {
      clear_table();
    }
  }
  uint64_t key[P];
  double val[P];
  uint32_t valid = 0;
  const bool filtered = (a.num_include | a.num_exclude) != 0;  // workgroup-uniform
  if (filtered) __syncthreads();  // (the filter reads the lattice's view)
  constexpr int HALF = P % 6 == 0 ? 6 : P % 5 == 0 ? 5 : 4;  // arcs a thread asks for at a time (what the registers hold beside the keys and values)
  static_assert(P % HALF == 0, "");
  auto arc_pass = [&](auto from_lds) {
    constexpr bool LDS = decltype(from_lds)::value;
#pragma unroll
    for (int half = 0; half < P / HALF; ++half) {
      if (half * HALF >= jmax) break;
      int32_t lab[HALF], t0[HALF], t1[HALF];
      float gr[HALF], ac[HALF];
      bool word[HALF];
#pragma unroll
      for (int q = 0; q < HALF; ++q) {
        const int i = min((half * HALF + q) * T + tid, n - 1);
        lab[q] = l_olabel[i];
        t0[q] = l_src[i];
        t1[q] = l_dst[i];
        gr[q] = a.graph[a0 + i];
        ac[q] = a.acoustic[a0 + i];
        word[q] = a.sp.penalized ? a.ilabel[a0 + i] != 0 : false;
      }
      double al[HALF], be[HALF];
#pragma unroll
      for (int q = 0; q < HALF; ++q) {
        al[q] = LDS ? st_alpha[t0[q]] : l_alpha[t0[q]];
        be[q] = LDS ? st_beta[t1[q]] : l_beta[t1[q]];
        t0[q] = LDS ? st_tm[t0[q]] : l_tm[t0[q]];
        if (!POSITION) t1[q] = LDS ? st_tm[t1[q]] : l_tm[t1[q]];
      }
#pragma unroll
      for (int q = 0; q < HALF; ++q) {
        const int j = half * HALF + q;
        const int i = j * T + tid;
        // fw[src] + bw[dst] - cost (kwsbin2/lattice-word-index-segment.cc:160-163), the posterior pass's expression
        val[j] = accum_normalise(al[q] + be[q] - arc_cost(gr[q], ac[q], word[q], a.sp), tot);
        bool ok = i < n && lab[q] != 0;
        if (filtered && ok) ok = filter_keeps(lab[q]);
        const uint64_t l = (uint64_t)(uint32_t)(lab[q] - pk.label_min);
        key[j] = POSITION ? ((l << pk.time_bits) | (uint64_t)(uint32_t)t0[q])
                          : ((((l << pk.time_bits) | (uint64_t)(uint32_t)t0[q]) << pk.time_bits) | (uint64_t)(uint32_t)t1[q]);
        if (ok) valid |= 1u << j;
      }
    }
  };
  if (T == 1024 && staged) {
    if constexpr (T == 1024) arc_pass(std::true_type());
    __syncthreads();  // (every thread is done with the staged rows: the table takes their place)
    clear_table();
  } else {
    arc_pass(std::false_type());
  }
  __syncthreads();
  KLAT_STAMP(1);

  // ---- 2. hash grouping: kRounds tries with every compare-and-swap of a round in flight, the rest queued ----
  {
    uint32_t packed[P];  // fingerprint << 15 | slot
    uint32_t pend = valid;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const uint32_t h = key_hash32(key[j]);
      packed[j] = (h << 15) | (h >> (32 - lh));  // (fingerprint: the hash's low 17 bits; slot: its high bits)
    }
    uint32_t stuck = 0;  // a matching fingerprint: the queue looks at the slot (no more tries)
#pragma unroll
    for (int round = 0; round < kRounds; ++round) {
      constexpr int NB = P <= 12 ? P : P % 8 == 0 ? 8 : 5;  // compare-and-swaps in flight together
      static_assert(P % NB == 0, "");
#pragma unroll
      for (int j0 = 0; j0 < P; j0 += NB) {
        uint32_t seen[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int j = j0 + q;
          seen[q] = kFree;
          if ((pend & ~stuck) & (1u << j)) seen[q] = atomicCAS(&table[packed[j] & 0x7FFFu], kFree, (packed[j] & ~0x7FFFu) | (uint32_t)(j * T + tid));
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const int j = j0 + q;
          if (!((pend & ~stuck) & (1u << j))) continue;
          if (seen[q] == kFree) pend &= ~(1u << j);                       // inserted: this arc stands for its key
          else if (((seen[q] ^ packed[j]) >> 15) == 0) stuck |= 1u << j;  // its key, as likely as not
          else if (round + 1 < kRounds) packed[j] = (packed[j] & ~0x7FFFu) | ((packed[j] + 1) & hmask);  // another key's slot: the next one
        }
      }
    }
    if (pend) {
      const int mine_n = __popc(pend);
      int at = atomicAdd(&sh_qn, mine_n);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        if (!(pend & (1u << j))) continue;
        const uint4 e = {(uint32_t)key[j], (uint32_t)(key[j] >> 32), packed[j], (uint32_t)(j * T + tid)};
        *reinterpret_cast<uint4*>(&queue[at++]) = e;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  KLAT_STAMP(2);
  // the queue: linear probing from the slot an entry names; a fingerprint match is checked against the inserter's
  // key, and an arc that finds its key present puts itself and the inserter on the list of repeated keys
  const int qn = sh_qn;
  for (int e = tid; e < qn; e += T) {
    const uint4 raw = *reinterpret_cast<const uint4*>(&queue[e]);
    const uint64_t k = ((uint64_t)raw.y << 32) | raw.x;
    const uint32_t fp = raw.z >> 15, mine = (raw.z & ~0x7FFFu) | raw.w;
    uint32_t h = raw.z & 0x7FFFu;
    for (int probe = 0; probe <= (int)hmask; ++probe) {
      uint32_t seen = table[h];
      if (seen == kFree) {
        seen = atomicCAS(&table[h], kFree, mine);
        if (seen == kFree) break;  // inserted
      }
      if ((seen >> 15) == fp) {
        const int other = (int)(seen & 0x7FFFu);
        if (slow_key<POSITION>(other) == k) {
          const int me = (int)raw.w;
          atomicOr(&dupbit[me >> 5], 1u << (me & 31));
          const uint32_t before = atomicOr(&dupbit[other >> 5], 1u << (other & 31));
          const bool first = !(before & (1u << (other & 31)));
          const int at = atomicAdd(&sh_ndl, first ? 2 : 1);
          duplist[at] = ((uint32_t)other << 14) | (uint32_t)me;
          if (first) duplist[at + 1] = ((uint32_t)other << 14) | (uint32_t)other;
          break;
        }
      }
      h = (h + 1) & hmask;
    }
  }
  if (qn > 0) {  // workgroup-uniform
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  KLAT_STAMP(3);

  // ---- 3. repeated keys: (inserter | arc) words sorted, every run folded in arc order ----
  const int ndl = sh_ndl;  // workgroup-uniform
  uint32_t recmask = valid;  // the arcs that stand for a record: unique keys, plus first arcs of repeated keys
  uint32_t dupmask = 0;
  if (ndl > 0) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int i = j * T + tid;
      if ((valid & (1u << j)) && (dupbit[i >> 5] & (1u << (i & 31)))) {
        dupmask |= 1u << j;
        vals[i] = val[j];
      }
    }
    recmask &= ~dupmask;
    // The list holds (inserter | arc) words, the inserter's own among them: a counting sort by inserter (the table
    // is dead: three 16-bit arrays over the arcs take its place) puts every key's arcs side by side, the thread
    // that meets an inserter's own word orders its few arcs and folds them.  (Sorting the whole list with the
    // workgroup's bitonic network -- the first form -- took 63 of a lattice's 175 thousand cycles where a fifth of
    // the arcs repeat a key; it stays for a key with more arcs than one thread should order.)
    uint32_t* gcnt = table;                                            // [CAP / 2] arcs of an inserter's key, two per word
    uint32_t* goff = table + CAP / 2;                                  // [CAP / 2] where the next one goes -> the group's end
    uint16_t* grouped = reinterpret_cast<uint16_t*>(table + CAP);      // [CAP] the arcs, key by key
    __shared__ int sh_gmax;
    for (int i = tid; i < CAP / 2; i += T) gcnt[i] = 0;
    if (tid == 0) sh_gmax = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int p = tid; p < ndl; p += T) {
      const uint32_t y = duplist[p] >> 14;
      atomicAdd(&gcnt[y >> 1], 1u << ((y & 1) * 16));
    }
    __syncthreads();
    {  // exclusive scan of the CAP 16-bit counts, CAP / T consecutive ones a thread
      constexpr int PERW = (CAP / 2 + T - 1) / T;  // words per thread
      uint32_t w[PERW];
      int sum = 0, most = 0;
#pragma unroll
      for (int q = 0; q < PERW; ++q) {
        w[q] = tid * PERW + q < CAP / 2 ? gcnt[tid * PERW + q] : 0u;
        sum += (int)(w[q] & 0xFFFFu) + (int)(w[q] >> 16);
        most = max(most, max((int)(w[q] & 0xFFFFu), (int)(w[q] >> 16)));
      }
      if (most > 1) atomicMax(&sh_gmax, most);
      int total;
      int run = block_excl_sum(T, sum, &total, sh);
#pragma unroll
      for (int q = 0; q < PERW; ++q) {
        const int lo = run, hi = lo + (int)(w[q] & 0xFFFFu);
        run = hi + (int)(w[q] >> 16);
        if (tid * PERW + q < CAP / 2) goff[tid * PERW + q] = (uint32_t)lo | ((uint32_t)hi << 16);
      }
    }
    __syncthreads();
    const bool serial = sh_gmax <= 64;  // workgroup-uniform
    if (serial) {
      for (int p = tid; p < ndl; p += T) {
        const uint32_t w = duplist[p], y = w >> 14;
        const uint32_t old = atomicAdd(&goff[y >> 1], 1u << ((y & 1) * 16));
        grouped[(old >> ((y & 1) * 16)) & 0xFFFFu] = (uint16_t)(w & 0x3FFFu);
      }
      __syncthreads();
      for (int p = tid; p < ndl; p += T) {
        const uint32_t w = duplist[p], y = w >> 14;
        if (y != (w & 0x3FFFu)) continue;  // (the inserter's own word: one per key)
        const int g = (int)((gcnt[y >> 1] >> ((y & 1) * 16)) & 0xFFFFu);
        const int end = (int)((goff[y >> 1] >> ((y & 1) * 16)) & 0xFFFFu);
        uint16_t* grp = grouped + (end - g);
        for (int i = 1; i < g; ++i) {  // insertion sort by arc number (the key's arcs: a handful)
          const uint16_t x = grp[i];
          int k = i;
          while (k > 0 && grp[k - 1] > x) grp[k] = grp[k - 1], --k;
          grp[k] = x;
        }
        int arc = (int)grp[0];
        const int first_arc = arc;
        double acc = vals[arc], best = acc;
        int best_arc = arc;
        for (int q = 1; q < g; ++q) {
          arc = (int)grp[q];
          const double x = vals[arc];
          acc = hlog_add(acc, x);
          if (x > best) {  // strict: the first maximum in arc order wins
            best = x;
            best_arc = arc;
          }
        }
        vals[first_arc] = acc;  // (keys are disjoint: nobody else reads this arc's own value any more)
        if (POSITION) payload[first_arc] = payload_of<POSITION>(slow_key<POSITION>(first_arc), best_arc, pk.label_min, pk.time_bits);
        atomicOr(&headbit[first_arc >> 5], 1u << (first_arc & 31));
      }
    } else {
      uint32_t* dups = table;
      for (int p = tid; p < ndl; p += T) dups[p] = duplist[p];
      __syncthreads();
      block_sort_any(T, dups, ndl, [](uint32_t x, uint32_t y) { return x < y; });
      for (int p = tid; p < ndl; p += T) {
        const uint32_t w = dups[p];
        if (!(p == 0 || (dups[p - 1] >> 14) != (w >> 14))) continue;
        int arc = (int)(w & 0x3FFFu);
        const int first_arc = arc;
        double acc = vals[arc], best = acc;
        int best_arc = arc;
        for (int q = p + 1; q < ndl && (dups[q] >> 14) == (w >> 14); ++q) {
          arc = (int)(dups[q] & 0x3FFFu);
          const double x = vals[arc];
          acc = hlog_add(acc, x);
          if (x > best) {
            best = x;
            best_arc = arc;
          }
        }
        vals[first_arc] = acc;
        if (POSITION) payload[first_arc] = payload_of<POSITION>(slow_key<POSITION>(first_arc), best_arc, pk.label_min, pk.time_bits);
        atomicOr(&headbit[first_arc >> 5], 1u << (first_arc & 31));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int i = j * T + tid;
      if ((dupmask & (1u << j)) && (headbit[i >> 5] & (1u << (i & 31)))) {
        recmask |= 1u << j;
        val[j] = vals[i];  // the folded sum
      }
    }
  }
  KLAT_STAMP(4);

  // ---- 4. order by (logp desc, key): one bucket pass by value, then the exact order inside every bucket ----
  // range of the finite values, as floats rounded outwards (any common scale keeps the bucket function monotone)
  {
    float flo = INFINITY, fhi = -INFINITY;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      if (!(recmask & (1u << j))) continue;
      const float f = (float)val[j];
      if (f > -INFINITY && f < INFINITY) flo = fminf(flo, f), fhi = fmaxf(fhi, f);
    }
    for (int d = 32; d > 0; d >>= 1) flo = fminf(flo, __shfl_xor(flo, d, 64)), fhi = fmaxf(fhi, __shfl_xor(fhi, d, 64));
    if (lane == 0) sh_lo[wave] = flo, sh_hi[wave] = fhi;
    for (int i = tid; i < B / 2; i += T) bcnt[i] = 0;
  }
  int m;
  (void)block_excl_sum(T, __popc(recmask), &m, sh);  // (its barriers also publish the minima / maxima and the cleared counters)
  float flo = INFINITY, fhi = -INFINITY;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) flo = fminf(flo, sh_lo[w]), fhi = fmaxf(fhi, sh_hi[w]);
  // one ulp outwards: (float)val rounds to nearest, the true extremes may lie half an ulp beyond
  const double vmax = (double)fhi + fabs((double)fhi) * 1.2e-7, vspan = vmax - ((double)flo - fabs((double)flo) * 1.2e-7);
  const float scale = (vspan > 0.0 && vspan < INFINITY) ? (float)(B - 2) / __fsqrt_rn((float)vspan) : 0.0f;
  uint32_t where[P];  // bucket << 16 | arrival number inside it
  uint32_t zeromask = 0;  // records of the last bucket (log-zero)
#pragma unroll
  for (int j = 0; j < P; ++j) {
    where[j] = 0;
    if (recmask & (1u << j)) {
      // Log-zero (an arc on no complete path; NaN with it) has the LAST bucket to itself: such records all tie, the
      // reference orders them by their keys alone -- and the keys, unique as they are, are that bucket's order words:
      // plain 64-bit compares in LDS where value ties would recompute both records from memory (1 % of dead arcs made
      // this kernel six times slower, 20 % twenty-four times).
      int b = B - 1;
      if (val[j] > -INFINITY && val[j] < INFINITY) {
        // monotone in -logp: subtract, clamp, double -> float rounding, correctly rounded float sqrt, multiply, truncate
        const float t = __fsqrt_rn((float)fmax(vmax - val[j], 0.0)) * scale;
        b = t >= (float)(B - 2) ? B - 2 : (int)t;
      } else if (val[j] == INFINITY) {
        b = 0;
      } else {
        zeromask |= 1u << j;
      }
      const uint32_t old = atomicAdd(&bcnt[b >> 1], 1u << ((b & 1) * 16));
      where[j] = ((uint32_t)b << 16) | ((old >> ((b & 1) * 16)) & 0xFFFFu);
    }
  }
  __syncthreads();
  {  // in-place inclusive scan of the B 16-bit counters: B / T consecutive ones per thread
    const int per = (B / 2) / T;  // words per thread: 1 .. 8 (2 T <= B <= H / 2, launch_one)
    uint32_t w[8];
    int sum = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      w[q] = q < per ? bcnt[tid * per + q] : 0u;
      sum += (int)(w[q] & 0xFFFFu) + (int)(w[q] >> 16);
    }
    int total;
    int run = block_excl_sum(T, sum, &total, sh);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int lo = run + (int)(w[q] & 0xFFFFu);
      const int hi = lo + (int)(w[q] >> 16);
      run = hi;
      if (q < per) bcnt[tid * per + q] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
  }
  __syncthreads();
  const uint16_t* bend = reinterpret_cast<const uint16_t*>(bcnt);  // bend[b] = end of bucket b
  constexpr uint64_t idmask = (1ull << RB) - 1ull;
  // an order word: the image's upper 64 - RB bits | the arc.  Two words whose upper parts differ are ordered by
  // them; equal upper parts (values within 2^-38 of one another, or equal) go back to the records themselves.
  // From here on a record's registers hold the image of its value (what the order words and the record want).
  uint64_t img[P];
#pragma unroll
  for (int j = 0; j < P; ++j) {
    img[j] = hdesc_key(val[j]);
    if (recmask & (1u << j)) {
      const int b = (int)(where[j] >> 16);
      const int lo = b ? (int)bend[b - 1] : 0;
      order[lo + (int)(where[j] & 0xFFFFu)] = (zeromask & (1u << j)) ? key[j] : ((img[j] & ~idmask) | (uint64_t)(j * T + tid));
      where[j] = ((uint32_t)lo << 16) | (uint32_t)bend[b];  // the bucket's range from here on
    }
  }
  __syncthreads();
  KLAT_STAMP(5);
  // crowded buckets (clustered or tied values) are sorted by the whole workgroup, in place
  for (int b = tid; b < B; b += T)
    if ((int)bend[b] - (b ? (int)bend[b - 1] : 0) > kSerialMax) sh_big[atomicAdd(&sh_nbig, 1)] = b;
  __syncthreads();
  const int nbig = sh_nbig;  // workgroup-uniform
  for (int g = 0; g < nbig; ++g) {
    const int b = sh_big[g];
    const int lo = b ? (int)bend[b - 1] : 0, hi = (int)bend[b];
    if (b == B - 1)  // (workgroup-uniform) the keys of the log-zero records
      block_sort_any(T, order + lo, hi - lo, [](uint64_t x, uint64_t y) -> bool { return x < y; });
    else
      block_sort_any(T, order + lo, hi - lo, [&](uint64_t x, uint64_t y) -> bool {
        return ((x ^ y) & ~idmask) ? x < y : (x != y && tie_less<POSITION>((int)(x & idmask), (int)(y & idmask)));
      });
  }
  // every record finds its final position inside its bucket by counting the smaller entries (independent LDS
  // reads); a crowded bucket, or one with a word whose upper part equals the record's, takes the slow way
  uint32_t finalpos[(P + 1) / 2];  // two 16-bit positions per word
#pragma unroll
  for (int j = 0; j < P; ++j) {
    int p = 0;
    if ((recmask & ~zeromask) & (1u << j)) {
      const int lo = (int)(where[j] >> 16), hi = (int)(where[j] & 0xFFFFu);
      // the words below every word with my upper part, and those up to the last such word: mine is the only one
      // in between unless another record's value agrees with mine in its upper bits
      const uint64_t floor_w = img[j] & ~idmask, ceil_w = img[j] | idmask;
      bool slow = hi - lo > kSerialMax;
      p = lo;
      if (!slow) {
        int upto = 0;
#ifndef KLAT_IDX_PLAIN_LOOPS_OFF
#pragma clang loop vectorize(disable) unroll(disable)
#endif
        for (int q = lo; q < hi; ++q) {
          const uint64_t f = order[q];
          p += f < floor_w ? 1 : 0;
          upto += f <= ceil_w ? 1 : 0;
        }
        slow = upto - (p - lo) != 1;
      }
      if (slow) p = slow_position<POSITION, RB>(order, lo, hi, floor_w | (uint64_t)(j * T + tid), false, img[j], key[j]);
    }
    if (j & 1) finalpos[j >> 1] |= (uint32_t)p << 16;
    else finalpos[j >> 1] = (uint32_t)p;
  }
  if (zeromask) {  // the log-zero records, apart (no lane of most waves has any): their words are their keys
#pragma unroll
    for (int j = 0; j < P; ++j) {
      if (!(zeromask & (1u << j))) continue;
      const int lo = (int)(where[j] >> 16), hi = (int)(where[j] & 0xFFFFu);
      int p = lo;
      if (hi - lo > kSerialMax) {
        p = slow_position<POSITION, RB>(order, lo, hi, key[j], true, 0, 0);
      } else {
        for (int q = lo; q < hi; ++q) p += order[q] < key[j] ? 1 : 0;
      }
      finalpos[j >> 1] = (finalpos[j >> 1] & ~(0xFFFFu << ((j & 1) * 16))) | ((uint32_t)p << ((j & 1) * 16));
    }
  }
  KLAT_STAMP(6);

  // ---- 5. the records at their final positions: through LDS, so that they leave as full lines ----
  __syncthreads();  // every thread is done with `order` (and with the counters behind it)
  {
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // A whole lattice's row starts on a 16-byte boundary; a piece's row starts at an arbitrary cut, so row * 24 bytes may
    // be 8 past one.  The records are staged one 8-byte word further on then (h), which puts the 16-byte pairs of LDS on the
    // 16-byte pairs of the row: a lone head word, whole uint4 stores, a lone tail word.
    const int h = (int)(row & 1);
    const int stage_recs = (lds_bytes - 8) / (int)sizeof(Rec) & ~1;  // (even: every pass starts on the same parity)
    uint64_t* stage = reinterpret_cast<uint64_t*>(smem);  // [h + stage_recs * 3]
    uint64_t* out64 = reinterpret_cast<uint64_t*>(recs);
    for (int base = 0; base < m; base += stage_recs) {  // (m is workgroup-uniform)
#pragma unroll
      for (int j = 0; j < P; ++j) {
        if (!(recmask & (1u << j))) continue;
        const int p = (int)((finalpos[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu) - base;
        if ((unsigned)p >= (unsigned)stage_recs) continue;
        const int i = j * T + tid;
        Payload q;
        if (POSITION && (dupmask & (1u << j))) q = payload[i];  // (t0, t1) of the first arc attaining the maximum
        else q = payload_of<POSITION>(key[j], i, pk.label_min, pk.time_bits);
        stage[h + p * 3 + 0] = img[j];
        stage[h + p * 3 + 1] = q.k1;
        stage[h + p * 3 + 2] = q.k2;
      }
      lds_barrier();
      const int cnt = min(stage_recs, m - base) * 3;  // 8-byte words of this pass; base * 3 is even
      const uint4* stage4 = reinterpret_cast<const uint4*>(stage);
      uint4* out4 = reinterpret_cast<uint4*>(out64 + (size_t)base * 3 - h);
      for (int i = h + tid; i < (cnt + h) / 2; i += T) out4[i] = stage4[i];
      if (h && tid == 0) out64[(size_t)base * 3] = stage[1];
      if (((cnt + h) & 1) && tid == 0) out64[(size_t)base * 3 + cnt - 1] = stage[h + cnt - 1];
      lds_barrier();
    }
  }
  if (tid == 0) {
    if (item.slot < 0) {
      a.num_records[u] = m;
    } else {  // a piece: its own count for the merge, the lattice's sum for the record offsets
      a.item_records[item.slot] = m;
      atomicAdd(&a.num_records[u], m);
    }
  }
  KLAT_STAMP(7);
#undef KLAT_STAMP
}

template <bool POSITION, int T, int P, int WPE = 4>
hipError_t launch_one(const AccumArgs& a, const RadixPack& pk, int num_lattices, int log_h, hipStream_t stream) {
  const int H = 1 << log_h, cap = T * P;
  // the value counters: H / 2 of them inside the first region where the order words leave room, else H / 4 behind the bitmaps
  int cnt_off, log_b;
  if (8 * cap + H <= 4 * H) {
    cnt_off = 8 * cap, log_b = log_h - 1;
  } else {
    cnt_off = 4 * H + H / 8, log_b = log_h - 2;
  }
  while ((1 << log_b) < 2 * T) ++log_b;  // (the scan gives every thread a whole word at least)
  const int lds = std::max(4 * H + H / 8, cnt_off + 2 * (1 << log_b));
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&index_hash2_kernel<POSITION, T, P, WPE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((index_hash2_kernel<POSITION, T, P, WPE>), dim3(num_lattices), dim3(T), lds, stream, a, pk, log_h, cnt_off, log_b, lds);
  return hipGetLastError();
}

template <bool POSITION>
hipError_t launch(const AccumArgs& a, const RadixPack& pk, int cls, int count, hipStream_t stream) {
  switch (cls) {  // kHashClassArcs (klat_accum.h): threads x arcs a thread, table slots
    case 0: return launch_one<POSITION, 128, 12>(a, pk, count, 12, stream);
    case 1: return launch_one<POSITION, 256, 12>(a, pk, count, 13, stream);
    case 2: return launch_one<POSITION, 512, 12>(a, pk, count, 14, stream);
    case 3: return launch_one<POSITION, 512, 16>(a, pk, count, 14, stream);
    case 4: return launch_one<POSITION, 1024, 15>(a, pk, count, 15, stream);
    case 5: return launch_one<POSITION, 1024, 16>(a, pk, count, 15, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace

// Pieces of a long lattice (klat_accum.h): one workgroup per lattice.  Keys repeat only among arcs whose source states
// share their time, so a cut is good where every arc before it has an earlier source time than every arc behind it
// (klat_slice.h); a wave looks for boundary k = the last such place at or before k * shape.stride.
template <bool POSITION>
__global__ __launch_bounds__(1024) void index_slice_kernel(AccumArgs a, PieceShape shape, const int32_t* long_list) {
  __shared__ int bad;
  __shared__ int32_t before[kSliceBlocks + 1], from[kSliceBlocks + 1];
  const int u = long_list[blockIdx.x], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t a0 = a.arc_off[u], s0 = a.state_off[u];
  const int n = (int)(a.arc_off[u + 1] - a0);
  const int32_t* src = a.src + a0;
  const int32_t* tm = (POSITION ? a.state_len : a.state_time) + s0;
  const int first = a.piece_first[u], pieces = (n + shape.stride - 1) / shape.stride;
  IndexItem* items = a.piece_items + first;
  auto time_of = [&](int i) { return tm[src[i]]; };
  if (tid == 0) bad = n > 64 * kSliceBlocks ? 1 : 0;
  __syncthreads();
  if (!bad) slice_prepare(n, time_of, before, from);  // (workgroup-uniform)
  for (int k = 1 + wave; !bad && k < pieces; k += 16) {  // (wave-uniform)
    const int target = k * shape.stride;  // < n
    const int cut = slice_find(n, time_of, before, from, target - shape.window + 1, target, true);
    if (cut < 0) bad = 1;
    if (lane == 0) items[k].begin = cut;
  }
  if (tid == 0) items[0].begin = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const bool ok = bad == 0;
  const int mine = tid + 1 < pieces ? items[tid + 1].begin : n;  // the end of this thread's piece (at most 1 024 pieces: klat_accum_api.cc), read before anything is rewritten
  __syncthreads();
  if (tid < pieces) {
    IndexItem it;
    it.u = ok ? u : -1, it.begin = items[tid].begin, it.end = mine, it.slot = first + tid;
    items[tid] = it;
    a.item_records[first + tid] = 0;
  }
  if (tid == 0) a.piece_count[u] = ok ? pieces : 0, a.num_records[u] = 0;
}

hipError_t launch_index_slice(const AccumArgs& a, bool position, const PieceShape& shape, const int32_t* long_list, int count, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  if (position)
    hipLaunchKernelGGL(index_slice_kernel<true>, dim3(count), dim3(1024), 0, stream, a, shape, long_list);
  else
    hipLaunchKernelGGL(index_slice_kernel<false>, dim3(count), dim3(1024), 0, stream, a, shape, long_list);
  return hipGetLastError();
}

hipError_t launch_index_hash2(const AccumArgs& a, const RadixPack& pk, bool position, int cls, const IndexItem* items, int count,
                              hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  AccumArgs b = a;
  b.items = items;
  return position ? launch<true>(b, pk, cls, count, stream) : launch<false>(b, pk, cls, count, stream);
}

}  // namespace klat
