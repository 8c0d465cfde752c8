// klat_chain.hip -- forward-backward of long lattices cut into segments (see klat_chain.h).
//
// Replaces, for the lattices it takes, the same CPU graph walk as the lockstep sweep:
// ComputeLatticeAlphasAndBetas [upstream Kaldi, SURVEY.md A.1], called at
// latbin/lattice-to-word-frame-post.cc:88-90 and kwsbin2/lattice-word-index-segment.cc:63.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "klat_chain.h"
#include "klat_math.h"

namespace klat {

namespace {

constexpr int kPrepAhead = 4;  // arcs a thread of chain_prep_kernel has in flight
constexpr int kTP = 1024;  // chain_prep_kernel: a lattice of 20 000 states is 20 turns of its workgroup (256 threads: 80, and the batch waited for them)
// Waves per workgroup of the walks.  The waves share nothing, and single-wave workgroups pack a CU's LDS best
// (the dispatcher spreads them evenly over the four SIMDs: KLAT_CHAIN_DEBUG shows where every wave ran).
constexpr int kWalkWaves = 1;
constexpr int kSweepWaves = 1;
constexpr int kSweepBatch = 8;  // records per batch of the sweep (its output rows are LDS: 8 rows, 14.6 KB a wave, eleven waves a CU)

__device__ __forceinline__ int wave_max_int(int v) {
  for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
  return v;
}

// exclusive prefix sums of two ints per thread over the workgroup
__device__ void block_scan2(int v0, int v1, int* ex0, int* ex1, int* tot0, int* tot1, int* sh /* [2 * kTP/64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int i0 = v0, i1 = v1;
  for (int d = 1; d < 64; d <<= 1) {
    const int t0 = __shfl_up(i0, d, 64), t1 = __shfl_up(i1, d, 64);
    if (lane >= d) i0 += t0, i1 += t1;
  }
  __syncthreads();  // sh may still be read from a previous call
  if (lane == 63) sh[wave] = i0, sh[kTP / 64 + wave] = i1;
  __syncthreads();
  int o0 = 0, o1 = 0, s0 = 0, s1 = 0;
  for (int w = 0; w < kTP / 64; ++w) {
    if (w < wave) o0 += sh[w], o1 += sh[kTP / 64 + w];
    s0 += sh[w], s1 += sh[kTP / 64 + w];
  }
  *ex0 = o0 + i0 - v0, *ex1 = o1 + i1 - v1, *tot0 = s0, *tot1 = s1;
}

// exp(-cost) as a record: the double with its low nine fraction bits replaced by `ctl`.  A dead term
// (infinite cost, NaN) reads the constant zero.  A cost the double cannot hold sets *status.
__device__ __forceinline__ uint64_t chain_record(double cost, uint32_t ctl, double c9v, int32_t* status, bool tropical) {
  if (!(cost < 1.0e30)) return (uint64_t)__double_as_longlong(1.0) | (ctl & (kChainLast | kChainFinal)) | kChainSpecial | kChainWhich;
  if (tropical) {  // the (min,+) twin walks the costs themselves
    const uint64_t cb = (uint64_t)__double_as_longlong(cost);
    return ((cb + 0x100ull) & ~(uint64_t)kChainCtlMask) | ctl;
  }
  if (fabs(cost) > kChainMaxCost) {
    atomicOr(status, 1);
    cost = cost > 0 ? kChainMaxCost : -kChainMaxCost;
  }
  double wm;
  int k;
  bool dead;
  arc_weight(cost, wm, k, dead, ~0ull, c9v);
  uint64_t bits = (uint64_t)__double_as_longlong(ldexp(wm, k));
  bits = (bits + 0x100ull) & ~(uint64_t)kChainCtlMask;  // round to 43 fraction bits (a carry into the exponent is still the right double)
  return bits | ctl;
}

// One workgroup per lattice on this path: the records of its two streams and its segments' ranges.
// Forward stream, state by state upwards: [the constant-one term of the start state] the in-arcs (sources
// ascending: the most recently finished source last), LAST on the last of them -- or one zero term if
// there is none; then, for a final state, its final weight (FINAL, source = the state just finished).
// Backward stream, downwards: [the final weight as a constant-one term] the out-arcs (farthest first).
__global__ __launch_bounds__(kTP) void chain_prep_kernel(ChainPrepArgs a) {
  __shared__ int sh[2 * (kTP / 64)];
  const ChainLattice cl = a.lats[blockIdx.x];
  const int u = cl.lattice, S = cl.num_states, tid = threadIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int64_t sb = s0 + 2 * (int64_t)u;
  const int32_t* cnt = a.cnt + sb;
  const int32_t* row = a.row + sb;
  int32_t* pf = a.scratch_f + sb;  // records before state s, forward stream
  int32_t* pb = a.scratch_b + sb;  // records of the states below s, backward stream
  auto is_final = [&](int s) { return !(a.final_graph[s0 + s] == INFINITY && a.final_acoustic[s0 + s] == INFINITY); };
  int carry_f = 0, carry_b = 0;
  for (int base = 0; base < S; base += kTP) {
    const int s = base + tid;
    int nf = 0, nb = 0;
    if (s < S) {
      const int fin = is_final(s) ? 1 : 0;
      nf = max(cnt[s + 1] - cnt[s] + (s == 0 ? 1 : 0), 1) + fin;
      nb = max(row[s + 1] - row[s] + fin, 1);
    }
    int ef, eb, tf, tb;
    block_scan2(nf, nb, &ef, &eb, &tf, &tb, sh);
    if (s < S) pf[s] = carry_f + ef, pb[s] = carry_b + eb;
    carry_f += tf, carry_b += tb;
  }
  if (tid == 0) pf[S] = carry_f, pb[S] = carry_b;
  __threadfence_block();
  __syncthreads();
  const int Tb = carry_b;
  for (int k = tid; k < cl.num_segs; k += kTP) {  // segments of equal length
    const int b0 = (int)((int64_t)k * S / cl.num_segs), b1 = (int)((int64_t)(k + 1) * S / cl.num_segs);
    ChainSegment sg;
    sg.clat = blockIdx.x, sg.k = k, sg.first_state = b0, sg.num_states = b1 - b0;
    sg.rf_begin = cl.rec_f + (uint32_t)pf[b0], sg.rf_count = (uint32_t)(pf[b1] - pf[b0]);
    sg.rb_begin = cl.rec_b + (uint32_t)(Tb - pb[b1]), sg.rb_count = (uint32_t)(pb[b1] - pb[b0]);
    a.segs[cl.seg_begin + k] = sg;
  }
  const double c9v = arc_weight_c9();
  const int32_t* in_arc = a.in_arc + a0;
  const int32_t* out_arc = a.out_arc + a0;
  double mxc = 0.0;  // the largest finite |cost| this thread meets
  auto note = [&](double cost) {
    const double ac = fabs(cost);
    if (ac < 1.0e30) mxc = fmax(mxc, ac);
  };
  // A thread per ARC, three times.  First in the arcs' own order -- five arrays read as they lie --: the arc's record
  // (its weight; the distance between its states, the same in both streams) into a row of the batch's temporaries.  Then the
  // backward stream in the order of out_arc (a state's arcs among themselves: the rows read are the neighbours') and the
  // forward stream in that of in_arc: ONE random 8-byte read per arc, and its destination, where gathering the arc itself
  // there was five (12 GB of sectors for the mix's 38 M arcs).  An arc's record lies at its state's first record plus its
  // place among the state's arcs.  (A thread per state walked its arcs one by one -- a wave as long as its state of most
  // arcs, every record behind two dependent trips to memory.)
  const bool trop = a.tropical != 0;
  const int A = (int)(a.arc_off[u + 1] - a0);
  uint64_t* wrow = a.arc_records + a0;
  // (kPrepAhead arcs a thread at a time, every stage's loads asked for together: one arc after the other a thread's turn
  // was two or three trips to memory in a row, and all sixteen waves of the workgroup are in the same loop)
  for (int jb = tid; jb < A; jb += kTP * kPrepAhead) {
    float gr[kPrepAhead], ac[kPrepAhead];
    int il[kPrepAhead], sv[kPrepAhead], dv[kPrepAhead];
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) {
      const int64_t x = a0 + min(jb + q * kTP, A - 1);
      gr[q] = a.graph[x], ac[q] = a.acoustic[x], il[q] = a.ilabel[x], sv[q] = a.src[x], dv[q] = a.dst[x];
    }
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) {
      const int j = jb + q * kTP;
      if (j >= A) continue;
      const double cost = arc_cost(gr[q], ac[q], il[q] != 0, a.sp);
      note(cost);
      wrow[j] = chain_record(cost, (uint32_t)(dv[q] - sv[q]), c9v, a.status, trop);
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int ib = tid; ib < A; ib += kTP * kPrepAhead) {
    int j[kPrepAhead], sv[kPrepAhead], r0[kPrepAhead], r1[kPrepAhead], base[kPrepAhead];
    uint64_t w[kPrepAhead];
    bool fin[kPrepAhead];
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) j[q] = out_arc[min(ib + q * kTP, A - 1)];
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) sv[q] = a.src[a0 + j[q]], w[q] = wrow[j[q]];
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) r0[q] = row[sv[q]], r1[q] = row[sv[q] + 1], base[q] = pb[sv[q] + 1], fin[q] = is_final(sv[q]);
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) {
      const int i = ib + q * kTP;
      if (i < A) a.recs[cl.rec_b + (Tb - base[q]) + (fin[q] ? 1 : 0) + (i - r0[q])] = w[q] | (i == r1[q] - 1 ? kChainLast : 0);
    }
  }
  for (int ib = tid; ib < A; ib += kTP * kPrepAhead) {
    int j[kPrepAhead], sv[kPrepAhead], c0[kPrepAhead], c1[kPrepAhead], base[kPrepAhead];
    uint64_t w[kPrepAhead];
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) j[q] = in_arc[min(ib + q * kTP, A - 1)];
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) sv[q] = a.dst[a0 + j[q]], w[q] = wrow[j[q]];
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) c0[q] = cnt[sv[q]], c1[q] = cnt[sv[q] + 1], base[q] = pf[sv[q]];
#pragma unroll
    for (int q = 0; q < kPrepAhead; ++q) {
      const int i = ib + q * kTP;
      if (i < A) a.recs[cl.rec_f + base[q] + (i - c0[q]) + (sv[q] == 0 ? 1 : 0)] = w[q] | (i == c1[q] - 1 ? kChainLast : 0);
    }
  }
  // ... and a thread per state for what is not an arc: the start state's constant one, a state without terms, the final weights
  for (int s = tid; s < S; s += kTP) {
    const bool fin = is_final(s);
    const double fcost = fin ? arc_cost(a.final_graph[s0 + s], a.final_acoustic[s0 + s], false, a.sp) : 0.0;
    note(fcost);
    {
      uint64_t* r = a.recs + cl.rec_f + pf[s];
      const int n_in = cnt[s + 1] - cnt[s];
      const int nterm = n_in + (s == 0 ? 1 : 0);
      if (s == 0) r[0] = chain_record(0.0, kChainSpecial | (nterm == 1 ? kChainLast : 0), c9v, a.status, trop);
      if (nterm == 0) r[0] = chain_record(0.0, kChainSpecial | kChainWhich | kChainLast, c9v, a.status, trop);
      if (fin) r[max(nterm, 1)] = chain_record(fcost, 1u | kChainFinal, c9v, a.status, trop);
    }
    {
      uint64_t* r = a.recs + cl.rec_b + (Tb - pb[s + 1]);
      const int nterm = row[s + 1] - row[s] + (fin ? 1 : 0);
      if (fin) {
        uint64_t rec = chain_record(fcost, nterm == 1 ? kChainLast : 0, c9v, a.status, trop);
        if (!(rec & kChainSpecial)) rec |= kChainSpecial;  // (a dead final weight already reads the constant zero)
        r[0] = rec;
      }
      if (nterm == 0) r[0] = chain_record(0.0, kChainSpecial | kChainWhich | kChainLast, c9v, a.status, trop);
    }
  }
  for (int d = 32; d > 0; d >>= 1) mxc = fmax(mxc, __shfl_xor(mxc, d, 64));
  if ((tid & 63) == 0 && a.maxcost) atomicMax(a.maxcost, (unsigned long long)__double_as_longlong(mxc));  // (positive doubles order like their bits)
}

// (acc * 2^ref) += x * 2^xe for any exponents; acc keeps its alignment to 2^ref
__device__ __forceinline__ void chain_add_exact(double& acc, int& ref, double x, int xe) {
  if (x == 0.0) return;
  const double xm = __builtin_amdgcn_frexp_mant(x);
  const int xe2 = xe + __builtin_amdgcn_frexp_exp(x);
  if (acc == 0.0) {
    acc = xm;
    ref = xe2;
    return;
  }
  const double am = __builtin_amdgcn_frexp_mant(acc);
  const int ae = ref + __builtin_amdgcn_frexp_exp(acc);
  const int hi = max(ae, xe2);
  acc = ldexp(am, ae - hi) + ldexp(xm, xe2 - hi);
  ref = hi;
}

static constexpr int kZeroRowExp = -(1 << 27);  // Qe of a row of zeros
// a nonzero value this far below its lane's largest makes the lattice fall back to the streams
static constexpr double kChainTiny = 0x1p-400;

__device__ __forceinline__ int wave_min_int(int v) {
  for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
  return v;
}

// The walk of one lane over the records of its segment.
//   MODE 0  transfer (forward records): lane = (segment, one of the c states before it); out: the lane's
//           row of Q and its entry of Qf
//   MODE 1  sweep: lane = segment, DIR 0 forward / 1 backward; out: alpha / beta of the segment's states
// A lane's values are PLAIN doubles times 2^E, E the lane's own exponent: a term is one LDS read and one
// fma, the last W states live in the lane's column of an LDS ring.  Every `rescale_iters` groups of four
// records the lane brings the largest of its live values -- the states a term can still name and the open
// accumulator -- back to [0.5, 1) and moves E.  klat_batch_create picks the interval from the batch's
// largest cost so that nothing can leave the doubles' range in between; a live value more than 2^400 below
// the largest (it could, later) raises the batch's flag instead, and the lockstep streams, which these
// lattices keep, sweep them again (SweepArgs::run_if).
// What a SIMD can do here is about one instruction per 4.4 cycles whatever the waves wait for, so the
// loop is written for few instructions: 16 records per iteration, loaded one iteration ahead; while every
// lane of the wave still has a whole batch the steps carry no bounds test; the rare terms (constants, final
// weights, the values injected at a cut) live in a second copy of the batch's body that a batch takes only
// when one of its 16 x 64 records asks for it.
// TROP: the (min,+) twin over cost records -- a value is the cheapest cost, a term source + cost, a state the minimum of
// its terms; no scale, nothing to rescale, nothing that does not fit (klat_viterbi: ComputeLatticeBeam / PruneLattice,
// latbin/lattice-prune-dyn-beam.cc:27-90).
template <int MODE, int DIR, bool TROP>
__device__ __forceinline__ void chain_walk(const ChainArgs& a, double* ring, double* outbuf, const LnTabEntry* lntab) {
  const int lane = threadIdx.x & 63;
  const int job = blockIdx.x * (MODE == 0 ? kWalkWaves : kSweepWaves) + (threadIdx.x >> 6);  // one wave = one job, the waves of a workgroup are on their own
  const int W = a.W, Wm = a.W - 1, c = a.c;
  int seg, comp = 0;
  if (MODE == 0) {
    seg = job * (64 >> a.lgc) + (lane >> a.lgc);
    comp = lane & (c - 1);
  } else {
    seg = job * 64 + lane;
  }
  const bool have = seg < a.num_segs;
  int trouble = 0;  // why the lattice does not fit this arithmetic (bits: klat_chain.h)
  unsigned long long dbg_flush = 0, dbg_wait = 0, dbg_t0 = a.dbg ? __builtin_readcyclecounter() : 0;
  int npend = 0;  // (a lane without a segment keeps 0: the wave-wide maximum below reads it)
  // the wave's shortest and longest record run, over the lanes that have a segment (the shuffles read all 64)
  const int n_all = have ? (int)(DIR ? a.segs[seg].rb_count : a.segs[seg].rf_count) : -1;
  const int nmax = wave_max_int(n_all), nmin = wave_min_int(have ? n_all : 0x7FFFFFFF);
  if (have) {
    const ChainSegment sg = a.segs[seg];
    const ChainLattice cl = a.lats[sg.clat];
    const int first = sg.first_state, end = sg.first_state + sg.num_states, span = cl.span;
    const bool last_seg = sg.k == cl.num_segs - 1;
    double* col = ring + lane;
    int32_t* oute = reinterpret_cast<int32_t*>(outbuf + kSweepBatch * 64) + lane;  // exponents beside the output rows
    double* outx = outbuf + lane;
    constexpr double kZero = TROP ? (double)INFINITY : 0.0;  // the semiring's zero and one
    constexpr double kOne = TROP ? 0.0 : 1.0;
    for (int r = 0; r < W; ++r) col[r * 64] = kZero;
    double acc = kZero, lastX = kZero, finA = kZero;
    int E = 0, finR = 0;
    int pos = DIR ? end - 1 : first;
    if (MODE == 0) {
      if (sg.k > 0) col[((first - c + comp) & Wm) * 64] = kOne;  // the lane's own state is one, the others zero
    } else if (DIR == 0) {
      if (sg.k > 0) {  // alpha of the c states before the segment (chain_combine_kernel), already to one exponent
        if (!TROP) E = a.bnd_fe[seg];
        for (int i = 0; i < c; ++i) {
          lastX = a.bnd_fm[(size_t)seg * c + i];
          col[((first - c + i) & Wm) * 64] = lastX;
        }
      }
    } else {
      if (!TROP && !last_seg) E = a.bnd_be[seg];  // (the first values to arrive are the ones injected below)
    }
    double* vals = nullptr;
    if (MODE == 1) vals = (DIR ? a.beta : a.alpha) + padded_state_base(a.state_off[cl.lattice], cl.lattice);
    int out_pos = pos;  // the first of the finished states whose values wait in outbuf
    const uint64_t* rp = a.recs + (DIR ? sg.rb_begin : sg.rf_begin);
    const int n = n_all;

    auto step = [&](auto special, uint64_t rec) {
      constexpr bool SPECIAL = decltype(special)::value;
      const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
      const double w = __hiloint2double((int)hi, (int)(lo & ~kChainCtlMask));
      const int rel = (int)(lo & kChainRelMask);
      const bool last = (lo & kChainLast) != 0;
      const int cell = (DIR ? pos + rel : pos - rel) & Wm;
      double x = col[cell * 64];
      if (MODE == 1) x = select64(lane_mask(rel == 1), lastX, x);  // the state finished by the previous step: from registers, not through LDS
      if (SPECIAL && (lo & kChainSpecial)) {  // a constant source: zero, or one in the lane's scale
        x = kZero;
        if (!(lo & kChainWhich)) {
          if (TROP) {
            x = kOne;
          } else {
            if (E > 900 || E < -900) trouble |= kChainTroubleOne;
            x = ldexp(1.0, max(-2000, min(2000, -E)));
          }
        }
      }
      if (SPECIAL && (lo & kChainFinal)) {  // a final weight: into the lane's total (transfer), nothing (sweep)
        if (MODE == 0) {
          if (TROP)
            finA = fmin(finA, x + w);
          else
            chain_add_exact(finA, finR, x * w, E);
        }
        x = kZero;
      }
      acc = TROP ? fmin(acc, x + w) : fma(x, w, acc);
      if (SPECIAL && MODE == 1 && DIR == 1) {  // what the arcs that leave the segment contribute to its last c states
        const int i = pos - (end - c);
        if (last && !last_seg && i >= 0) {
          const double y = a.bnd_bm[(size_t)seg * c + i];
          if (TROP) {
            acc = fmin(acc, y);
          } else {
            const int sh = a.bnd_be[seg] - E;
            if (y != 0.0 && (sh > 900 || sh < -900)) trouble |= kChainTroubleInject;
            acc += ldexp(y, max(-2000, min(2000, sh)));
          }
        }
      }
      // every step leaves the open state's sum in the state's own ring cell (nobody reads it before its LAST)
      col[(pos & Wm) * 64] = acc;
      const uint64_t L = lane_mask(last);
      if (MODE == 1) {
        lastX = select64(L, acc, lastX);
        outx[npend * 64] = acc;  // (the slot is written over until a LAST moves on)
        oute[npend * 64] = E;
        npend += last ? 1 : 0;
      }
      acc = select64(L, kZero, acc);
      pos += last ? (DIR ? -1 : 1) : 0;
    };
    // the largest of the live values (the open state's sum, the states behind it that an arc can reach) back to [0.5, 1)
    auto rescale = [&]() {
      if (TROP) return;
      double mx = acc;
      for (int i = 1; i <= span; ++i) mx = fmax(mx, col[((DIR ? pos + i : pos - i) & Wm) * 64]);
      const int ex = __builtin_amdgcn_frexp_exp(mx);
      if (!(mx < 1.0e300)) trouble |= kChainTroubleRange;  // (an infinity, a NaN)
      acc = ldexp(acc, -ex);
      if (MODE == 1) lastX = ldexp(lastX, -ex);
      for (int i = 1; i <= span; ++i) {
        double* p = col + ((DIR ? pos + i : pos - i) & Wm) * 64;
        const double v = ldexp(*p, -ex);
        if (v != 0.0 && v < kChainTiny) trouble |= kChainTroubleSpread;
        *p = v;
      }
      E += ex;
    };
    // the natural logs of the states finished since the last flush, into the batch's rows
    auto flush = [&]() {
      if (MODE == 1) {
        // two states per turn (their logarithms are two independent chains of fma), as long as some lane has one
        for (int j = 0; __any(j < npend); j += 2) {
          const double x0 = outx[j * 64], x1 = outx[(j + 1) * 64];
          const int e0 = oute[j * 64], e1 = oute[(j + 1) * 64];
          const double m0 = __builtin_amdgcn_frexp_mant(x0), m1 = __builtin_amdgcn_frexp_mant(x1);
          const LnTabEntry t0 = lntab[ln_table_index(m0)], t1 = lntab[ln_table_index(m1)];
          const double v0 = TROP ? x0 : table_ln(m0, e0 + __builtin_amdgcn_frexp_exp(x0), t0);
          const double v1 = TROP ? x1 : table_ln(m1, e1 + __builtin_amdgcn_frexp_exp(x1), t1);
          if (j < npend) vals[DIR ? out_pos - j : out_pos + j] = v0;
          if (j + 1 < npend) vals[DIR ? out_pos - j - 1 : out_pos + j + 1] = v1;
        }
        out_pos = pos;
        npend = 0;
      }
    };

    // The records come a BATCH of kB ahead, loaded at the top of an iteration and used in the next one (a lane reads
    // up to kB records past its segment: the array ends in slack).  Loads inside a branch, one load per step, or a
    // batch of four made every step (or every fourth) wait a memory latency.
    constexpr int kB = MODE == 0 ? 16 : kSweepBatch;
    int until = a.rescale_iters;
    const std::true_type yes;
    const std::false_type no;
    if constexpr (MODE == 0) {
      // Transfer: the c lanes of a segment walk the SAME records.  Loaded by every lane, a batch was eight
      // instructions of 64 separate requests each -- the texture path takes them lane by lane even when eight
      // name the same address, and the kernel sat at ~500 of its cycles per wave and batch whatever the steps
      // cost.  So the lanes of a segment share the work: each loads one 16-byte piece of the batch (two
      // records), the pieces meet in the wave's LDS stage, and a step reads its record there (one broadcast read).
      const int G = 64 >> a.lgc;
      uint64_t* stg = reinterpret_cast<uint64_t*>(outbuf) + (size_t)(lane >> a.lgc) * kB;  // [2][G][kB], this segment's slice
      uint4 pr[4];
      auto issue = [&](int t) {
        const uint4* p = reinterpret_cast<const uint4*>(rp + min(t, n));
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k * c < kB / 2) pr[k] = p[min(comp + k * c, kB / 2 - 1)];
      };
      auto land = [&](int buf) {
        uint32_t ctl = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = comp + k * c;
          if (k * c < kB / 2 && q < kB / 2) {
            reinterpret_cast<uint4*>(stg + (size_t)buf * G * kB)[q] = pr[k];
            ctl |= pr[k].x | pr[k].z;
          }
        }
        return __any((ctl & (kChainSpecial | kChainFinal)) != 0);
      };
      auto body = [&](auto special, auto guarded, int t0, int buf) {
        const uint64_t* r = stg + (size_t)buf * G * kB;
#pragma unroll
        for (int i = 0; i < kB; ++i) {
          if (!decltype(guarded)::value || t0 + i < n) step(special, r[i]);
          if ((i & 3) == 3 && --until == 0) {
            until = a.rescale_iters;
            rescale();
          }
        }
      };
      issue(0);
      bool rare = land(0);
      int buf = 0, t0 = 0;
      for (; t0 < nmax; t0 += kB, buf ^= 1) {
        issue(t0 + kB);
        if (t0 + kB <= nmin) {  // every lane has the whole batch: no bounds tests; the rare terms only if some record asks
          if (rare)
            body(yes, no, t0, buf);
          else
            body(no, no, t0, buf);
        } else {
          body(yes, yes, t0, buf);
        }
        rare = land(buf ^ 1);
      }
    } else {
      // Sweep: a lane walks its own records: eight 16-byte loads per lane.  The outputs of a batch leave right
      // before the next batch's loads are issued, so the one wait at the bottom (the compiler drains the queue
      // there: loads and stores share its counter) covers both.
      uint4 cur[kB / 2], nxt[kB / 2];
      auto load_batch = [&](int t, uint4 (&q)[kB / 2]) {
        const uint4* p = reinterpret_cast<const uint4*>(rp + min(t, n));
#pragma unroll
        for (int i = 0; i < kB / 2; ++i) q[i] = p[i];
      };
      auto body = [&](auto special, auto guarded, int t0) {
#pragma unroll
        for (int i = 0; i < kB / 2; ++i) {
          if (!decltype(guarded)::value || t0 + 2 * i < n) step(special, ((uint64_t)cur[i].y << 32) | cur[i].x);
          if (!decltype(guarded)::value || t0 + 2 * i + 1 < n) step(special, ((uint64_t)cur[i].w << 32) | cur[i].z);
          if ((i & 1) == 1 && --until == 0) {
            until = a.rescale_iters;
            rescale();
          }
        }
      };
      load_batch(0, cur);
      for (int t0 = 0; t0 < nmax; t0 += kB) {
        const unsigned long long tf0 = a.dbg ? __builtin_readcyclecounter() : 0;
        flush();
        if (a.dbg) dbg_flush += __builtin_readcyclecounter() - tf0;
        load_batch(t0 + kB, nxt);
        if (t0 + kB <= nmin) {
          uint32_t any_ctl = 0;
#pragma unroll
          for (int i = 0; i < kB / 2; ++i) any_ctl |= cur[i].x | cur[i].z;
          bool rare = (any_ctl & (kChainSpecial | kChainFinal)) != 0;
          if (DIR == 1) rare |= !last_seg && pos >= end - c;
          if (__any(rare))
            body(yes, no, t0);
          else
            body(no, no, t0);
        } else {
          body(yes, yes, t0);
        }
        const unsigned long long tw0 = a.dbg ? __builtin_readcyclecounter() : 0;
#pragma unroll
        for (int i = 0; i < kB / 2; ++i) cur[i] = nxt[i];
        if (a.dbg) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          dbg_wait += __builtin_readcyclecounter() - tw0;
        }
      }
      flush();
    }

    if (MODE == 0) {
      // the lane's row of Q: the segment's last c states, scaled once more -- those an arc of this lattice can
      // leave across the cut, that is; the ones behind them (c covers the longest arc of the batch) are never read
      // again and pass as zero; its entry of Qf
      rescale();
      double row_max = 0.0;
      for (int i = 0; i < c; ++i) {
        const double v = i >= c - span ? col[((end - c + i) & Wm) * 64] : kZero;
        row_max = fmax(row_max, v);
        a.Qm[((size_t)seg * c + comp) * c + i] = v;
      }
      if (TROP) {
        a.Qfm[(size_t)seg * c + comp] = finA;
      } else {
        a.Qe[(size_t)seg * c + comp] = row_max != 0.0 ? E : kZeroRowExp;  // (a row of zeros has no scale: it must not set the others')
        a.Qfm[(size_t)seg * c + comp] = __builtin_amdgcn_frexp_mant(finA);
        a.Qfe[(size_t)seg * c + comp] = finR + __builtin_amdgcn_frexp_exp(finA);
      }
    }
  }
  if (trouble) atomicOr(a.flag, trouble);
  if (a.dbg && lane == 0 && MODE == 1) {  // (KLAT_CHAIN_DEBUG: cycles in the flushes, in the waits for the next batch, in all)
    unsigned long long* d = a.dbg + a.dbg_stride + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    d[4] = dbg_flush, d[5] = dbg_wait, d[6] = __builtin_readcyclecounter() - dbg_t0;
  }
}

template <int MODE, bool TROP>
__global__ __launch_bounds__(64 * kWalkWaves) void chain_walk_kernel(ChainArgs a) {
  extern __shared__ double smem[];
  LnTabEntry* lntab = reinterpret_cast<LnTabEntry*>(smem);
  const int wave = threadIdx.x >> 6;
  // doubles: the ring; then the sweep's output rows and their exponents, or the transfer's record stage [2][64 / c][16]
  const int per_wave = a.W * 64 + (MODE == 1 ? kSweepBatch * 64 + kSweepBatch * 32 : 2 * (64 >> a.lgc) * 16);
  double* ring = smem + 2 * kLnTabSize + wave * per_wave;
  double* outbuf = ring + a.W * 64;
  if (threadIdx.x < kLnTabSize) lntab[threadIdx.x] = a.ln_table[threadIdx.x];
  if (MODE == 0 && blockIdx.x == 0 && threadIdx.x == 0 && a.flag_next) *a.flag_next = 0;  // (nobody looks at the next sweep's flag during this one)
  __syncthreads();
  const unsigned long long t_begin = a.dbg ? wall_clock64() : 0;
  if (MODE == 0) {
    chain_walk<0, 0, TROP>(a, ring, outbuf, lntab);
  } else if (blockIdx.y == 0) {
    chain_walk<1, 0, TROP>(a, ring, outbuf, lntab);
  } else {
    chain_walk<1, 1, TROP>(a, ring, outbuf, lntab);
  }
  if (a.dbg && (threadIdx.x & 63) == 0) {  // (KLAT_CHAIN_DEBUG: where and when every wave ran)
    const size_t w = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (MODE == 0 ? kWalkWaves : kSweepWaves) + wave;
    unsigned long long* d = a.dbg + (MODE == 0 ? 0 : a.dbg_stride) + w * (MODE == 0 ? 4 : 8);
    d[0] = t_begin;
    d[1] = wall_clock64();
    d[2] = __builtin_amdgcn_s_getreg(63492);  // HW_ID
    d[3] = __builtin_amdgcn_s_getreg(63508);  // XCC_ID
  }
}

// Over the 16 lanes that hold a vector (one DPP row): rotations within the row, no LDS traffic.
template <int N>
__device__ __forceinline__ int row_ror_int(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, true);
}
template <int N>
__device__ __forceinline__ double row_ror_f64(double v) {
  return __hiloint2double(row_ror_int<N>(__double2hiint(v)), row_ror_int<N>(__double2loint(v)));
}
__device__ __forceinline__ int group_max_int(int v) {
  static_assert(kChainMaxSpan == 16, "one DPP row");
  v = max(v, row_ror_int<8>(v));
  v = max(v, row_ror_int<4>(v));
  v = max(v, row_ror_int<2>(v));
  v = max(v, row_ror_int<1>(v));
  return v;
}
__device__ __forceinline__ double group_sum_f64(double v) {
  v += row_ror_f64<8>(v);
  v += row_ror_f64<4>(v);
  v += row_ror_f64<2>(v);
  v += row_ror_f64<1>(v);
  return v;
}
// lane j's value to every lane (j known at compile time: two v_readlane)
template <int J>
__device__ __forceinline__ double lane_value(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), J), __builtin_amdgcn_readlane(__double2loint(v), J));
}
template <int J, int N, class F>
__device__ __forceinline__ void for_lanes(F f) {
  if constexpr (J < N) {
    f(std::integral_constant<int, J>());
    for_lanes<J + 1, N>(f);
  }
}
static constexpr int kNoExp = -(1 << 28);
static constexpr int kCombineChunk = 8;  // segments whose matrices are in LDS together

// The serial part, in the same scaled arithmetic, on TWO levels: a lattice's segments form groups of
// kChainGroup; chain_group_kernel multiplies every group's matrices (all groups at once: M = Q Q' Q'' ..., and
// Mf, the paths into the group's final states); LEVEL 0 of this kernel walks a lattice's GROUPS with those
// products -- the running vector before every group, forward (alpha of the c states before it) and backward (the
// y of the cut behind it), and tot_forward; LEVEL 1 walks the segments of ONE group from the vector LEVEL 0 left
// for it.  A lattice of K segments is K / 16 + 2 x 16 serial steps instead of K (the single-level kernel was the
// longest of the path: 0.2 ms for the 187 segments of a 20 000-state lattice).
// One wave per (lattice | group, direction); lane i < c holds entry i of the running vector as a plain double, the
// vector shares one exponent.  The matrices come through LDS a chunk of eight at a time, the next chunk's loads
// in flight while this one is used: a step is a few hundred cycles, a trip to memory ten times that.
template <int C, int LEVEL>
__global__ __launch_bounds__(64) void chain_combine_kernel(ChainArgs a) {
  extern __shared__ double smem[];
  constexpr int c = C, cc = C * C;
  const int lane = threadIdx.x;
  const bool bwd = blockIdx.y != 0;
  // the items this wave walks: LEVEL 0 the groups of lattice blockIdx.x, LEVEL 1 the segments of group blockIdx.x
  int first, count, lat_first_seg = -1;
  ChainLattice cl;
  if (LEVEL == 0) {
    cl = a.lats[blockIdx.x];
    first = cl.grp_begin, count = cl.num_grps;
  } else {
    const ChainGroup g = a.grps[blockIdx.x];
    cl = a.lats[g.clat];
    first = g.seg_first, count = g.nseg, lat_first_seg = cl.seg_begin;
  }
  const double* Mm = LEVEL == 0 ? a.Gm : a.Qm;
  const int32_t* Me = LEVEL == 0 ? a.Ge : a.Qe;
  const double* Mfm = LEVEL == 0 ? a.Gfm : a.Qfm;
  const int32_t* Mfe = LEVEL == 0 ? a.Gfe : a.Qfe;
  double* out_m = LEVEL == 0 ? (bwd ? a.gb_bm : a.gb_fm) : (bwd ? a.bnd_bm : a.bnd_fm);
  int32_t* out_e = LEVEL == 0 ? (bwd ? a.gb_be : a.gb_fe) : (bwd ? a.bnd_be : a.bnd_fe);
  const int pd = cc + c, pi = 2 * c;  // per item: the matrix and its final-state vector; their exponents
  double* sd = smem;                  // [2][kCombineChunk][pd]
  int32_t* si = reinterpret_cast<int32_t*>(smem + 2 * kCombineChunk * pd);  // [2][kCombineChunk][pi]
  const bool mine = lane < c;
  const int me = mine ? lane : 0;
  auto item_of = [&](int s) { return (size_t)first + (size_t)(bwd ? count - 1 - s : s); };
  int trouble = 0;
  // stage: registers <- memory (issue), later LDS <- registers
  constexpr int kRD = (C * C + C + 63) / 64;  // doubles per lane and item
  double rd[kCombineChunk][kRD];
  int32_t ri[kCombineChunk];
  auto issue = [&](int chunk) {
#pragma unroll
    for (int s = 0; s < kCombineChunk; ++s) {
      const size_t it = item_of(min(chunk * kCombineChunk + s, count - 1));
#pragma unroll
      for (int r = 0; r < kRD; ++r) {
        const int idx = lane + 64 * r;
        rd[s][r] = idx < cc ? Mm[it * cc + idx] : (idx < pd ? Mfm[it * c + idx - cc] : 0.0);
      }
      ri[s] = lane < c ? Me[it * c + lane] : (lane < pi ? Mfe[it * c + lane - c] : 0);
    }
  };
  auto land = [&](int buf) {
#pragma unroll
    for (int s = 0; s < kCombineChunk; ++s) {
#pragma unroll
      for (int r = 0; r < kRD; ++r) {
        const int idx = lane + 64 * r;
        if (idx < pd) sd[(buf * kCombineChunk + s) * pd + idx] = rd[s][r];
      }
      if (lane < pi) si[(buf * kCombineChunk + s) * pi + lane] = ri[s];
    }
  };
  if (count <= 0) return;
  issue(0);
  // forward: alpha of the c states before the item (the lattice's start: entry 0 alone carries weight -- every row
  // of its first matrix is the row of the start state, the constant-one term feeds all lanes alike); backward: the y
  // of the cut behind the item (the lattice's end: nothing)
  double vec = (!bwd && lane == 0) ? 1.0 : 0.0;
  int Ev = 0;
  if (LEVEL == 1) {
    const double* gm = bwd ? a.gb_bm : a.gb_fm;
    const int32_t* ge = bwd ? a.gb_be : a.gb_fe;
    vec = mine ? gm[(size_t)blockIdx.x * c + lane] : 0.0;
    Ev = ge[blockIdx.x];
  }
  land(0);
  __syncthreads();
  double tot = 0.0;
  int totE = 0;
  const int nchunks = (count + kCombineChunk - 1) / kCombineChunk;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) issue(ch + 1);
    for (int s = 0; s < kCombineChunk && ch * kCombineChunk + s < count; ++s) {
      const int step = ch * kCombineChunk + s;
      const size_t it = item_of(step);
      const double* qd = sd + (buf * kCombineChunk + s) * pd;
      const int32_t* qi = si + (buf * kCombineChunk + s) * pi;
      const double qfm = qd[cc + me];
      const int qe = qi[me], qfe = qi[c + me];
      // the vector at this item's cut
      if (LEVEL == 0 || bwd || (int)it != lat_first_seg) {
        if (mine) out_m[it * c + lane] = vec;
        if (lane == 0) out_e[it] = Ev;
      }
      if (!bwd) {
        if (LEVEL == 0) {  // tot += sum_j a[j] Mf[j]: every group's paths into its final states
          const double t = mine ? vec * qfm : 0.0;
          const int te = t != 0.0 ? Ev + qfe : kNoExp;
          const int hi = group_max_int(te);
          const double sum = group_sum_f64(t != 0.0 ? ldexp(t, te - hi) : 0.0);
          if (hi != kNoExp) chain_add_exact(tot, totE, sum, hi);
        }
        if (step == count - 1) break;
        // r[j] = a[j] 2^(Me[j]) to one exponent Er; new[i] = sum_j r[j] M[j][i]
        const int re = (mine && vec != 0.0 && qe != kZeroRowExp) ? qe : kNoExp;
        const int Er = group_max_int(re);
        double r = 0.0;
        if (re != kNoExp) {
          if (re - Er < -900) trouble |= kChainTroubleRows;
          r = ldexp(vec, re - Er);
        }
        double nw = 0.0;
        for_lanes<0, C>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const double rj = lane_value<j>(r);
          if (mine) nw = fma(rj, qd[j * c + lane], nw);
        });
        const int ne = (mine && nw != 0.0) ? __builtin_amdgcn_frexp_exp(nw) : kNoExp;
        const int ex = group_max_int(ne);
        if (ex == kNoExp) {  // nothing reaches the next cut
          vec = 0.0;
        } else {
          vec = mine ? ldexp(nw, -ex) : 0.0;
          if (vec != 0.0 && vec < kChainTiny) trouble |= kChainTroubleVector;
          Ev += Er + ex;
        }
      } else {
        if (step == count - 1) break;
        // y[j] <- sum_i M[j][i] y[i] + Mf[j]: through the item to the y of the cut behind it, or into a final state
        double val = 0.0;
        for_lanes<0, C>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const double yi = lane_value<i>(vec);
          if (mine) val = fma(qd[lane * c + i], yi, val);
        });
        int ve = (qe == kZeroRowExp ? 0 : qe) + Ev;
        if (mine) chain_add_exact(val, ve, qfm, qfe);
        const int ne = (mine && val != 0.0) ? ve + __builtin_amdgcn_frexp_exp(val) : kNoExp;
        const int ex = group_max_int(ne);
        if (ex == kNoExp) {
          vec = 0.0;
        } else {
          vec = (mine && val != 0.0) ? ldexp(val, max(-2000, ve - ex)) : 0.0;
          if (ne != kNoExp && ne - ex < -400) trouble |= kChainTroubleVector;
          Ev = ex;
        }
      }
    }
    if (ch + 1 < nchunks) {
      land(buf ^ 1);
      __syncthreads();
    }
  }
  if (LEVEL == 0 && !bwd && lane == 0) {  // tot_forward, natural log, into the row's extra slot
    const double m = __builtin_amdgcn_frexp_mant(tot);
    const double e = (double)(totE + __builtin_amdgcn_frexp_exp(tot));
    a.alpha[padded_state_base(a.state_off[cl.lattice], cl.lattice) + cl.num_states] = tot == 0.0 ? -INFINITY : log(m) + e * kLn2;
  }
  if (trouble) atomicOr(a.flag, trouble);
}

// The product of a group's transfer matrices and its final-state vector (one wave per group; all in LDS):
//   P = Q_1;  Pf = Qf_1;  for t = 2 .. n:  Pf += P Qf_t;  P = P Q_t
// Rows are plain doubles to one exponent each, as the walks leave them; a row of zeros has the exponent kZeroRowExp.
template <int C>
__global__ __launch_bounds__(64) void chain_group_kernel(ChainArgs a) {
  constexpr int c = C, cc = C * C;
  constexpr int kRE = (cc + 63) / 64;  // matrix entries per lane
  __shared__ double P[cc], Q[cc], Qf[c];
  __shared__ int32_t Pe[c], Qe[c], Qfe[c], rowx[c];
  const ChainGroup g = a.grps[blockIdx.x];
  const int lane = threadIdx.x;
  const bool mine = lane < c;
  int trouble = 0;
  double pf = 0.0;  // lane j < c: Pf[j] aligned to 2^pfe
  int pfe = 0;
  for (int t = 0; t < g.nseg; ++t) {
    const size_t seg = (size_t)g.seg_first + t;
    __syncthreads();  // (the previous turn's reads of Q)
    // Q_t into LDS, its rows brought to one exponent (that of its largest nonzero row)
    const int qe_own = mine ? a.Qe[seg * c + lane] : kZeroRowExp;
    const int re = qe_own != kZeroRowExp ? qe_own : kNoExp;
    const int Er = group_max_int(re);  // (lanes 0 .. 15 hold it; the others get it below)
    if (mine) {
      Qe[lane] = qe_own;
      Qf[lane] = a.Qfm[seg * c + lane];
      Qfe[lane] = a.Qfe[seg * c + lane];
    }
    if (lane == 0) rowx[0] = Er;
    __syncthreads();
    const int Eall = rowx[0];
#pragma unroll
    for (int r = 0; r < kRE; ++r) {
      const int e = lane + 64 * r;
      if (e < cc) {
        const int m = e / c;
        const int qe = Qe[m];
        double v = a.Qm[seg * cc + e];
        if (t > 0) {
          if (qe == kZeroRowExp) {
            v = 0.0;
          } else {
            if (v != 0.0 && qe - Eall < -900) trouble |= kChainTroubleRows;
            v = ldexp(v, max(-2000, qe - Eall));
          }
        }
        Q[e] = v;
      }
    }
    __syncthreads();
    if (t == 0) {  // P = Q_1, Pf = Qf_1
#pragma unroll
      for (int r = 0; r < kRE; ++r) {
        const int e = lane + 64 * r;
        if (e < cc) P[e] = Q[e];
      }
      if (mine) {
        Pe[lane] = Qe[lane] == kZeroRowExp ? 0 : Qe[lane];
        pf = Qf[lane];
        pfe = Qfe[lane];
      }
      continue;
    }
    // Pf[j] += sum_m P[j][m] Qf_t[m], the Qf to one exponent (their largest)
    {
      const int fe = (mine && Qf[lane] != 0.0) ? Qfe[lane] : kNoExp;
      const int Fm = group_max_int(fe);
      if (lane == 0) rowx[1 % c] = Fm;  // (c >= 2)
      __syncthreads();
      const int Fall = rowx[1 % c];
      if (mine && Fall != kNoExp) {
        double sum = 0.0;
        for (int m = 0; m < c; ++m) {
          const double f = Qf[m];
          if (f == 0.0) continue;
          if (Qfe[m] - Fall < -900) trouble |= kChainTroubleRows;
          sum = fma(P[lane * c + m], ldexp(f, max(-2000, Qfe[m] - Fall)), sum);
        }
        chain_add_exact(pf, pfe, sum, Pe[lane] + Fall);
      }
      __syncthreads();
    }
    // P <- P Q_t (entry by entry), then every row back to [0.5, 1)
    double nw[kRE];
#pragma unroll
    for (int r = 0; r < kRE; ++r) {
      const int e = lane + 64 * r;
      nw[r] = 0.0;
      if (e < cc) {
        const int j = e / c, i = e % c;
        for (int m = 0; m < c; ++m) nw[r] = fma(P[j * c + m], Q[m * c + i], nw[r]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRE; ++r) {
      const int e = lane + 64 * r;
      if (e < cc) P[e] = nw[r];
    }
    __syncthreads();
    if (mine) {
      double mx = 0.0;
      for (int i = 0; i < c; ++i) mx = fmax(mx, P[lane * c + i]);
      rowx[lane] = mx != 0.0 ? __builtin_amdgcn_frexp_exp(mx) : kNoExp;
      if (!(mx < 1.0e300)) trouble |= kChainTroubleRange;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRE; ++r) {
      const int e = lane + 64 * r;
      if (e < cc) {
        const int ex = rowx[e / c];
        if (ex != kNoExp) {
          const double v = ldexp(P[e], -ex);
          if (v != 0.0 && v < kChainTiny) trouble |= kChainTroubleSpread;
          P[e] = v;
        }
      }
    }
    if (mine && rowx[lane] != kNoExp && Eall != kNoExp) Pe[lane] += Eall + rowx[lane];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRE; ++r) {
    const int e = lane + 64 * r;
    if (e < cc) a.Gm[(size_t)blockIdx.x * cc + e] = P[e];
  }
  if (mine) {
    double mx = 0.0;
    for (int i = 0; i < c; ++i) mx = fmax(mx, P[lane * c + i]);
    a.Ge[(size_t)blockIdx.x * c + lane] = mx != 0.0 ? Pe[lane] : kZeroRowExp;
    a.Gfm[(size_t)blockIdx.x * c + lane] = __builtin_amdgcn_frexp_mant(pf);
    a.Gfe[(size_t)blockIdx.x * c + lane] = pfe + __builtin_amdgcn_frexp_exp(pf);
  }
  if (trouble) atomicOr(a.flag, trouble);
}

// ---- the (min,+) twin of the combine: the same two levels, no scales ----
// P = Q_1;  Pf = Qf_1;  for t = 2 .. n:  Pf = min(Pf, P (x) Qf_t);  P = P (x) Q_t    ((x): min over the inner index of the sums)
template <int C>
__global__ __launch_bounds__(64) void chain_group_trop_kernel(ChainArgs a) {
  constexpr int c = C, cc = C * C;
  constexpr int kRE = (cc + 63) / 64;
  __shared__ double P[cc], Q[cc], Qf[c];
  const ChainGroup g = a.grps[blockIdx.x];
  const int lane = threadIdx.x;
  const bool mine = lane < c;
  double pf = INFINITY;
  for (int t = 0; t < g.nseg; ++t) {
    const size_t seg = (size_t)g.seg_first + t;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRE; ++r) {
      const int e = lane + 64 * r;
      if (e < cc) Q[e] = a.Qm[seg * cc + e];
    }
    if (mine) Qf[lane] = a.Qfm[seg * c + lane];
    __syncthreads();
    if (t == 0) {
#pragma unroll
      for (int r = 0; r < kRE; ++r) {
        const int e = lane + 64 * r;
        if (e < cc) P[e] = Q[e];
      }
      if (mine) pf = Qf[lane];
      continue;
    }
    if (mine)
      for (int m = 0; m < c; ++m) pf = fmin(pf, P[lane * c + m] + Qf[m]);
    double nw[kRE];
#pragma unroll
    for (int r = 0; r < kRE; ++r) {
      const int e = lane + 64 * r;
      nw[r] = INFINITY;
      if (e < cc) {
        const int j = e / c, i = e % c;
        for (int m = 0; m < c; ++m) nw[r] = fmin(nw[r], P[j * c + m] + Q[m * c + i]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRE; ++r) {
      const int e = lane + 64 * r;
      if (e < cc) P[e] = nw[r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRE; ++r) {
    const int e = lane + 64 * r;
    if (e < cc) a.Gm[(size_t)blockIdx.x * cc + e] = P[e];
  }
  if (mine) a.Gfm[(size_t)blockIdx.x * c + lane] = pf;
}

// LEVEL 0: the groups of lattice blockIdx.x; LEVEL 1: the segments of group blockIdx.x (see chain_combine_kernel).
// Lane i < c holds entry i of the running vector; the best complete path goes into the forward row's extra slot.
template <int C, int LEVEL>
__global__ __launch_bounds__(64) void chain_combine_trop_kernel(ChainArgs a) {
  constexpr int c = C;
  const int lane = threadIdx.x;
  const bool bwd = blockIdx.y != 0;
  int first, count, lat_first_seg = -1;
  ChainLattice cl;
  if (LEVEL == 0) {
    cl = a.lats[blockIdx.x];
    first = cl.grp_begin, count = cl.num_grps;
  } else {
    const ChainGroup g = a.grps[blockIdx.x];
    cl = a.lats[g.clat];
    first = g.seg_first, count = g.nseg, lat_first_seg = cl.seg_begin;
  }
  const double* Mm = LEVEL == 0 ? a.Gm : a.Qm;
  const double* Mfm = LEVEL == 0 ? a.Gfm : a.Qfm;
  double* out_m = LEVEL == 0 ? (bwd ? a.gb_bm : a.gb_fm) : (bwd ? a.bnd_bm : a.bnd_fm);
  const bool mine = lane < c;
  double vec = (!bwd && lane == 0) ? 0.0 : INFINITY;
  if (LEVEL == 1) vec = mine ? (bwd ? a.gb_bm : a.gb_fm)[(size_t)blockIdx.x * c + lane] : INFINITY;
  double tot = INFINITY;
  for (int step = 0; step < count; ++step) {
    const size_t it = (size_t)first + (size_t)(bwd ? count - 1 - step : step);
    if ((LEVEL == 0 || bwd || (int)it != lat_first_seg) && mine) out_m[it * c + lane] = vec;
    if (!bwd && LEVEL == 0) {
      double t = mine ? vec + Mfm[it * c + lane] : INFINITY;
      t = fmin(t, row_ror_f64<8>(t)), t = fmin(t, row_ror_f64<4>(t)), t = fmin(t, row_ror_f64<2>(t)), t = fmin(t, row_ror_f64<1>(t));
      tot = fmin(tot, t);
    }
    if (step == count - 1) break;
    double nw = INFINITY;
    if (!bwd) {
      for_lanes<0, C>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const double vj = lane_value<j>(vec);
        if (mine) nw = fmin(nw, vj + Mm[(it * c + j) * c + lane]);
      });
    } else {
      for_lanes<0, C>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const double yi = lane_value<i>(vec);
        if (mine) nw = fmin(nw, Mm[(it * c + lane) * c + i] + yi);
      });
      if (mine) nw = fmin(nw, Mfm[it * c + lane]);
    }
    vec = mine ? nw : INFINITY;
  }
  if (LEVEL == 0 && !bwd && lane == 0) a.alpha[padded_state_base(a.state_off[cl.lattice], cl.lattice) + cl.num_states] = tot;
}

}  // namespace

hipError_t launch_chain_prep(const ChainPrepArgs& a, int num_chain_lattices, hipStream_t stream) {
  if (num_chain_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(chain_prep_kernel, dim3(num_chain_lattices), dim3(kTP), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_chain_sweep(const ChainArgs& a, hipStream_t stream, bool tropical) {
  if (a.num_segs <= 0) return hipSuccess;
  const size_t ring_bytes = (size_t)a.W * 64 * sizeof(double);
  const size_t lds_a = kLnTabSize * sizeof(LnTabEntry) + kWalkWaves * (ring_bytes + (size_t)2 * (64 >> a.lgc) * 16 * sizeof(uint64_t));
  const size_t lds_c = kLnTabSize * sizeof(LnTabEntry) + kSweepWaves * (ring_bytes + (size_t)kSweepBatch * 64 * (sizeof(double) + sizeof(int32_t)));
  const size_t lds_b = (size_t)2 * kCombineChunk * ((size_t)(a.c * a.c + a.c) * sizeof(double) + (size_t)2 * a.c * sizeof(int32_t));
  const int jobs_a = (a.num_segs + (64 >> a.lgc) - 1) / (64 >> a.lgc), jobs_c = (a.num_segs + 63) / 64;
  const dim3 grid_a((jobs_a + kWalkWaves - 1) / kWalkWaves), grid_c((jobs_c + kSweepWaves - 1) / kSweepWaves, 2);
  if (tropical)
    hipLaunchKernelGGL((chain_walk_kernel<0, true>), grid_a, dim3(64 * kWalkWaves), lds_a, stream, a);
  else
    hipLaunchKernelGGL((chain_walk_kernel<0, false>), grid_a, dim3(64 * kWalkWaves), lds_a, stream, a);
  auto combine = [&](auto cc) {
    constexpr int C = decltype(cc)::value;
    if (tropical) {
      hipLaunchKernelGGL(chain_group_trop_kernel<C>, dim3(a.num_grps), dim3(64), 0, stream, a);
      hipLaunchKernelGGL((chain_combine_trop_kernel<C, 0>), dim3(a.num_lats, 2), dim3(64), 0, stream, a);
      hipLaunchKernelGGL((chain_combine_trop_kernel<C, 1>), dim3(a.num_grps, 2), dim3(64), 0, stream, a);
    } else {
      hipLaunchKernelGGL(chain_group_kernel<C>, dim3(a.num_grps), dim3(64), 0, stream, a);
      hipLaunchKernelGGL((chain_combine_kernel<C, 0>), dim3(a.num_lats, 2), dim3(64), lds_b, stream, a);
      hipLaunchKernelGGL((chain_combine_kernel<C, 1>), dim3(a.num_grps, 2), dim3(64), lds_b, stream, a);
    }
  };
  switch (a.c) {
    case 2: combine(std::integral_constant<int, 2>()); break;
    case 4: combine(std::integral_constant<int, 4>()); break;
    case 8: combine(std::integral_constant<int, 8>()); break;
    default: combine(std::integral_constant<int, 16>()); break;
  }
  if (tropical)
    hipLaunchKernelGGL((chain_walk_kernel<1, true>), grid_c, dim3(64 * kSweepWaves), lds_c, stream, a);
  else
    hipLaunchKernelGGL((chain_walk_kernel<1, false>), grid_c, dim3(64 * kSweepWaves), lds_c, stream, a);
  return hipGetLastError();
}

}  // namespace klat
