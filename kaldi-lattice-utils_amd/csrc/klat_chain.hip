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

constexpr int kTP = 256;  // chain_prep_kernel
// Waves per workgroup of the walks.  The waves share nothing: four to a workgroup only because the dispatcher
// spreads the waves of ONE workgroup over a CU's four SIMDs, while single-wave workgroups piled up on one
// (measured: a CU finished a step every 104 cycles whether it held 4 or 16 such workgroups).
constexpr int kWalkWaves = 4;

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
__device__ __forceinline__ uint64_t chain_record(double cost, uint32_t ctl, double c9v, int32_t* status) {
  if (!(cost < 1.0e30)) return (uint64_t)__double_as_longlong(1.0) | (ctl & (kChainLast | kChainFinal)) | kChainSpecial | kChainWhich;
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
  for (int k = tid; k < cl.num_segs; k += kTP) {
    const int b0 = k * a.seg_len, b1 = (k == cl.num_segs - 1) ? S : (k + 1) * a.seg_len;
    ChainSegment sg;
    sg.clat = blockIdx.x, sg.k = k, sg.first_state = b0, sg.num_states = b1 - b0;
    sg.rf_begin = cl.rec_f + (uint32_t)pf[b0], sg.rf_count = (uint32_t)(pf[b1] - pf[b0]);
    sg.rb_begin = cl.rec_b + (uint32_t)(Tb - pb[b1]), sg.rb_count = (uint32_t)(pb[b1] - pb[b0]);
    a.segs[cl.seg_begin + k] = sg;
  }
  const double c9v = arc_weight_c9();
  const int32_t* in_arc = a.in_arc + a0;
  const int32_t* out_arc = a.out_arc + a0;
  for (int s = tid; s < S; s += kTP) {
    const bool fin = is_final(s);
    const double fcost = fin ? arc_cost(a.final_graph[s0 + s], a.final_acoustic[s0 + s], false, a.sp) : 0.0;
    {
      uint64_t* r = a.recs + cl.rec_f + pf[s];
      const int i0 = cnt[s], i1 = cnt[s + 1];
      const int nterm = i1 - i0 + (s == 0 ? 1 : 0);
      if (s == 0) *r++ = chain_record(0.0, kChainSpecial | (nterm == 1 ? kChainLast : 0), c9v, a.status);
      for (int i = i0; i < i1; ++i) {
        const int64_t x = a0 + in_arc[i];
        const double cost = arc_cost(a.graph[x], a.acoustic[x], a.ilabel[x] != 0, a.sp);
        *r++ = chain_record(cost, (uint32_t)(s - a.src[x]) | (i == i1 - 1 ? kChainLast : 0), c9v, a.status);
      }
      if (nterm == 0) *r++ = chain_record(0.0, kChainSpecial | kChainWhich | kChainLast, c9v, a.status);
      if (fin) *r++ = chain_record(fcost, 1u | kChainFinal, c9v, a.status);
    }
    {
      uint64_t* r = a.recs + cl.rec_b + (Tb - pb[s + 1]);
      const int o0 = row[s], o1 = row[s + 1];
      const int nterm = o1 - o0 + (fin ? 1 : 0);
      if (fin) {
        uint64_t rec = chain_record(fcost, nterm == 1 ? kChainLast : 0, c9v, a.status);
        if (!(rec & kChainSpecial)) rec |= kChainSpecial;  // (a dead final weight already reads the constant zero)
        *r++ = rec;
      }
      for (int i = o0; i < o1; ++i) {
        const int64_t x = a0 + out_arc[i];
        const double cost = arc_cost(a.graph[x], a.acoustic[x], a.ilabel[x] != 0, a.sp);
        *r++ = chain_record(cost, (uint32_t)(a.dst[x] - s) | (i == o1 - 1 ? kChainLast : 0), c9v, a.status);
      }
      if (nterm == 0) *r++ = chain_record(0.0, kChainSpecial | kChainWhich | kChainLast, c9v, a.status);
    }
  }
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
template <int MODE, int DIR>
__device__ __forceinline__ void chain_walk(const ChainArgs& a, double* ring, double* outbuf, const LnTabEntry* lntab) {
  const int lane = threadIdx.x & 63;
  const int job = blockIdx.x * kWalkWaves + (threadIdx.x >> 6);  // one wave = one job, the waves of a workgroup are on their own
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
    int32_t* oute = reinterpret_cast<int32_t*>(outbuf + 16 * 64) + lane;  // exponents beside the output rows
    double* outx = outbuf + lane;
    for (int r = 0; r < W + 2; ++r) col[r * 64] = 0.0;
    col[W * 64] = 1.0;  // row W: the constant one; row W + 1: zero
    double acc = 0.0, lastX = 0.0, finA = 0.0;
    int E = 0, finR = 0;
    int pos = DIR ? end - 1 : first;
    if (MODE == 0) {
      if (sg.k > 0) col[((first - c + comp) & Wm) * 64] = 1.0;  // the lane's own state is one, the others zero
    } else if (DIR == 0) {
      if (sg.k > 0) {  // alpha of the c states before the segment (chain_combine_kernel), already to one exponent
        E = a.bnd_fe[seg];
        for (int i = 0; i < c; ++i) {
          lastX = a.bnd_fm[(size_t)seg * c + i];
          col[((first - c + i) & Wm) * 64] = lastX;
        }
      }
    } else {
      if (!last_seg) E = a.bnd_be[seg];  // (the first values to arrive are the ones injected below)
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
      int cell = (DIR ? pos + rel : pos - rel) & Wm;
      if (a.exp & 1) cell = 0;
      if (SPECIAL && (lo & kChainSpecial)) cell = W + (int)((lo >> 7) & 1u);  // constants
      double x = (a.exp & 4) ? 1.0 : col[cell * 64];
      if (MODE == 1) x = select64(lane_mask(rel == 1), lastX, x);  // the state finished by the previous step: from registers, not through LDS
      if (SPECIAL && (lo & (kChainSpecial | kChainWhich)) == kChainSpecial) {  // the constant one, in the lane's scale
        if (E > 900 || E < -900) trouble |= kChainTroubleOne;
        x = ldexp(1.0, max(-2000, min(2000, -E)));
      }
      if (SPECIAL && (lo & kChainFinal)) {  // a final weight: into the lane's total (transfer), nothing (sweep)
        if (MODE == 0) chain_add_exact(finA, finR, x * w, E);
        x = 0.0;
      }
      acc = fma(x, w, acc);
      if (SPECIAL && MODE == 1 && DIR == 1) {  // what the arcs that leave the segment contribute to its last c states
        const int i = pos - (end - c);
        if (last && !last_seg && i >= 0) {
          const double y = a.bnd_bm[(size_t)seg * c + i];
          const int sh = a.bnd_be[seg] - E;
          if (y != 0.0 && (sh > 900 || sh < -900)) trouble |= kChainTroubleInject;
          acc += ldexp(y, max(-2000, min(2000, sh)));
        }
      }
      // every step leaves the open state's sum in the state's own ring cell (nobody reads it before its LAST)
      if (!(a.exp & 2)) col[((a.exp & 1) ? 1 : (pos & Wm)) * 64] = acc;
      const uint64_t L = lane_mask(last);
      if (MODE == 1) {
        lastX = select64(L, acc, lastX);
        outx[npend * 64] = acc;  // (the slot is written over until a LAST moves on)
        oute[npend * 64] = E;
        npend += last ? 1 : 0;
      }
      acc = select64(L, 0.0, acc);
      pos += last ? (DIR ? -1 : 1) : 0;
    };
    // the largest of the live values (the open state's sum, the states behind it that an arc can reach) back to [0.5, 1)
    auto rescale = [&]() {
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
        const int mx = wave_max_int(npend);
        for (int j = 0; j < mx; ++j) {
          const double x = outx[j * 64];
          const double m = __builtin_amdgcn_frexp_mant(x);
          const int e = oute[j * 64] + __builtin_amdgcn_frexp_exp(x);
          const double v = table_ln(m, e, lntab[ln_table_index(m)]);
          if (j < npend) vals[DIR ? out_pos - j : out_pos + j] = v;
        }
        out_pos = pos;
        npend = 0;
      }
    };

    // The records come a BATCH of kB ahead: eight 16-byte loads per lane at the top of an iteration, used in the
    // next one (a lane reads up to kB records past its segment: the array ends in slack).  Loads inside a branch,
    // one load per step, or a batch of four made every step (or every fourth) wait a memory latency.  The outputs
    // of a batch leave right before the next batch's loads are issued, so the one wait at the bottom (the
    // compiler drains the queue there: loads and stores share its counter) covers both.
    constexpr int kB = 16;
    uint4 cur[kB / 2], nxt[kB / 2];
    auto load_batch = [&](int t, uint4 (&q)[kB / 2]) {
      const uint4* p = reinterpret_cast<const uint4*>(rp + min(t, n));
#pragma unroll
      for (int i = 0; i < kB / 2; ++i) q[i] = p[i];
    };
    int until = a.rescale_iters;
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
    const std::true_type yes;
    const std::false_type no;
    load_batch(0, cur);
    int t0 = 0;
    for (; t0 + kB <= nmin; t0 += kB) {  // every lane has the whole batch
      flush();
      load_batch(t0 + kB, nxt);
      uint32_t any_ctl = 0;
#pragma unroll
      for (int i = 0; i < kB / 2; ++i) any_ctl |= cur[i].x | cur[i].z;
      bool rare = (any_ctl & (kChainSpecial | kChainFinal)) != 0;
      if (MODE == 1 && DIR == 1) rare |= !last_seg && pos >= end - c;
      if (__any(rare))
        body(yes, no, t0);
      else
        body(no, no, t0);
#pragma unroll
      for (int i = 0; i < kB / 2; ++i) cur[i] = nxt[i];
    }
    for (; t0 < nmax; t0 += kB) {  // the segments' ends
      flush();
      load_batch(t0 + kB, nxt);
      body(yes, yes, t0);
#pragma unroll
      for (int i = 0; i < kB / 2; ++i) cur[i] = nxt[i];
    }
    flush();

    if (MODE == 0) {
      // the lane's row of Q: the segment's last c states, scaled once more -- those an arc of this lattice can
      // leave across the cut, that is; the ones behind them (c covers the longest arc of the batch) are never read
      // again and pass as zero; its entry of Qf
      rescale();
      double row_max = 0.0;
      for (int i = 0; i < c; ++i) {
        const double v = i >= c - span ? col[((end - c + i) & Wm) * 64] : 0.0;
        row_max = fmax(row_max, v);
        a.Qm[((size_t)seg * c + comp) * c + i] = v;
      }
      a.Qe[(size_t)seg * c + comp] = row_max != 0.0 ? E : kZeroRowExp;  // (a row of zeros has no scale: it must not set the others')
      a.Qfm[(size_t)seg * c + comp] = __builtin_amdgcn_frexp_mant(finA);
      a.Qfe[(size_t)seg * c + comp] = finR + __builtin_amdgcn_frexp_exp(finA);
    }
  }
  if (trouble) atomicOr(a.flag, trouble);
}

template <int MODE>
__global__ __launch_bounds__(64 * kWalkWaves) void chain_walk_kernel(ChainArgs a) {
  extern __shared__ double smem[];
  LnTabEntry* lntab = reinterpret_cast<LnTabEntry*>(smem);
  const int wave = threadIdx.x >> 6;
  const int per_wave = (a.W + 2) * 64 + (MODE == 1 ? 16 * 64 + 8 * 64 : 0);  // doubles: the ring; the sweep's output rows and their exponents
  double* ring = smem + 2 * kLnTabSize + wave * per_wave;
  double* outbuf = ring + (a.W + 2) * 64;
  if (threadIdx.x < kLnTabSize) lntab[threadIdx.x] = a.ln_table[threadIdx.x];
  __syncthreads();
  if (MODE == 0) {
    chain_walk<0, 0>(a, ring, outbuf, lntab);
  } else if ((blockIdx.y ^ a.dir_first) == 0) {
    chain_walk<1, 0>(a, ring, outbuf, lntab);
  } else {
    chain_walk<1, 1>(a, ring, outbuf, lntab);
  }
}

__device__ __forceinline__ int group_max_int(int v) {  // over the kChainMaxSpan lanes that hold a vector
  for (int d = 1; d < kChainMaxSpan; d <<= 1) v = max(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ double group_sum_f64(double v) {
  for (int d = 1; d < kChainMaxSpan; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}
static constexpr int kNoExp = -(1 << 28);
static constexpr int kCombineChunk = 8;  // segments whose matrices are in LDS together

// One wave per (lattice, direction): the serial part, in the same scaled arithmetic -- lane i < c holds
// entry i of the running vector as a plain double, the vector shares one exponent.  The matrices come through
// LDS a chunk of eight segments at a time, the next chunk's loads in flight while this one is used: a step
// is a few hundred cycles, a trip to memory ten times that.
__global__ __launch_bounds__(64) void chain_combine_kernel(ChainArgs a) {
  extern __shared__ double smem[];
  const ChainLattice cl = a.lats[blockIdx.x];
  const int lane = threadIdx.x, c = a.c, K = cl.num_segs, cc = a.c * a.c;
  const int pd = cc + c, pi = 2 * c;  // per segment: Qm and Qfm; Qe and Qfe
  double* sd = smem;                  // [2][kCombineChunk][pd]
  int32_t* si = reinterpret_cast<int32_t*>(smem + 2 * kCombineChunk * pd);  // [2][kCombineChunk][pi]
  const bool bwd = blockIdx.y != 0;
  const bool mine = lane < c;
  const int me = mine ? lane : 0;
  const int nsteps = bwd ? K - 1 : K;  // forward: k = 0 .. K-1; backward: k = K-1 .. 1
  auto seg_of = [&](int s) { return (size_t)cl.seg_begin + (size_t)(bwd ? K - 1 - s : s); };
  int trouble = 0;
  // stage: registers <- memory (issue), later LDS <- registers
  constexpr int kRD = (kChainMaxSpan * kChainMaxSpan + kChainMaxSpan + 63) / 64;  // doubles per lane and segment
  double rd[kCombineChunk][kRD];
  int32_t ri[kCombineChunk];
  auto issue = [&](int chunk) {
#pragma unroll
    for (int s = 0; s < kCombineChunk; ++s) {
      const size_t seg = seg_of(min(chunk * kCombineChunk + s, nsteps - 1));
#pragma unroll
      for (int r = 0; r < kRD; ++r) {
        const int idx = lane + 64 * r;
        rd[s][r] = idx < cc ? a.Qm[seg * cc + idx] : (idx < pd ? a.Qfm[seg * c + idx - cc] : 0.0);
      }
      ri[s] = lane < c ? a.Qe[seg * c + lane] : (lane < pi ? a.Qfe[seg * c + lane - c] : 0);
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
  if (nsteps <= 0) return;
  issue(0);
  land(0);
  __syncthreads();
  double vec = (!bwd && lane == 0) ? 1.0 : 0.0;  // forward: alpha before the segment; backward: y of the next cut
  int Ev = 0;
  double tot = 0.0;
  int totE = 0;
  const int nchunks = (nsteps + kCombineChunk - 1) / kCombineChunk;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) issue(ch + 1);
    for (int s = 0; s < kCombineChunk && ch * kCombineChunk + s < nsteps; ++s) {
      const int step = ch * kCombineChunk + s;
      const size_t seg = seg_of(step);
      const double* qd = sd + (buf * kCombineChunk + s) * pd;
      const int32_t* qi = si + (buf * kCombineChunk + s) * pi;
      const double qfm = qd[cc + me];
      const int qe = qi[me], qfe = qi[c + me];
      if (!bwd) {
        // forward: alpha of the c states before segment k+1 = alpha before segment k, through Q_k; the total
        // collects every segment's paths into its final states.  (Segment 0: every row of Q_0 is the row of
        // the start state -- the constant-one term feeds all lanes alike -- so entry 0 alone carries weight.)
        const int k = step;
        if (k > 0) {
          if (mine) a.bnd_fm[seg * c + lane] = vec;
          if (lane == 0) a.bnd_fe[seg] = Ev;
        }
        {  // tot += sum_j a[j] Qf[j]
          const double t = mine ? vec * qfm : 0.0;
          const int te = t != 0.0 ? Ev + qfe : kNoExp;
          const int hi = group_max_int(te);
          const double sum = group_sum_f64(t != 0.0 ? ldexp(t, te - hi) : 0.0);
          if (hi != kNoExp) chain_add_exact(tot, totE, sum, hi);
        }
        if (k == K - 1) break;
        // r[j] = a[j] 2^(Qe[j]) to one exponent Er; new[i] = sum_j r[j] Qm[j][i]
        const int re = (mine && vec != 0.0 && qe != kZeroRowExp) ? qe : kNoExp;
        const int Er = group_max_int(re);
        double r = 0.0;
        if (re != kNoExp) {
          if (re - Er < -900) trouble |= kChainTroubleRows;
          r = ldexp(vec, re - Er);
        }
        double nw = 0.0;
#pragma unroll
        for (int j = 0; j < kChainMaxSpan; ++j) {
          const double rj = __shfl(r, j, 64);
          if (j < c && mine) nw = fma(rj, qd[j * c + lane], nw);
        }
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
        // backward: y_k[j] = what the paths that leave state (first state of segment k) - c + j through an arc
        // into segment k contribute to its beta: through Q_k to the y of the next cut, or into a final state
        double val = 0.0;
#pragma unroll
        for (int i = 0; i < kChainMaxSpan; ++i) {
          const double yi = __shfl(vec, i, 64);
          if (i < c && mine) val = fma(qd[lane * c + i], yi, val);
        }
        int ve = qe + Ev;
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
        if (mine) a.bnd_bm[(seg - 1) * c + lane] = vec;
        if (lane == 0) a.bnd_be[seg - 1] = Ev;
      }
    }
    if (ch + 1 < nchunks) {
      land(buf ^ 1);
      __syncthreads();
    }
  }
  if (!bwd && lane == 0) {  // tot_forward, natural log, into the row's extra slot
    const double m = __builtin_amdgcn_frexp_mant(tot);
    const double e = (double)(totE + __builtin_amdgcn_frexp_exp(tot));
    a.alpha[padded_state_base(a.state_off[cl.lattice], cl.lattice) + cl.num_states] = tot == 0.0 ? -INFINITY : log(m) + e * kLn2;
  }
  if (trouble) atomicOr(a.flag, trouble);
}

}  // namespace

hipError_t launch_chain_prep(const ChainPrepArgs& a, int num_chain_lattices, hipStream_t stream) {
  if (num_chain_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(chain_prep_kernel, dim3(num_chain_lattices), dim3(kTP), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_chain_sweep(const ChainArgs& a, hipStream_t stream) {
  if (a.num_segs <= 0) return hipSuccess;
  static const int exp_bits = std::getenv("KLAT_CHAIN_EXP") ? std::atoi(std::getenv("KLAT_CHAIN_EXP")) : 0;
  const_cast<ChainArgs&>(a).exp = exp_bits;
  static const int pad = std::getenv("KLAT_CHAIN_LDS_PAD") ? std::atoi(std::getenv("KLAT_CHAIN_LDS_PAD")) : 0;  // (experiments)
  const size_t ring_bytes = (size_t)(a.W + 2) * 64 * sizeof(double);
  const size_t lds_a = kLnTabSize * sizeof(LnTabEntry) + kWalkWaves * ring_bytes + (size_t)pad;
  const size_t lds_c = kLnTabSize * sizeof(LnTabEntry) + kWalkWaves * (ring_bytes + (16 + 8) * 64 * sizeof(double)) + (size_t)pad;
  const size_t lds_b = (size_t)2 * kCombineChunk * ((size_t)(a.c * a.c + a.c) * sizeof(double) + (size_t)2 * a.c * sizeof(int32_t));
  const int jobs_a = (a.num_segs + (64 >> a.lgc) - 1) / (64 >> a.lgc), jobs_c = (a.num_segs + 63) / 64;
  hipLaunchKernelGGL(chain_walk_kernel<0>, dim3((jobs_a + kWalkWaves - 1) / kWalkWaves), dim3(64 * kWalkWaves), lds_a, stream, a);
  hipLaunchKernelGGL(chain_combine_kernel, dim3(a.num_lats, 2), dim3(64), lds_b, stream, a);
  hipLaunchKernelGGL(chain_walk_kernel<1>, dim3((jobs_c + kWalkWaves - 1) / kWalkWaves, 2), dim3(64 * kWalkWaves), lds_c, stream, a);
  return hipGetLastError();
}

}  // namespace klat
