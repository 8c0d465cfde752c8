// klat_sweep.hip -- the forward/backward lockstep sweep kernel (gfx950, wave64, no MFMA).
//
// Replaces the CPU graph walk of ComputeLatticeAlphasAndBetas [upstream Kaldi, SURVEY.md A.1],
// called at latbin/lattice-to-word-frame-post.cc:88-90, kwsbin2/lattice-word-index-segment.cc:63
// and kwsbin2/lattice-word-index-position.cc:66.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "klat_device.h"
#include "klat_kernels.h"
#include "klat_math.h"

namespace klat {

// ---------------------------------------------------------------------------------------------
// cross-lane helpers for the G lanes of one lattice (G consecutive lanes).  Quad permutes and
// row mirrors are DPP modifiers (no LDS traffic); 32/64-lane groups fall back to ds_bpermute.
template <int CTRL>
__device__ __forceinline__ int dpp_int(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_double(double v) {
  const int lo = dpp_int<CTRL>(__double2loint(v));
  const int hi = dpp_int<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
static constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
static constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
static constexpr int kDppHalfMirror = 0x141;  // lane i <- lane 7-i  (within 8)
static constexpr int kDppMirror = 0x140;      // lane i <- lane 15-i (within 16)

template <int G>
__device__ __forceinline__ int group_max(int v) {
  if (G >= 2) v = max(v, dpp_int<kDppXor1>(v));
  if (G >= 4) v = max(v, dpp_int<kDppXor2>(v));
  if (G >= 8) v = max(v, dpp_int<kDppHalfMirror>(v));
  if (G >= 16) v = max(v, dpp_int<kDppMirror>(v));
  if (G >= 32) v = max(v, __shfl_xor(v, 16, 64));
  if (G >= 64) v = max(v, __shfl_xor(v, 32, 64));
  return v;
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  if (G >= 2) v += dpp_double<kDppXor1>(v);
  if (G >= 4) v += dpp_double<kDppXor2>(v);
  if (G >= 8) v += dpp_double<kDppHalfMirror>(v);
  if (G >= 16) v += dpp_double<kDppMirror>(v);
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int G>
__device__ __forceinline__ double group_min(double v) {
  if (G >= 2) v = fmin(v, dpp_double<kDppXor1>(v));
  if (G >= 4) v = fmin(v, dpp_double<kDppXor2>(v));
  if (G >= 8) v = fmin(v, dpp_double<kDppHalfMirror>(v));
  if (G >= 16) v = fmin(v, dpp_double<kDppMirror>(v));
  if (G >= 32) v = fmin(v, __shfl_xor(v, 16, 64));
  if (G >= 64) v = fmin(v, __shfl_xor(v, 32, 64));
  return v;
}

// Workgroup barriers of the producer/consumer pair.  __syncthreads() would drain vmcnt(0): the
// producer would then wait at every barrier for the stream loads it has just prefetched (a full
// HBM latency per period).  Only LDS has to be ordered here, so the producer keeps its newest
// KEEP global operations (the period's prefetch loads) in flight; everything older -- including
// the previous period's copy-out stores, which FAR reads two periods later rely on -- is done.
template <int KEEP>
__device__ __forceinline__ void producer_barrier() {
  static_assert(KEEP >= 3 && KEEP < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
}
__device__ __forceinline__ void consumer_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// RingLayout: a ring cell is a mantissa (double) and an exponent (int32) kept in two LDS arrays
// indexed by the same cell number.  As 16-byte records (m, e, pad) the 64 lanes of a read hit only
// a quarter of the banks -- the kernel ran with LDS 57 % busy, two thirds of it bank conflicts
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE); the split arrays spread a wave's cells over all banks.

// (ln_table_index / table_ln: klat_math.h)

// One staged term: what the producer wave hands to the consumer wave through LDS.
struct __attribute__((aligned(16))) StagedTerm {
  double wm;     // 2^frac part of exp(-cost); 0 for dead / padding terms
  int32_t k;     // binary exponent of exp(-cost); unspecified when wm == 0
  uint32_t ctl;  // the stream word (klat_device.h), FAR terms resolved
};
// one 16-byte LDS write (hipcc splits the struct assignment into two 8-byte halves, which hit the
// banks with a two-way conflict)
__device__ __forceinline__ void store_staged(StagedTerm* dst, const StagedTerm& t) {
  *reinterpret_cast<int4*>(dst) = make_int4(__double2loint(t.wm), __double2hiint(t.wm), t.k, (int)t.ctl);
}
// LAST of a term word as 0 / 1 in one instruction (hipcc turns the C expression, and the bfe builtin,
// into a shift and an AND)
static_assert(kLast == 1u << 28, "last_bit() extracts bit 28");
__device__ __forceinline__ int last_bit(uint32_t w) {
  int r;
  asm("v_bfe_u32 %0, %1, 28, 1" : "=v"(r) : "v"(w));
  return r;
}
static constexpr uint32_t kCtlFirst = kFirst;
static constexpr uint32_t kCtlLast = kLast;
static constexpr uint32_t kCtlBypass = kBypass;
static constexpr uint32_t kCtlOne = kOne;

// ---------------------------------------------------------------------------------------------
// Lockstep sweep.  One workgroup of TWO waves per job; a job is 64/G lattices, G lanes each,
// walking the forward (dir 0) or backward (dir 1) arc stream of those lattices round by round.
//
//   wave 0, PRODUCER  does everything that does not depend on state values: reads the arc stream
//                     (the only HBM read stream: 12 B per term, read once, 1 KiB dwordx4 wave loads,
//                     prefetched kPrefetchChunks chunks = 16 rounds ahead in registers); evaluates
//                     per term the cost (ScaleLattice / insertion penalty fused) and the 2^k * wm
//                     split of exp(-cost); and turns the states the consumer finished one period
//                     ago into natural log and stores them (all G lanes of a lattice in parallel).
//                     It stages {wm, k, ctl} of the next period in LDS.  About 30 VALU
//                     instructions per term: the kernel is bound by instruction issue, and every
//                     one of them counts.
//   wave 1, CONSUMER  owns the serial dependency chain and touches only LDS: value of the source
//                     state -> multiply -> ldexp -> add into the lane's accumulator -> (G lanes
//                     combined with DPP) -> frexp -> ring cell; the result is also forwarded IN
//                     REGISTERS to the next round.
// The two waves sit on different SIMDs of one CU and meet at one workgroup barrier per period of
// CPB chunks (double-buffered stage).
//
// The values of the last W visited states of each lattice live in an LDS ring of 16-byte cells
// ([slot][lattice]).  The ring read of round r+1 is issued before round r's result is written and
// is patched from registers when it names the state round r just finished (BYPASS), so the LDS
// write->read latency is off the chain.  Chunks in which some lane has a constant-one source take
// an instantiation of the round with one more select.  A source older than W states (FAR) never
// reaches the consumer: the producer reads its stored natural-log value back from alpha / beta
// and stages the term as a constant-one source with the cost c - v (see put_far).
//
// Two instantiations share this body:
//   FAST   every term of a state is aligned to ONE reference exponent -- that of the lattice's
//          previous non-zero state -- so a term costs multiply + ldexp + add and the G lanes
//          combine with a plain sum.  A term whose exponent falls outside +-900 of the reference
//          (arc costs hundreds of nats apart inside one state) sets the job's flag.
//   !FAST  exact for any input: every lane carries its own (mantissa, exponent) accumulator and
//          re-references on demand.  It runs afterwards and only for flagged jobs.
// CPB = stream chunks (of 4 rounds) staged per barrier: 2 when W >= 32, else 1.  FARP (CPB = 2 only):
// the values of sources older than the ring are prefetched one period ahead (see put_far).
// TROPICAL: the (min,+) twin over the same streams and the same machinery -- the staged term is
// the arc cost itself, a state is the cheapest cost of reaching it (no exponents, no logarithm,
// nothing to flag), alpha / beta receive the Viterbi forward / backward costs
// (ComputeLatticeBeam / PruneLattice, latbin/lattice-prune-dyn-beam.cc:27-90).
// D = chunks of the stream a producer lane holds in registers ahead of the one it stages: 4 (16 rounds,
// ~112 VGPRs: seven workgroups per CU) or 8 (32 rounds, 157 VGPRs: six per CU) -- the deep form is
// worth 3-4 % where the batch leaves room for it (launch_pair).
template <int G, bool FAST, int CPB, bool FARP, bool TROPICAL = false, int D = kPrefetchChunks>
__global__ __launch_bounds__(128) void sweep_kernel(SweepArgs args) {
  extern __shared__ double smem[];
  constexpr int LPW = 64 / G;
  static_assert(kDeepPrefetchChunks % D == 0, "jobs are padded to multiples of the deepest prefetch");
  const int W = args.ring_slots;
  if (args.run_if != nullptr && __hip_atomic_load(args.run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  if (!FAST && args.job_flags[blockIdx.x] == 0) return;  // workgroup-uniform

  // ring cells [W][LPW] plus LPW dummy cells (one per lattice: its lanes without a state to commit
  // all write there; nothing reads it), mantissas and exponents in separate
  // arrays (see RingLayout); then the stage buffers [2][CPB][4][64]
  const int ncell = W * LPW + LPW;  // ring cells + one dummy cell per lattice slot
  double* ringM = smem;
  int32_t* ringE = reinterpret_cast<int32_t*>(smem + ncell);
  StagedTerm* stage = reinterpret_cast<StagedTerm*>(ringE + ncell);
  LnTabEntry* lntab = reinterpret_cast<LnTabEntry*>(stage + 2 * CPB * 256);  // [kLnTabSize], producer only
  const int lane = threadIdx.x & 63;
  const int slot = lane / G;
  const int sub = lane % G;
  const SweepJob job = args.jobs[blockIdx.x];
  const bool swapped = (args.role_mask & 0x100u) ? job.dir == 0 : (args.role_mask & 0x200u) ? job.dir != 0 : ((args.role_mask >> ((blockIdx.x >> 8) & 7)) & 1u) != 0;
  const bool producer = (threadIdx.x < 64) != swapped;  // wave 0, or wave 1 where the mask says so
  const int nchunks = job.nchunks;
  // long chains first: both waves of a job on the batch's critical path (priority tag 3, skewed
  // batches only) outrank the waves they share a SIMD with
  const bool critical = (job.tag & kJobPrioMask) == kJobPrioMask;
  if (critical) __builtin_amdgcn_s_setprio(3);
  const LaneInfo li = args.lanes[job.lane_tab + slot];
  const int S = li.num_states;
  const bool bwd = job.dir != 0;
  // position q of the stream -> element of the lattice's padded output row: forward walks the
  // states upwards (position S is the virtual super-final state, whose value is tot_forward and
  // lands in the row's extra slot), backward downwards from S-1
  double* vals = (bwd ? args.beta : args.alpha) + padded_state_base(li.state_base, li.lattice);
  const int q_off = bwd ? S - 1 : 0;
  const int q_step = bwd ? -1 : 1;

  if (producer) {
    // ------------------------------------------------------------------ wave 0
    if (!TROPICAL)
      for (int i = lane; i < kLnTabSize; i += 64) lntab[i] = args.ln_table[i];
    const bool live = li.state_base >= 0;
    const bool early_copy = W < 12 * CPB;  // see the period schedule below

    // cost_mode: 0 = stream_cost(), 1 = arc_cost() (StreamCost::general), 2 = graph scale 1 and no
    // insertion penalty (StreamCost::plain: the usual --acoustic-scale-only call): the graph cost is
    // the stored float itself, three instructions per term less; 3 = FUSED: the packer applied the scales
    // (klat_pack_options.fuse_scales), a slot holds (float)cost and a 16-bit residual -- 10 bytes instead of 12
    auto run = [&](auto cost_mode) {
      constexpr bool GENERAL = decltype(cost_mode)::value == 1;
      constexpr bool PLAIN = decltype(cost_mode)::value == 2;
      constexpr bool FUSED = decltype(cost_mode)::value == 3;
      const double lo_unit = args.lo_unit;
      const ScaleParams sp = args.sp;
      const StreamCost sc = args.sc;
      // The stream stores the units (16 B = 4 rounds) of the first AL lanes of every lattice only
      // (klat_device.h): a chunk row is AL * LPW units, this lane's unit is sub * LPW + slot.  A lane
      // beyond AL loads nothing; its padding words are made up when the chunk is staged.
      const int AL = job_stored_lanes(job.tag);
      const int row = AL * LPW;
      const bool stored = sub < AL;
      const uint64_t stored_mask = lane_mask(stored);
      const double c9v = arc_weight_c9();  // (kept in a vector register for the whole job)
      // (a lane beyond AL reads its lattice's first unit again -- the same cache line, no HBM
      // traffic, and no branch around the prefetch loads -- and ignores it)
      const int unit = (stored ? sub : 0) * LPW + slot;
      const int4* __restrict__ sw = reinterpret_cast<const int4*>(args.sw) + job.chunk_off + unit;
      const float4* __restrict__ sg = reinterpret_cast<const float4*>(args.sg) + job.chunk_off + unit;
      const float4* __restrict__ sa = reinterpret_cast<const float4*>(args.sa) + job.chunk_off + unit;
      const int2* __restrict__ sl = reinterpret_cast<const int2*>(args.sl) + job.chunk_off + unit;
      int4 wb[D];
      float4 gb[D], ab[D];
      auto load_chunk = [&](int j, int64_t cn) {
        wb[j] = sw[cn * row];
        gb[j] = sg[cn * row];
        if (FUSED) {  // four residuals in 8 bytes; handed on as the bit patterns of their integers
          const int2 l = sl[cn * row];
          ab[j] = make_float4(__int_as_float((int)(short)(l.x & 0xFFFF)), __int_as_float(l.x >> 16),
                              __int_as_float((int)(short)(l.y & 0xFFFF)), __int_as_float(l.y >> 16));
        } else {
          ab[j] = sa[cn * row];
        }
      };
#pragma unroll
      for (int j = 0; j < D; ++j) load_chunk(j, min(j, nchunks - 1));
      int p = 0;                  // position the lattice is accumulating at the current round
      // The consumer works on period n between barriers n and n+1, so the iteration that stages
      // period n can copy out the states of period n-2: out_* is that range, pend_* the range of
      // period n-1 (still being worked on).
      int out_p0 = 0, out_n = 0, pend_p0 = 0, pend_n = 0;

      auto weigh = [&](StagedTerm& t, uint32_t w, float g, float a) {
        const double c = FUSED     ? fma((double)__float_as_int(a), lo_unit, (double)g)
                         : GENERAL ? arc_cost(g, a, (w & kWord) != 0, sp)
                         : PLAIN   ? (double)g + (double)(sc.as * a)
                                   : stream_cost(w, g, a, sc);
        if (TROPICAL) {
          t.wm = (c < 1.0e30 && stored) ? c : INFINITY;  // Zero weights, NaN, padding
          t.k = 0;
        } else {
          bool dead;
          arc_weight(c, t.wm, t.k, dead, stored_mask, c9v);
        }
      };
      auto put = [&](StagedTerm* dst, uint32_t w, float g, float a) {
        StagedTerm t;
        weigh(t, w, g, a);
        t.ctl = w;
        store_staged(dst, t);
        p += last_bit(w);
      };
      // The same for a chunk in which some lane's source is older than the ring (FAR).  The
      // source's natural-log value v is already in alpha / beta -- this very wave stored it, in
      // program order before this load -- so the term becomes a constant-one source with the
      // cost c - v: exp(-(c - v)) = exp(v) * exp(-c), formed off the chain with the same routine
      // as every other weight.  The consumer never sees FAR.
      // v = the source's value when the term is FAR (ignored otherwise)
      auto put_far_v = [&](StagedTerm* dst, uint32_t w, float g, float a, double v) {
        StagedTerm t;
        double c = FUSED     ? fma((double)__float_as_int(a), lo_unit, (double)g)
                   : GENERAL ? arc_cost(g, a, (w & kWord) != 0, sp)
                   : PLAIN   ? (double)g + (double)(sc.as * a)
                             : stream_cost(w, g, a, sc);
        const bool last = (w & kLast) != 0;
        uint32_t ctl = w;
        if (w & kFar) {
          if (TROPICAL)
            c = c + v;  // the cost of the path through the source; a constant-zero source below
          else
            c = (v == -INFINITY) ? INFINITY : c - v;  // a log-zero source kills the term
          ctl = (w & ~(kRelMask | kFar)) | kOne;     // RD cell 0: ignored for a constant-one source
          if (last && sub == 0)
            ctl |= kCommit | ((uint32_t)((p & (W - 1)) * LPW + slot) << kWrShift);
          else
            ctl |= (uint32_t)(W * LPW + slot) << kWrShift;  // the lattice's dummy cell
        }
        if (TROPICAL) {
          t.wm = (c < 1.0e30 && stored) ? c : INFINITY;
          t.k = 0;
        } else {
          bool dead;
          arc_weight(c, t.wm, t.k, dead, stored_mask, c9v);
        }
        t.ctl = ctl;
        store_staged(dst, t);
        p += last ? 1 : 0;
      };
      // the value read when the term is staged (L1 bypassed: the line may have been cached before
      // this wave stored the value) ...
      auto put_far = [&](StagedTerm* dst, uint32_t w, float g, float a) {
        double v = 0.0;
        if (w & kFar) {
          const int q = p - (int)(w & kRelMask);
          v = __hip_atomic_load(&vals[q_off + q_step * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        put_far_v(dst, w, g, a, v);
      };
      // ... or, FARP, one period ahead: right after the copy-out stores of the iteration that
      // stages period n, the values the FAR terms of period n+1 name are loaded into fv.  Such a
      // source lies >= W >= 32 positions behind its term, i.e. it was finished in period n-2 or
      // earlier, and the stores of those periods precede these loads in this wave's program order.
      // Always 4*CPB loads per period (the barrier's vmcnt counts on it): rounds without a FAR
      // term, and the periods behind the job's end, read the lane's slot of the sink row.
      double fv[FARP ? CPB : 1][4] = {};
      double* const sink_row = args.sink + (size_t)blockIdx.x * 64 + lane;
      auto far_loads = [&](int c_next, const int4& wn0, const int4& wn1) {
        int pp = p;
#pragma unroll
        for (int i = 0; i < CPB; ++i) {
          const bool inside = c_next + i < nchunks;  // wave-uniform
          const int4& wn = i == 0 ? wn0 : wn1;
          const uint32_t w4[4] = {(uint32_t)wn.x, (uint32_t)wn.y, (uint32_t)wn.z, (uint32_t)wn.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t w = w4[r];
            const int q = pp - (int)(w & kRelMask);
            const double* srcp = (inside && (w & kFar) != 0) ? &vals[q_off + q_step * q] : sink_row;
            fv[i][r] = __hip_atomic_load(srcp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pp += last_bit(w);
          }
        }
      };
      // Copy-out: turns the states the consumer finished during the previous period into natural
      // log (the only log of the sweep, off the chain, spread over the lattice's G lanes) and
      // stores them.  Three phases -- ring cells, table entries, arithmetic + store -- which the
      // period schedule below interleaves with the staging work, so that the two dependent LDS
      // round trips cost no waiting.
      constexpr int kIters = (4 * CPB + G - 1) / G;  // a period finishes at most 4*CPB states
      double cm[kIters];
      int ce[kIters];
      LnTabEntry ct[kIters];
      auto copy_cells = [&]() {
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
          const int j = sub + it * G;
          const int q = out_p0 + (j < out_n ? j : 0);  // idle lanes read some cell of their slot
          const int cell = (q & (W - 1)) * LPW + slot;
          cm[it] = ringM[cell];
          ce[it] = ringE[cell];
        }
      };
      auto copy_table = [&]() {
#pragma unroll
        for (int it = 0; it < kIters; ++it)
          if (!TROPICAL) ct[it] = lntab[ln_table_index(cm[it])];
      };
      // exactly kIters store instructions per period (the barrier's vmcnt counts on it): lanes
      // without a state park their value in the job's sink row
      double* const sink = args.sink + (size_t)blockIdx.x * 64 + lane;
      auto copy_store = [&]() {
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
          const int j = sub + it * G;
          const double v = TROPICAL ? cm[it] : table_ln(cm[it], ce[it], ct[it]);
          double* dst = (live && j < out_n) ? vals + (q_off + q_step * (out_p0 + j)) : sink;
          *dst = v;
        }
      };
      auto stage_chunk = [&](int cj, int4 w4, float4 g4, const float4& a4, const double (&v)[4]) {
        // stage buffer: [barrier period parity][chunk within the period][round][lane]
        const int sidx = (((cj / CPB) & 1) * CPB + (cj % CPB)) * 256 + lane;
        StagedTerm* st = stage + sidx;
        // (a lane beyond the job's stored lanes holds its lattice's first unit a second time: the same
        // flags, cells and source as lane 0 of the lattice -- it reads what that lane reads and, on a
        // commit, writes the same value to the same cell -- and arc_weight's `live` mask takes its weight away)
        const bool any_far = __any(((uint32_t)(w4.x | w4.y | w4.z | w4.w) & kFar) != 0);
        if (FARP && any_far) {
          put_far_v(st, (uint32_t)w4.x, g4.x, a4.x, v[0]);
          put_far_v(st + 64, (uint32_t)w4.y, g4.y, a4.y, v[1]);
          put_far_v(st + 128, (uint32_t)w4.z, g4.z, a4.z, v[2]);
          put_far_v(st + 192, (uint32_t)w4.w, g4.w, a4.w, v[3]);
        } else if (any_far) {  // rare
          put_far(st, (uint32_t)w4.x, g4.x, a4.x);
          put_far(st + 64, (uint32_t)w4.y, g4.y, a4.y);
          put_far(st + 128, (uint32_t)w4.z, g4.z, a4.z);
          put_far(st + 192, (uint32_t)w4.w, g4.w, a4.w);
        } else {
          put(st, (uint32_t)w4.x, g4.x, a4.x);
          put(st + 64, (uint32_t)w4.y, g4.y, a4.y);
          put(st + 128, (uint32_t)w4.z, g4.z, a4.z);
          put(st + 192, (uint32_t)w4.w, g4.w, a4.w);
        }
      };
      // One period = CPB chunks, staged while the consumer works on the previous period:
      //   ring cells of the states to copy out -> stage chunk 0 -> table entries -> stage chunk 1
      //   -> logs + stores -> prefetch loads for D chunks ahead -> barrier.
      // The iteration that stages period n stores the states of period n-2.  vmcnt counts loads
      // and stores together, in order, so the barrier leaves the newest two periods of this
      // wave's memory operations in flight (a fixed number: kIters stores + 3*CPB loads each) and
      // never waits on a prefetch; nothing but this wave reads what it stores.  A FAR source of a
      // term of period n has at least W - 1 commit rounds between itself and the term: with
      // W >= 12*CPB it was finished in period n-3 or earlier and stored by iteration n-1.  A
      // shorter ring (W = 8) may need a state of period n-2: its copy-out then runs first.
      // nchunks is a multiple of D, D of CPB.
      static_assert(D % CPB == 0, "prefetch depth must be whole periods");
      for (int c = 0; c < nchunks; c += D) {
#pragma unroll
        for (int j0 = 0; j0 < D; j0 += CPB) {
          const int period_p0 = p;
          copy_cells();
          if (early_copy) {
            copy_table();
            copy_store();
          }
          stage_chunk(c + j0, wb[j0], gb[j0], ab[j0], fv[0]);
          if (!early_copy) copy_table();
          if (CPB == 2) stage_chunk(c + j0 + 1, wb[j0 + 1], gb[j0 + 1], ab[j0 + 1], fv[FARP ? 1 : 0]);
          if (!early_copy) copy_store();
          if (FARP) far_loads(c + j0 + CPB, wb[(j0 + CPB) % D], wb[(j0 + CPB + 1) % D]);
#pragma unroll
          for (int i = 0; i < CPB; ++i) load_chunk(j0 + i, min(c + j0 + i + D, nchunks - 1));
          // this period is staged; the consumer has finished the previous one,
          producer_barrier<2 * (3 * CPB + kIters + (FARP ? 4 * CPB : 0))>();
          out_p0 = pend_p0;             // whose states are now stable in the ring (W >= 2 periods of states)
          out_n = pend_n;
          pend_p0 = period_p0;
          pend_n = p - period_p0;
        }
      }
      copy_cells();
      copy_table();
      copy_store();
      __syncthreads();  // the consumer has finished the last period
      out_p0 = pend_p0;
      out_n = pend_n;
      copy_cells();
      copy_table();
      copy_store();
    };
    if (args.fused)
      run(std::integral_constant<int, 3>());
    else if (args.sc.general)
      run(std::integral_constant<int, 1>());
    else if (args.sc.plain)
      run(std::integral_constant<int, 2>());
    else
      run(std::integral_constant<int, 0>());
    return;
  }

  // -------------------------------------------------------------------- wave 1: the chain
  // the chain wave is the critical path of its job: let it win issue arbitration against the
  // producer waves (of other jobs) that share its SIMD -- those spend part of their time at barriers
  if (!critical) __builtin_amdgcn_s_setprio(2);

  // every ring cell starts as log-zero so that padding terms read something well-defined
  for (int i = lane; i < W * LPW; i += 64) {
    ringM[i] = TROPICAL ? INFINITY : 0.0;
    ringE[i] = kExpMin;
  }

  double accM = 0.0;   // lane accumulator: mantissa scaled by 2^accE (!FAST) or 2^refE (FAST)
  int accE = kExpMin;
  int refE = 0;        // FAST: exponent of the lattice's previous non-zero state
  bool sticky = false; // FAST: some term left the safe exponent window
  double lastM = TROPICAL ? INFINITY : 0.0;  // value of the state finished by the previous round (register bypass)
  int lastE = kExpMin;
  double preM = 0.0;   // ring read for the upcoming round, software-pipelined one round ahead
  int preE = kExpMin;

  // SPECIAL = this chunk may hold constant-one sources; PRE = issue the next round's ring
  // read (the last round of a chunk cannot: its successor is staged behind the barrier)
  auto round = [&](auto special, auto do_pre, const StagedTerm& t, uint32_t ctl_next) {
    const uint32_t ctl = t.ctl;
    const bool last = (ctl & kCtlLast) != 0;
    const bool first = (ctl & kCtlFirst) != 0;
    if constexpr (TROPICAL) {
      // source cost (ring, register bypass or the constant 0) + arc cost -> min over the state's
      // terms -> min over the lattice's G lanes -> ring cell and next round's bypass
      double from = preM;
      if (decltype(do_pre)::value) preM = ringM[(int)(ctl_next & kCellMask)];
      from = select64(lane_mask((ctl & kCtlBypass) != 0), lastM, from);
      if (decltype(special)::value) from = (ctl & kCtlOne) ? 0.0 : from;
      const double tc = from + t.wm;
      accM = select64(lane_mask(first), tc, fmin(accM, tc));
      const double best = G > 1 ? group_min<G>(accM) : accM;
      lastM = best;
      ringM[(int)((ctl >> kWrShift) & kCellMask)] = best;
      (void)last;
      return;
    }
    double sm = preM;
    int se = preE;
    // issue next round's ring read now, ahead of this round's ring write
    if (decltype(do_pre)::value) {
      const int cell = (int)(ctl_next & kCellMask);
      preM = ringM[cell];
      preE = ringE[cell];
    }
    const uint64_t bypass = lane_mask((ctl & kCtlBypass) != 0);
    sm = select64(bypass, lastM, sm);
    se = select32(bypass, lastE, se);
    if (decltype(special)::value) {
      const bool one = (ctl & kCtlOne) != 0;  // constant one = 0.5 * 2^1
      sm = one ? 0.5 : sm;
      se = one ? 1 : se;
    }
    // dead (zero weight / padding) terms have wm == 0 and zero sources sm == 0: tm is then 0 and
    // whatever the exponents hold is ignored (unsigned arithmetic: t.k is unspecified for them)
    const double tm = sm * t.wm;
    const int te_raw = (int)((uint32_t)se + (uint32_t)t.k);
    double cM;
    int cE;
    if (FAST) {
      const int shift = (int)((uint32_t)te_raw - (uint32_t)refE);
      sticky |= (tm != 0.0) & ((unsigned)(shift + 900) > 1800u);
      // (no test of FIRST: the round that closed the previous state left 0 in the accumulator -- one
      // compare less per round, and the select is off the chain)
      accM += ldexp(tm, shift);
      cM = G > 1 ? group_sum<G>(accM) : accM;
      cE = refE;
    } else {
      // only the exponent tag needs forcing so that a zero never becomes a reference exponent
      const int te = (tm == 0.0) ? kExpMin : te_raw;
      const int ref = first ? te : accE;
      const int shift = te - ref;
      if (__any(shift > 900)) {
        // a later term dwarfs the lane's reference exponent -> re-reference (exact form)
        if (first) {
          accM = tm;
          accE = te;
        } else {
          me_accumulate(accM, accE, tm, te);
        }
      } else {
        accM = (first ? 0.0 : accM) + ldexp(tm, shift);
        accE = ref;
      }
      cM = accM;
      cE = accE;
      if (G > 1) {
        cE = group_max<G>(accE);
        cM = group_sum<G>(ldexp(accM, accE - cE));
      }
    }
    // close the state: computed every round (straight-line), committed when LAST
    const double m = __builtin_amdgcn_frexp_mant(cM);
    int e;
    if (FAST) {
      // a zero state keeps the reference exponent as its (irrelevant) exponent: frexp gives 0, 0
      e = cE + __builtin_amdgcn_frexp_exp(cM);
      const uint64_t closing = lane_mask(last);
      refE = select32(closing, e, refE);
      accM = select64(closing, 0.0, accM);
    } else {
      e = (cM != 0.0) ? cE + __builtin_amdgcn_frexp_exp(cM) : kExpMin;
    }
    lastM = m;
    lastE = e;
    {  // unpredicated: lanes that do not commit a state write their own dummy cell behind the ring
      const int cell = (int)((ctl >> kWrShift) & kCellMask);
      ringM[cell] = m;
      ringE[cell] = e;
    }
  };

  const std::true_type yes;
  const std::false_type no;
  // four rounds; LAST_PRE = the chunk that follows belongs to the same period, so its first
  // round's ring read (cell in next_ctl) can be issued from this chunk's last round
  auto chunk = [&](const StagedTerm& t0, const StagedTerm& t1, const StagedTerm& t2, const StagedTerm& t3,
                   auto last_pre, uint32_t next_ctl) {
    if (__any(((t0.ctl | t1.ctl | t2.ctl | t3.ctl) & kCtlOne) != 0)) {
      round(yes, yes, t0, t1.ctl);
      round(yes, yes, t1, t2.ctl);
      round(yes, yes, t2, t3.ctl);
      round(yes, last_pre, t3, next_ctl);
    } else {
      round(no, yes, t0, t1.ctl);
      round(no, yes, t1, t2.ctl);
      round(no, yes, t2, t3.ctl);
      round(no, last_pre, t3, next_ctl);
    }
  };
  for (int c = 0; c < nchunks; c += CPB) {
    consumer_barrier();  // the period is staged (and, for c == 0, the ring is initialised)
    // the whole period's terms are read at once: one LDS round trip per period, not per chunk
    const int sidx = ((c / CPB) & 1) * CPB * 256 + lane;
    const StagedTerm* st = stage + sidx;
    const StagedTerm a0 = st[0], a1 = st[64], a2 = st[128], a3 = st[192];
    {  // after the previous period's last ring write: the fresh value
      const int cell = (int)(a0.ctl & kCellMask);
      preM = ringM[cell];
      preE = ringE[cell];
    }
    if (CPB == 2) {
      const StagedTerm b0 = st[256], b1 = st[320], b2 = st[384], b3 = st[448];
      chunk(a0, a1, a2, a3, yes, b0.ctl);
      chunk(b0, b1, b2, b3, no, 0u);
    } else {
      chunk(a0, a1, a2, a3, no, 0u);
    }
  }
  __syncthreads();  // lets the producer copy out the last period's states
  if (FAST && !TROPICAL) {
    // every job's flag is written, set or not: no clearing pass before the launch
    const bool flagged = __any(sticky);
    if (lane == 0) args.job_flags[blockIdx.x] = flagged ? 1 : 0;
  }
}

static size_t sweep_lds_bytes(int ring_slots, int LPW, int cpb) {
  return ((size_t)ring_slots * LPW + LPW) * (sizeof(double) + sizeof(int32_t)) +
         (size_t)2 * cpb * 256 * sizeof(StagedTerm) + kLnTabSize * sizeof(LnTabEntry);
}

template <int G>
static void launch_pair(const SweepArgs& args, int num_jobs, hipStream_t stream, hipEvent_t fast_begin, hipEvent_t fast_end, int which) {
  const int LPW = 64 / G;
  // the producer copies a period's states out one period later: the ring must hold two periods
  const int cpb = args.ring_slots >= 32 ? 2 : 1;
  const size_t lds = sweep_lds_bytes(args.ring_slots, LPW, cpb);
  auto launch_one = [&](auto kernel) {
    // the largest rings (4032 cells) take the workgroup past the default 64 KiB of dynamic LDS
    if (lds > 60 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(num_jobs), dim3(128), lds, stream, args);
  };
  auto launch = [&](auto kernel) {  // the exact sweep
    if (which & kSweepExact) launch_one(kernel);
  };
  auto launch_fast = [&](auto kernel) {
    if (!(which & kSweepFast)) return;
    if (fast_begin) (void)hipEventRecord(fast_begin, stream);
    launch_one(kernel);
    if (fast_end) (void)hipEventRecord(fast_end, stream);
  };
  if (cpb == 2 && args.far_prefetch) {
    launch_fast(&sweep_kernel<G, true, 2, true>);
    launch(&sweep_kernel<G, false, 2, true>);
  } else if (cpb == 2 && args.deep_prefetch) {
    launch_fast(&sweep_kernel<G, true, 2, false, false, kDeepPrefetchChunks>);
    launch(&sweep_kernel<G, false, 2, false>);
  } else if (cpb == 2) {
    launch_fast(&sweep_kernel<G, true, 2, false>);
    launch(&sweep_kernel<G, false, 2, false>);
  } else {
    launch_fast(&sweep_kernel<G, true, 1, false>);
    launch(&sweep_kernel<G, false, 1, false>);
  }
}

template <int G>
static void launch_tropical(const SweepArgs& args, int num_jobs, hipStream_t stream) {
  const int LPW = 64 / G;
  const int cpb = args.ring_slots >= 32 ? 2 : 1;
  const size_t lds = sweep_lds_bytes(args.ring_slots, LPW, cpb);
  auto launch = [&](auto kernel) {
    if (lds > 60 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(num_jobs), dim3(128), lds, stream, args);
  };
  if (cpb == 2 && args.far_prefetch)
    launch(&sweep_kernel<G, true, 2, true, true>);
  else if (cpb == 2)
    launch(&sweep_kernel<G, true, 2, false, true>);
  else
    launch(&sweep_kernel<G, true, 1, false, true>);
}

// The (min,+) twin: args.alpha / args.beta receive the Viterbi forward / backward costs (padded rows
// like alpha / beta); job_flags and ln_table are not used.
hipError_t launch_sweep_tropical(const SweepArgs& args, int num_jobs, int G, hipStream_t stream) {
  if (num_jobs <= 0) return hipSuccess;
  const int LPW = 64 / G;
  if ((size_t)args.ring_slots * LPW + 64 > kCellMask + 1) return hipErrorInvalidValue;
  switch (G) {
    case 1: launch_tropical<1>(args, num_jobs, stream); break;
    case 2: launch_tropical<2>(args, num_jobs, stream); break;
    case 4: launch_tropical<4>(args, num_jobs, stream); break;
    case 8: launch_tropical<8>(args, num_jobs, stream); break;
    case 16: launch_tropical<16>(args, num_jobs, stream); break;
    case 32: launch_tropical<32>(args, num_jobs, stream); break;
    case 64: launch_tropical<64>(args, num_jobs, stream); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Runs the fast sweep over every job (it writes every job's flag) and the exact sweep over the jobs the
// fast one flagged (normally none: those workgroups exit at once).
hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream, hipEvent_t fast_begin,
                        hipEvent_t fast_end, int which) {
  if (num_jobs <= 0) {  // nothing to sweep: the timing events must still be recorded
    hipError_t e = fast_begin ? hipEventRecord(fast_begin, stream) : hipSuccess;
    if (e == hipSuccess && fast_end) e = hipEventRecord(fast_end, stream);
    return e;
  }
  const int LPW = 64 / G;
  if ((size_t)args.ring_slots * LPW + 64 > kCellMask + 1) return hipErrorInvalidValue;
  switch (G) {
    case 1: launch_pair<1>(args, num_jobs, stream, fast_begin, fast_end, which); break;
    case 2: launch_pair<2>(args, num_jobs, stream, fast_begin, fast_end, which); break;
    case 4: launch_pair<4>(args, num_jobs, stream, fast_begin, fast_end, which); break;
    case 8: launch_pair<8>(args, num_jobs, stream, fast_begin, fast_end, which); break;
    case 16: launch_pair<16>(args, num_jobs, stream, fast_begin, fast_end, which); break;
    case 32: launch_pair<32>(args, num_jobs, stream, fast_begin, fast_end, which); break;
    case 64: launch_pair<64>(args, num_jobs, stream, fast_begin, fast_end, which); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace klat
