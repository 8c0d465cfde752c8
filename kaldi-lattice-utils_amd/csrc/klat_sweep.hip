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

// One staged term: what the producer wave hands to the consumer wave through LDS.
struct __attribute__((aligned(16))) StagedTerm {
  double wm;   // 2^frac part of exp(-cost); 0 for dead / padding terms
  int32_t k;   // binary exponent of exp(-cost)
  uint32_t w;  // stream word (NIL also set for zero-weight terms)
};

// ---------------------------------------------------------------------------------------------
// Lockstep sweep.  One workgroup of TWO waves per job; a job is 64/G lattices, G lanes each,
// walking the forward (dir 0) or backward (dir 1) arc stream of those lattices round by round.
//
//   wave 0, PRODUCER  reads the arc stream -- the only HBM stream: 12 B per term, read once, as
//                     1 KiB dwordx4 wave loads, prefetched kPrefetchChunks chunks (16 rounds) ahead
//                     in registers -- and evaluates, per term, the cost (ScaleLattice / insertion
//                     penalty fused) and the 2^k * wm split of exp(-cost).  None of this depends
//                     on state values.  It leaves {wm, k, word} of the next chunk (4 rounds) in an
//                     LDS stage buffer.
//   wave 1, CONSUMER  owns the serial dependency chain: value of the source state -> multiply ->
//                     ldexp -> add into the lane's accumulator -> (G lanes combined with DPP) ->
//                     frexp; the result is forwarded IN REGISTERS to the next round.
// The two waves sit on different SIMDs of one CU and meet at one workgroup barrier per chunk
// (double-buffered stage), so the chain wave issues only the chain.
//
// The values of the last W visited states of each lattice live in an LDS ring of 16-byte cells
// ([slot][lattice]).  The ring read of round r+1 is issued before round r's result is written
// and is patched from registers when it names the state round r just finished, so the LDS
// write->read latency is off the chain.  Chunks in which some lane has a constant-one source
// (start state / final weight) or a source older than W states (FAR: re-read from the cells
// this kernel itself wrote) take a slower instantiation of the same round.
//
// Two instantiations share this body:
//   FAST   every term of a state is aligned to ONE reference exponent -- that of the lattice's
//          previous non-zero state -- so a term costs multiply + ldexp + add and the G lanes
//          combine with a plain sum.  A term whose exponent falls outside +-900 of the reference
//          (arc costs hundreds of nats apart inside one state) sets the job's flag.
//   !FAST  exact for any input: every lane carries its own (mantissa, exponent) accumulator and
//          re-references on demand.  It runs afterwards and only for flagged jobs.
template <int G, bool FAST>
__global__ __launch_bounds__(128) void sweep_kernel(SweepArgs args) {
  extern __shared__ MECell smem[];
  constexpr int LPW = 64 / G;
  constexpr int D = kPrefetchChunks;
  const int W = args.ring_slots;
  if (!FAST && args.job_flags[blockIdx.x] == 0) return;  // workgroup-uniform

  MECell* ring = smem;                                                  // [W][LPW]
  StagedTerm* stage = reinterpret_cast<StagedTerm*>(smem + (size_t)W * LPW);  // [2][4][64]
  const int lane = threadIdx.x & 63;
  const bool producer = threadIdx.x < 64;  // wave 0
  const SweepJob job = args.jobs[blockIdx.x];
  const int nchunks = job.nchunks;

  if (producer) {
    // ------------------------------------------------------------------ wave 0: weights
    const ScaleParams sp = args.sp;
    const int4* __restrict__ sw = reinterpret_cast<const int4*>(args.sw) + job.chunk_off * 64 + lane;
    const float4* __restrict__ sg = reinterpret_cast<const float4*>(args.sg) + job.chunk_off * 64 + lane;
    const float4* __restrict__ sa = reinterpret_cast<const float4*>(args.sa) + job.chunk_off * 64 + lane;
    int4 wb[D];
    float4 gb[D], ab[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int64_t cn = min(j, nchunks - 1);
      wb[j] = sw[cn * 64];
      gb[j] = sg[cn * 64];
      ab[j] = sa[cn * 64];
    }
    auto put = [&](StagedTerm* dst, uint32_t w, float g, float a) {
      StagedTerm t;
      bool dead;
      arc_weight(arc_cost(g, a, (w & kWord) != 0, sp), (w & kNil) != 0, t.wm, t.k, dead);
      t.w = dead ? (w | kNil) : w;
      *dst = t;
    };
    // chunk c is staged during the consumer's chunk c-1; nchunks is a multiple of D
    for (int c = 0; c < nchunks; c += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int4 w4 = wb[j];
        const float4 g4 = gb[j];
        const float4 a4 = ab[j];
        const int64_t cn = min(c + j + D, nchunks - 1);
        wb[j] = sw[cn * 64];
        gb[j] = sg[cn * 64];
        ab[j] = sa[cn * 64];
        StagedTerm* st = stage + ((c + j) & 1) * 256 + lane;
        put(st, (uint32_t)w4.x, g4.x, a4.x);
        put(st + 64, (uint32_t)w4.y, g4.y, a4.y);
        put(st + 128, (uint32_t)w4.z, g4.z, a4.z);
        put(st + 192, (uint32_t)w4.w, g4.w, a4.w);
        __syncthreads();  // chunk c+j staged; the consumer has finished chunk c+j-1
      }
    }
    return;
  }

  // -------------------------------------------------------------------- wave 1: the chain
  const int slot = lane / G;
  const int sub = lane % G;
  const LaneInfo li = args.lanes[job.lane_tab + slot];
  const int S = li.num_states;
  const bool bwd = job.dir != 0;
  MECell* cells = bwd ? args.bme : args.fme;
  // position p of the stream -> cell: forward walks the lattice's cells upwards (position S,
  // the virtual super-final state, is the extra cell), backward downwards from state S-1
  const int64_t cbase = li.state_base + li.lattice;  // state_off[u] + u
  MECell* out = cells + (li.state_base < 0 ? 0 : (bwd ? cbase + S - 1 : cbase));
  const int ostep = bwd ? -1 : 1;

  // every ring cell starts as log-zero so that padding terms read something well-defined
  for (int i = lane; i < W * LPW; i += 64) {
    ring[i].m = 0.0;
    ring[i].e = kExpMin;
  }

  int p = 0;           // position of the state this lattice is currently accumulating
  double accM = 0.0;   // lane accumulator: mantissa scaled by 2^accE (!FAST) or 2^refE (FAST)
  int accE = kExpMin;
  int refE = 0;        // FAST: exponent of the lattice's previous non-zero state
  bool sticky = false; // FAST: some term left the safe exponent window
  double lastM = 0.0;  // value of the state finished by the previous round (register bypass)
  int lastE = kExpMin;
  bool prevLast = false;

  // ring read for the upcoming round, software-pipelined one round ahead
  MECell pre;
  auto preload = [&](uint32_t wnext, int pnext) {
    const int rs = (pnext - (int)(wnext & kRelMask)) & (W - 1);
    pre = ring[rs * LPW + slot];
  };

  // SPECIAL = this chunk may hold constant-one or FAR sources; PRE = issue the next round's ring
  // read (the last round of a chunk cannot: its successor is staged behind the barrier)
  auto round = [&](auto special, auto do_pre, uint32_t w, double wm, int k, uint32_t wnext) {
    const int rel = (int)(w & kRelMask);
    const bool last = (w & kLast) != 0;
    const bool first = (w & kFirst) != 0;
    double sm = pre.m;
    int se = (int)pre.e;
    // issue next round's ring read now, ahead of this round's ring write
    if (decltype(do_pre)::value) preload(wnext, p + (last ? 1 : 0));
    const bool bypass = prevLast & (rel == 1);
    sm = bypass ? lastM : sm;
    se = bypass ? lastE : se;
    if (decltype(special)::value) {
      const bool one = (rel == 0) & ((w & kNil) == 0);  // constant one = 0.5 * 2^1
      sm = one ? 0.5 : sm;
      se = one ? 1 : se;
      if (w & kFar) {
        const int q = p - rel;
        const MECell* src = cells + cbase + (bwd ? (int64_t)(S - 1 - q) : (int64_t)q);
        sm = __hip_atomic_load(&src->m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        se = (int)__hip_atomic_load(&src->e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // dead (zero weight / padding) terms have wm == 0 and zero sources sm == 0: tm is then 0
    const double tm = sm * wm;
    double cM;
    int cE;
    if (FAST) {
      const int shift = se + k - refE;
      sticky |= (tm != 0.0) & ((unsigned)(shift + 900) > 1800u);
      const double add = ldexp(tm, shift);
      accM = first ? add : accM + add;
      cM = G > 1 ? group_sum<G>(accM) : accM;
      cE = refE;
    } else {
      // only the exponent tag needs forcing so that a zero never becomes a reference exponent
      const int te = (((w & kNil) != 0) | (sm == 0.0)) ? kExpMin : se + k;
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
    const bool nz = cM != 0.0;
    const double m = __builtin_amdgcn_frexp_mant(cM);
    const int e = nz ? cE + __builtin_amdgcn_frexp_exp(cM) : kExpMin;
    lastM = m;
    lastE = e;
    prevLast = last;
    if (last && sub == 0) {
      MECell cell;
      cell.m = m;
      cell.e = e;
      ring[(p & (W - 1)) * LPW + slot] = cell;
      *out = cell;
    }
    p += last ? 1 : 0;
    out += last ? ostep : 0;
    if (FAST) refE = (last & nz) ? e : refE;
  };

  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();  // chunk c is staged (and, for c == 0, the ring is initialised)
    const StagedTerm* st = stage + (c & 1) * 256 + lane;
    const StagedTerm t0 = st[0], t1 = st[64], t2 = st[128], t3 = st[192];
    preload(t0.w, p);  // after the previous chunk's last ring write: reads the fresh value
    constexpr uint32_t kOneMask = kRelMask | kNil;  // rel == 0 on a real term = constant one
    const bool sp0 = ((t0.w & kOneMask) == 0) | ((t0.w & kFar) != 0), sp1 = ((t1.w & kOneMask) == 0) | ((t1.w & kFar) != 0);
    const bool sp2 = ((t2.w & kOneMask) == 0) | ((t2.w & kFar) != 0), sp3 = ((t3.w & kOneMask) == 0) | ((t3.w & kFar) != 0);
    const std::true_type yes;
    const std::false_type no;
    if (__any(sp0 | sp1 | sp2 | sp3)) {
      round(yes, yes, t0.w, t0.wm, t0.k, t1.w);
      round(yes, yes, t1.w, t1.wm, t1.k, t2.w);
      round(yes, yes, t2.w, t2.wm, t2.k, t3.w);
      round(yes, no, t3.w, t3.wm, t3.k, 0u);
    } else {
      round(no, yes, t0.w, t0.wm, t0.k, t1.w);
      round(no, yes, t1.w, t1.wm, t1.k, t2.w);
      round(no, yes, t2.w, t2.wm, t2.k, t3.w);
      round(no, no, t3.w, t3.wm, t3.k, 0u);
    }
  }
  if (FAST) {
    if (__any(sticky) && lane == 0) args.job_flags[blockIdx.x] = 1;
  }
}

template <int G>
static void launch_pair(const SweepArgs& args, int num_jobs, size_t lds, hipStream_t stream) {
  hipLaunchKernelGGL((sweep_kernel<G, true>), dim3(num_jobs), dim3(128), lds, stream, args);
  hipLaunchKernelGGL((sweep_kernel<G, false>), dim3(num_jobs), dim3(128), lds, stream, args);
}

// Clears the job flags, runs the fast sweep over every job and the exact sweep over the jobs the
// fast one flagged (normally none: those workgroups exit at once).
hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream) {
  if (num_jobs <= 0) return hipSuccess;
  const int LPW = 64 / G;
  const size_t lds = (size_t)args.ring_slots * LPW * sizeof(MECell) + 2 * 4 * 64 * sizeof(StagedTerm);
  hipError_t e = hipMemsetAsync(args.job_flags, 0, sizeof(int32_t) * (size_t)num_jobs, stream);
  if (e != hipSuccess) return e;
  switch (G) {
    case 1: launch_pair<1>(args, num_jobs, lds, stream); break;
    case 2: launch_pair<2>(args, num_jobs, lds, stream); break;
    case 4: launch_pair<4>(args, num_jobs, lds, stream); break;
    case 8: launch_pair<8>(args, num_jobs, lds, stream); break;
    case 16: launch_pair<16>(args, num_jobs, lds, stream); break;
    case 32: launch_pair<32>(args, num_jobs, lds, stream); break;
    case 64: launch_pair<64>(args, num_jobs, lds, stream); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace klat
