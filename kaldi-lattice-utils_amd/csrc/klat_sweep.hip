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

// ---------------------------------------------------------------------------------------------
// Lockstep sweep.  One wave per job; a job is 64/G lattices, G lanes each, walking the forward
// (dir 0) or backward (dir 1) arc stream of those lattices round by round.  The arc stream is
// the only HBM stream: 12 B per term, read once, as 1 KiB dwordx4 wave loads, prefetched
// kPrefetchChunks chunks (16 rounds) ahead in registers.
//
// What sits on the serial dependency chain of a lattice is kept minimal:
//   * phase A (per chunk of 4 rounds, independent of state values): cost and the 2^k * wm split
//     of exp(-cost) -- four independent instruction streams the scheduler interleaves;
//   * phase B (per round, straight-line): value of the source state -> multiply -> ldexp ->
//     add into the lane's accumulator -> (G lanes combined with DPP) -> frexp; the result is
//     forwarded IN REGISTERS to the next round.
// The values of the last W visited states of each lattice live in an LDS ring of 16-byte cells
// ([slot][lattice]).  The ring read of round r+1 is issued before round r's result is written
// and is patched from registers when it names the state round r just finished, so the LDS
// write->read latency is off the chain.  Chunks in which some lane has a constant-one source
// (start state / final weight) or a source older than W states (FAR: re-read from the cells
// this kernel itself wrote) take a slower instantiation of the same round.
template <int G>
__global__ __launch_bounds__(64) void sweep_kernel(SweepArgs args) {
  extern __shared__ MECell ring[];  // [W][LPW]
  constexpr int LPW = 64 / G;
  constexpr int D = kPrefetchChunks;
  const int W = args.ring_slots;

  const int lane = threadIdx.x;
  const int slot = lane / G;
  const int sub = lane % G;
  const SweepJob job = args.jobs[blockIdx.x];
  const LaneInfo li = args.lanes[job.lane_tab + slot];
  const int S = li.num_states;
  const bool bwd = job.dir != 0;
  MECell* cells = bwd ? args.bme : args.fme;
  // position p of the stream -> cell: forward walks the lattice's cells upwards (position S,
  // the virtual super-final state, is the extra cell), backward downwards from state S-1
  const int64_t cbase = li.state_base + li.lattice;  // state_off[u] + u
  MECell* out = cells + (li.state_base < 0 ? 0 : (bwd ? cbase + S - 1 : cbase));
  const int ostep = bwd ? -1 : 1;
  const ScaleParams sp = args.sp;

  const int4* __restrict__ sw = reinterpret_cast<const int4*>(args.sw) + job.chunk_off * 64 + lane;
  const float4* __restrict__ sg = reinterpret_cast<const float4*>(args.sg) + job.chunk_off * 64 + lane;
  const float4* __restrict__ sa = reinterpret_cast<const float4*>(args.sa) + job.chunk_off * 64 + lane;

  // every ring cell starts as log-zero so that padding terms read something well-defined
  for (int i = lane; i < W * LPW; i += 64) {
    ring[i].m = 0.0;
    ring[i].e = kExpMin;
  }
  __builtin_amdgcn_wave_barrier();

  int p = 0;           // position of the state this lattice is currently accumulating
  double accM = 0.0;   // lane accumulator: mantissa scaled by 2^accE
  int accE = kExpMin;
  double lastM = 0.0;  // value of the state finished by the previous round (register bypass)
  int lastE = kExpMin;
  bool prevLast = false;

  int4 wb[D];
  float4 gb[D], ab[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const int64_t cn = min(j, job.nchunks - 1);
    wb[j] = sw[cn * 64];
    gb[j] = sg[cn * 64];
    ab[j] = sa[cn * 64];
  }

  // ring read for the upcoming round, software-pipelined one round ahead
  MECell pre;
  auto preload = [&](uint32_t wnext, int pnext) {
    const int rs = (pnext - (int)(wnext & kRelMask)) & (W - 1);
    pre = ring[rs * LPW + slot];
  };
  preload((uint32_t)wb[0].x, 0);

  // SPECIAL = this chunk may hold constant-one or FAR sources
  auto round = [&](auto special, uint32_t w, double wm, int k, bool dead, uint32_t wnext) {
    const int rel = (int)(w & kRelMask);
    const bool last = (w & kLast) != 0;
    double sm = pre.m;
    int se = (int)pre.e;
    // issue next round's ring read now, ahead of this round's ring write
    preload(wnext, p + (last ? 1 : 0));
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
    // dead (zero weight / padding) terms have wm == 0, so tm is already 0; only the exponent
    // tag needs forcing so that a zero never becomes a lane's reference exponent
    const double tm = sm * wm;
    const int te = (dead | (sm == 0.0)) ? kExpMin : se + k;
    const bool first = (w & kFirst) != 0;
    const int ref = first ? te : accE;
    const int shift = te - ref;
    if (__any(shift > 900)) {
      // rare: a later term dwarfs the lane's reference exponent -> re-reference (robust form)
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
    // close the state: computed every round (straight-line), committed when LAST
    double cM = accM;
    int cE = accE;
    if (G > 1) {
      cE = group_max<G>(accE);
      cM = group_sum<G>(ldexp(accM, accE - cE));
    }
    const double m = __builtin_amdgcn_frexp_mant(cM);
    const int e = (cM == 0.0) ? kExpMin : cE + __builtin_amdgcn_frexp_exp(cM);
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
  };

  for (int c = 0; c < job.nchunks; c += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int4 w4 = wb[j];
      const float4 g4 = gb[j];
      const float4 a4 = ab[j];
      const int64_t cn = min(c + j + D, job.nchunks - 1);
      wb[j] = sw[cn * 64];
      gb[j] = sg[cn * 64];
      ab[j] = sa[cn * 64];
      const uint32_t wn = (uint32_t)wb[(j + 1) % D].x;  // first word of the next chunk
      const uint32_t x = (uint32_t)w4.x, y = (uint32_t)w4.y, z = (uint32_t)w4.z, v = (uint32_t)w4.w;
      // phase A: four independent cost -> weight evaluations (padding terms get weight zero)
      double wm0, wm1, wm2, wm3;
      int k0, k1, k2, k3;
      bool d0, d1, d2, d3;
      arc_weight(arc_cost(g4.x, a4.x, (x & kWord) != 0, sp), (x & kNil) != 0, wm0, k0, d0);
      arc_weight(arc_cost(g4.y, a4.y, (y & kWord) != 0, sp), (y & kNil) != 0, wm1, k1, d1);
      arc_weight(arc_cost(g4.z, a4.z, (z & kWord) != 0, sp), (z & kNil) != 0, wm2, k2, d2);
      arc_weight(arc_cost(g4.w, a4.w, (v & kWord) != 0, sp), (v & kNil) != 0, wm3, k3, d3);
      // pin phase A here: without this the compiler sinks each evaluation into the round that
      // uses it, back onto the dependency chain
      asm volatile("" : "+v"(wm0), "+v"(wm1), "+v"(wm2), "+v"(wm3), "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
      // does any lane of this chunk need the constant-one / FAR handling?
      constexpr uint32_t kOneMask = kRelMask | kNil;  // rel == 0 on a real term = constant one
      const bool sp0 = ((x & kOneMask) == 0) | ((x & kFar) != 0), sp1 = ((y & kOneMask) == 0) | ((y & kFar) != 0);
      const bool sp2 = ((z & kOneMask) == 0) | ((z & kFar) != 0), sp3 = ((v & kOneMask) == 0) | ((v & kFar) != 0);
      // phase B: the chain
      if (__any(sp0 | sp1 | sp2 | sp3)) {
        const std::true_type t;
        round(t, x, wm0, k0, d0, y);
        round(t, y, wm1, k1, d1, z);
        round(t, z, wm2, k2, d2, v);
        round(t, v, wm3, k3, d3, wn);
      } else {
        const std::false_type f;
        round(f, x, wm0, k0, d0, y);
        round(f, y, wm1, k1, d1, z);
        round(f, z, wm2, k2, d2, v);
        round(f, v, wm3, k3, d3, wn);
      }
    }
  }
}

hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream) {
  if (num_jobs <= 0) return hipSuccess;
  const int LPW = 64 / G;
  const size_t lds = (size_t)args.ring_slots * LPW * sizeof(MECell);
  dim3 grid(num_jobs), block(64);
  switch (G) {
    case 1: hipLaunchKernelGGL(sweep_kernel<1>, grid, block, lds, stream, args); break;
    case 2: hipLaunchKernelGGL(sweep_kernel<2>, grid, block, lds, stream, args); break;
    case 4: hipLaunchKernelGGL(sweep_kernel<4>, grid, block, lds, stream, args); break;
    case 8: hipLaunchKernelGGL(sweep_kernel<8>, grid, block, lds, stream, args); break;
    case 16: hipLaunchKernelGGL(sweep_kernel<16>, grid, block, lds, stream, args); break;
    case 32: hipLaunchKernelGGL(sweep_kernel<32>, grid, block, lds, stream, args); break;
    case 64: hipLaunchKernelGGL(sweep_kernel<64>, grid, block, lds, stream, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace klat
