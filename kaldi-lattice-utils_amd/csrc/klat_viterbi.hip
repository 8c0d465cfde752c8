// klat_viterbi.hip -- the tropical twin of the sweep: Viterbi forward and backward
// costs over the same lockstep arc streams, and from them ComputeLatticeBeam
// (latbin/lattice-prune-dyn-beam.cc:27-90) and the pruning decisions of PruneLattice(beam, &clat)
// [upstream Kaldi; call sites kwsbin2/lattice-word-index-segment.cc:49-51,
// kwsbin2/lattice-word-index-position.cc:49-51, latbin/lattice-prune-dyn-beam.cc:175].
//
// (min,+) needs no exp, no exponents and no producer: one wave per job walks its stream alone --
// cost of the term (ScaleLattice / insertion penalty fused as in the sweep), source value from the
// LDS ring (or 0 for the start / a final weight, or from registers when it is the state finished
// one round earlier), min over the lattice's G lanes with DPP, ring cell, and the state's cost to
// HBM when it is complete.  A source older than the ring is this wave's own earlier store.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_kernels.h"
#include "klat_math.h"

namespace klat {

namespace {

template <int CTRL>
__device__ __forceinline__ double vdpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ double group_min(double v) {
  if (G >= 2) v = fmin(v, vdpp<0xB1>(v));    // quad_perm [1,0,3,2]
  if (G >= 4) v = fmin(v, vdpp<0x4E>(v));    // quad_perm [2,3,0,1]
  if (G >= 8) v = fmin(v, vdpp<0x141>(v));   // row_half_mirror
  if (G >= 16) v = fmin(v, vdpp<0x140>(v));  // row_mirror
  if (G >= 32) v = fmin(v, __shfl_xor(v, 16, 64));
  if (G >= 64) v = fmin(v, __shfl_xor(v, 32, 64));
  return v;
}

template <int G>
__global__ __launch_bounds__(64) void viterbi_kernel(ViterbiArgs a) {
  extern __shared__ double ring[];  // [W][LPW] + 64 dummy cells
  constexpr int LPW = 64 / G;
  const int W = a.ring_slots;
  const int lane = threadIdx.x;
  const int slot = lane / G, sub = lane % G;
  const SweepJob job = a.jobs[blockIdx.x];
  const LaneInfo li = a.lanes[job.lane_tab + slot];
  const int S = li.num_states;
  const bool bwd = job.dir != 0;
  const bool live = li.state_base >= 0;
  double* vals = (bwd ? a.bwd_cost : a.fwd_cost) + padded_state_base(li.state_base, li.lattice);
  const int q_off = bwd ? S - 1 : 0, q_step = bwd ? -1 : 1;
  for (int i = lane; i < W * LPW + 64; i += 64) ring[i] = INFINITY;
  __syncthreads();
  const int4* sw = reinterpret_cast<const int4*>(a.sw) + job.chunk_off * 64 + lane;
  const float4* sg = reinterpret_cast<const float4*>(a.sg) + job.chunk_off * 64 + lane;
  const float4* sa = reinterpret_cast<const float4*>(a.sa) + job.chunk_off * 64 + lane;
  const ScaleParams sp = a.sp;
  const StreamCost sc = a.sc;
  int p = 0;            // position the lattice is accumulating
  double acc = INFINITY, lastv = INFINITY;
  int4 w4 = sw[0];
  float4 g4 = sg[0], a4 = sa[0];
  for (int c = 0; c < job.nchunks; ++c) {
    const int4 wc = w4;
    const float4 gc = g4, ac = a4;
    const int64_t cn = min(c + 1, job.nchunks - 1);
    w4 = sw[cn * 64];
    g4 = sg[cn * 64];
    a4 = sa[cn * 64];
    const uint32_t ws[4] = {(uint32_t)wc.x, (uint32_t)wc.y, (uint32_t)wc.z, (uint32_t)wc.w};
    const float gs[4] = {gc.x, gc.y, gc.z, gc.w}, as[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t w = ws[r];
      double cost = sc.general ? arc_cost(gs[r], as[r], (w & kWord) != 0, sp) : stream_cost(w, gs[r], as[r], sc);
      cost = (cost < 1.0e30) ? cost : INFINITY;  // Zero weights, NaN, padding
      const bool last = (w & kLast) != 0;
      double src;
      int cell;
      bool commit;
      if (w & kFar) {  // the packer left the cells of terms older than the ring to the kernel
        const int q = p - (int)(w & kRelMask);
        src = live ? __hip_atomic_load(&vals[q_off + q_step * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INFINITY;
        commit = last && sub == 0;
        cell = commit ? (p & (W - 1)) * LPW + slot : W * LPW + lane;
      } else {
        src = ring[w & kCellMask];
        if (w & kBypass) src = lastv;
        if (w & kOne) src = 0.0;
        commit = (w & kCommit) != 0;
        cell = (int)((w >> kWrShift) & kCellMask);
      }
      const double t = src + cost;
      acc = (w & kFirst) ? t : fmin(acc, t);
      const double v = G > 1 ? group_min<G>(acc) : acc;
      lastv = v;
      ring[cell] = v;
      if (commit && live) vals[q_off + q_step * p] = v;
      p += last ? 1 : 0;
    }
  }
}

// ComputeLatticeBeam and the PruneLattice decisions of one lattice per workgroup: arcs whose best
// path costs more than best + beam are dropped (keep_arc = 0), so are such final weights;
// lattice_beam = the largest (best path through an arc or final) - best, i.e. the beam at which
// nothing would be pruned (latbin/lattice-prune-dyn-beam.cc:64-88); as there, an arc that lies on no
// complete path makes it +inf.
__global__ __launch_bounds__(256) void prune_kernel(PruneArgs a) {
  __shared__ double red[256];
  const int u = blockIdx.x;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0);
  const int A = (int)(a.arc_off[u + 1] - a0);
  if (S == 0) {
    if (threadIdx.x == 0) a.lattice_beam[u] = 0.0;
    return;
  }
  const double* fv = a.fwd_cost + padded_state_base(s0, u);
  const double* bv = a.bwd_cost + padded_state_base(s0, u);
  const double best = fv[S];  // the forward pass's virtual super-final state: best_final_cost
  const double cutoff = best + a.beam;
  double worst = best;
  for (int i = threadIdx.x; i < A; i += 256) {
    const double c = arc_cost(a.graph[a0 + i], a.acoustic[a0 + i], a.ilabel[a0 + i] != 0, a.sp);
    const double through = fv[a.src[a0 + i]] + (c + bv[a.dst[a0 + i]]);
    if (a.keep_arc) a.keep_arc[a0 + i] = (through > cutoff) ? 0 : 1;
    if (through > worst) worst = through;
  }
  for (int s = threadIdx.x; s < S; s += 256) {
    const double fc = arc_cost(a.final_graph[s0 + s], a.final_acoustic[s0 + s], false, a.sp);
    const bool is_final = fc != INFINITY;
    const double through = fv[s] + fc;
    if (a.keep_final) a.keep_final[s0 + s] = (is_final && !(through > cutoff)) ? 1 : 0;
    if (is_final && through > worst) worst = through;
  }
  red[threadIdx.x] = worst;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + d]);
    __syncthreads();
  }
  if (threadIdx.x == 0) a.lattice_beam[u] = red[0] - best;
}

}  // namespace

hipError_t launch_viterbi(const ViterbiArgs& a, int num_jobs, int G, hipStream_t stream) {
  if (num_jobs <= 0) return hipSuccess;
  const size_t lds = ((size_t)a.ring_slots * (64 / G) + 64) * sizeof(double);
  switch (G) {
    case 1: hipLaunchKernelGGL(viterbi_kernel<1>, dim3(num_jobs), dim3(64), lds, stream, a); break;
    case 2: hipLaunchKernelGGL(viterbi_kernel<2>, dim3(num_jobs), dim3(64), lds, stream, a); break;
    case 4: hipLaunchKernelGGL(viterbi_kernel<4>, dim3(num_jobs), dim3(64), lds, stream, a); break;
    case 8: hipLaunchKernelGGL(viterbi_kernel<8>, dim3(num_jobs), dim3(64), lds, stream, a); break;
    case 16: hipLaunchKernelGGL(viterbi_kernel<16>, dim3(num_jobs), dim3(64), lds, stream, a); break;
    case 32: hipLaunchKernelGGL(viterbi_kernel<32>, dim3(num_jobs), dim3(64), lds, stream, a); break;
    case 64: hipLaunchKernelGGL(viterbi_kernel<64>, dim3(num_jobs), dim3(64), lds, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_prune(const PruneArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(prune_kernel, dim3(num_lattices), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace klat
