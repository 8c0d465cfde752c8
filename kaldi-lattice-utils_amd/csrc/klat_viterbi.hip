// klat_viterbi.hip -- the beam decisions of the tropical twin: from the Viterbi
// forward and backward costs -- computed by the sweep kernel's TROPICAL instantiation over the same
// lockstep arc streams (klat_sweep.hip, launch_sweep_tropical) -- ComputeLatticeBeam
// (latbin/lattice-prune-dyn-beam.cc:27-90) and the pruning decisions of PruneLattice(beam, &clat)
// [upstream Kaldi; call sites kwsbin2/lattice-word-index-segment.cc:49-51,
// kwsbin2/lattice-word-index-position.cc:49-51, latbin/lattice-prune-dyn-beam.cc:175].
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_kernels.h"
#include "klat_math.h"

namespace klat {

namespace {

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

hipError_t launch_prune(const PruneArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(prune_kernel, dim3(num_lattices), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace klat
