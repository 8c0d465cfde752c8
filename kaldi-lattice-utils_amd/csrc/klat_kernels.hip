// klat_kernels.hip -- the fully parallel kernels around the sweep: per-lattice totals and the
// per-arc log posteriors (a8 of SURVEY.md 8a: kwsbin2/lattice-word-index-segment.cc:160-163,
// latbin/lattice-to-word-frame-post.cc:102-104) fused with the natural-log alpha/beta outputs
// of the B2 interface.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_kernels.h"
#include "klat_math.h"

namespace klat {

// Per-lattice total = 0.5*(tot_forward + tot_backward) (Appendix A.1 return value).
// An empty lattice (Start()==kNoStateId) gets total 0, as ProcessLattice sets it
// (kwsbin2/lattice-word-index-segment.cc:64-69).
__global__ __launch_bounds__(256) void totals_kernel(TotalsArgs a) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= a.num_lattices) return;
  const int64_t s0 = a.state_off[u];
  const int64_t S = a.state_off[u + 1] - s0;
  double tf = 0.0, tb = 0.0;
  if (S > 0) {
    const MECell f = a.fme[s0 + u + S];  // virtual super-final state of the forward sweep
    const MECell b = a.bme[s0 + u];      // beta[start]
    tf = me_to_ln(f.m, (int)f.e);
    tb = me_to_ln(b.m, (int)b.e);
  }
  a.tot_fwd[u] = tf;
  a.tot_bwd[u] = tb;
  a.total[u] = 0.5 * (tb + tf);
}

// a8 + normalisation: logpost[i] = alpha[src] - cost + beta[dst] - total, fully parallel over
// arcs in caller order.  Reads src,dst,graph,acoustic (16 B/arc, coalesced), gathers the 16-byte
// cells of the two end states (L2-resident: arcs are grouped by source), writes 8 B/arc; each
// work item also turns a slice of its lattice's cells into natural-log alpha/beta
// (vector<double>* alpha, *beta of the B2 interface), 16 B/state written.
template <bool PEN>
__global__ __launch_bounds__(256) void posterior_kernel(PosteriorArgs a) {
  const PostItem item = a.items[blockIdx.x];
  const int64_t base = a.state_off[item.lattice];
  const MECell* __restrict__ fme = a.fme + base + item.lattice;
  const MECell* __restrict__ bme = a.bme + base + item.lattice;
  const double tot = a.total[item.lattice];
  const ScaleParams sp = a.sp;
  for (int t = threadIdx.x; t < item.count; t += 256) {
    const int64_t i = item.arc_begin + t;
    const int s = a.src[i];
    const int d = a.dst[i];
    const bool word = PEN ? (a.ilabel[i] != 0) : false;
    const double c = arc_cost(a.graph[i], a.acoustic[i], word, sp);
    const MECell f = fme[s];
    const MECell b = bme[d];
    const double prod = f.m * b.m;
    const int e = (int)f.e + (int)b.e;
    double r = -INFINITY;
    if (prod != 0.0 && c < 1.0e30) r = fma((double)e, kLn2, log(prod)) - c - tot;
    a.logpost[i] = r;
  }
  for (int t = threadIdx.x; t < item.state_count; t += 256) {
    const int s = item.state_begin + t;
    const MECell f = fme[s];
    const MECell b = bme[s];
    a.alpha[base + s] = me_to_ln(f.m, (int)f.e);
    a.beta[base + s] = me_to_ln(b.m, (int)b.e);
  }
}

hipError_t launch_totals(const TotalsArgs& a, hipStream_t stream) {
  if (a.num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(totals_kernel, dim3((a.num_lattices + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_posterior(const PosteriorArgs& a, int num_items, hipStream_t stream) {
  if (num_items <= 0) return hipSuccess;
  if (a.sp.penalized)
    hipLaunchKernelGGL(posterior_kernel<true>, dim3(num_items), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(posterior_kernel<false>, dim3(num_items), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace klat
