// klat_kernels.hip -- the fully parallel kernels around the sweep: per-lattice totals and the
// per-arc log posteriors (a8 of SURVEY.md 8a: kwsbin2/lattice-word-index-segment.cc:160-163,
// latbin/lattice-to-word-frame-post.cc:102-104).  The sweep itself leaves alpha and beta in
// natural log, so this stage is a pure streaming pass.
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
    const int64_t base = padded_state_base(s0, u);
    tf = a.alpha[base + S];  // value of the forward sweep's virtual super-final state
    tb = a.beta[base];       // beta[start]
  }
  a.tot_fwd[u] = tf;
  a.tot_bwd[u] = tb;
  a.total[u] = 0.5 * (tb + tf);
}

// a8 + normalisation: logpost[i] = alpha[src] - cost + beta[dst] - total, fully parallel over
// arcs in caller order.  Reads src,dst,graph,acoustic (16 B/arc, coalesced), gathers alpha[src]
// and beta[dst] (8 B each, L2-resident: arcs are grouped by source), writes 8 B/arc.
template <bool PEN>
__global__ __launch_bounds__(256) void posterior_kernel(PosteriorArgs a) {
  const PostItem item = a.items[blockIdx.x];
  const int64_t s0 = a.state_off[item.lattice];
  const int64_t S = a.state_off[item.lattice + 1] - s0;
  const int64_t base = padded_state_base(s0, item.lattice);
  const double* __restrict__ alpha = a.alpha + base;
  const double* __restrict__ beta = a.beta + base;
  // total = (tot_forward + beta[start]) / 2, 0 for an empty lattice (see totals_kernel: this kernel
  // took its work over, one launch less between the sweep and the posteriors); the lattice's first
  // item leaves the three numbers for klat_get_totals and the accumulators
  const double tf = S > 0 ? alpha[S] : 0.0, tb = S > 0 ? beta[0] : 0.0;
  const double tot = 0.5 * (tb + tf);
  if ((item.count & kPostItemFirst) && threadIdx.x == 0) {
    a.tot_fwd[item.lattice] = tf;
    a.tot_bwd[item.lattice] = tb;
    a.total[item.lattice] = tot;
  }
  const int count = item.count & (kPostItemFirst - 1);
  const ScaleParams sp = a.sp;
  for (int t = threadIdx.x; t < count; t += 256) {
    const int64_t i = item.arc_begin + t;
    const int s = a.src[i];
    const int d = a.dst[i];
    const bool word = PEN ? (a.ilabel[i] != 0) : false;
    const double c = arc_cost(a.graph[i], a.acoustic[i], word, sp);
    const double through = alpha[s] + beta[d] - c;
    // log-zero stays log-zero (also when the lattice has no path at all: total == -inf)
    a.logpost[i] = (through > -INFINITY) ? through - tot : -INFINITY;
  }
}

// one workgroup per lattice: dense[state_off[u] + s] = padded[state_off[u] + u + s]
__global__ __launch_bounds__(256) void unpad_kernel(UnpadArgs a) {
  for (int u = blockIdx.x; u < a.num_lattices; u += gridDim.x) {
    const int64_t s0 = a.state_off[u];
    const int64_t S = a.state_off[u + 1] - s0;
    const double* __restrict__ src = a.padded + padded_state_base(s0, u);
    for (int64_t s = threadIdx.x; s < S; s += 256) a.dense[s0 + s] = src[s];
  }
}

hipError_t launch_unpad(const UnpadArgs& a, hipStream_t stream) {
  if (a.num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(unpad_kernel, dim3(a.num_lattices < 65535 ? a.num_lattices : 65535), dim3(256), 0, stream, a);
  return hipGetLastError();
}

void fill_ln_table(LnTabEntry* t) {
  for (int i = 0; i < kLnTabSize; ++i) {
    const double c = 1.0 + ((double)i + 0.5) / (double)kLnTabSize;
    t[i].inv2 = 2.0 / c;
    t[i].lnc = log(0.5 * c);
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
