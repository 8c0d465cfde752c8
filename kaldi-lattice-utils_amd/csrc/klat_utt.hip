// klat_utt.hip -- utterance index: log P(word w occurs somewhere on the path | lattice), the score
// of kwsbin2/lattice-word-index-utterance.cc (known answer kwsbin2/README.md:25).
//
// The reference composes the lattice with the automaton Sigma* w Sigma* once per word (:171) and
// runs a backward pass over the product (:174-177) -- one task per (lattice, word).  The product's
// states are (s, seen-w?); the "seen" layer is the lattice's own beta, the other layer obeys
//     b0[s] = LogAdd over the arcs of s of  (olabel == w ? beta[d] : b0[d]) - cost,
// so a query is one more backward pass over the same arcs.  Here one LANE owns one word: a wave
// walks the states of its lattice backwards once for 64 words at a time; arc data are the same for
// all lanes (one broadcast load), the per-word layer b0 lives in a [state][lane] scratch row block
// (512-byte coalesced rows, L2-resident), beta comes from the forward-backward already on the
// device.  Workgroups of one wave pull (lattice, 64 words) items from a queue until it is empty.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_accum.h"
#include "klat_math.h"

namespace klat {

namespace {

__device__ __forceinline__ double utt_log_add(double x, double y) {
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  if (diff >= -36.04365338911715) return x + log1p(exp(diff));  // log(DBL_EPSILON)
  return x;  // also when both are -inf (diff is NaN)
}

__global__ __launch_bounds__(64) void utterance_kernel(UttArgs a) {
  const int lane = threadIdx.x;
  double* __restrict__ b0 = a.scratch + (size_t)blockIdx.x * a.scratch_states * 64;
  const ScaleParams sp = a.sp;
  for (;;) {
    int it = 0;
    if (lane == 0) it = atomicAdd(a.next_item, 1);
    it = __shfl(it, 0, 64);
    if (it >= a.num_items) break;  // every wave reaches this: the queue only grows
    const UttItem item = a.items[it];
    const int u = item.lattice;
    const int64_t s0 = a.state_off[u];
    const int S = (int)(a.state_off[u + 1] - s0);
    const int64_t a0 = a.arc_off[u];
    const double* __restrict__ beta = a.beta + padded_state_base(s0, u);
    const bool active = lane < item.count;
    const int32_t w = active ? a.words[item.word_begin + lane] : 0;
    int64_t hi = a.arc_off[u + 1];  // arcs of the states above s have been consumed
    for (int s = S - 1; s >= 0; --s) {
      int64_t lo = hi;
      while (lo > a0 && a.src[lo - 1] == s) --lo;  // arcs are grouped by source
      double acc = -INFINITY;
      for (int64_t i = lo; i < hi; ++i) {
        const int d = a.dst[i];
        const double c = arc_cost(a.graph[i], a.acoustic[i], a.ilabel[i] != 0, sp);
        const double v = (active && a.olabel[i] == w) ? beta[d] : b0[(size_t)d * 64 + lane];
        acc = utt_log_add(acc, v - c);
      }
      b0[(size_t)s * 64 + lane] = acc;
      hi = lo;
    }
    if (active) a.logp[item.word_begin + lane] = (S > 0) ? b0[lane] - beta[0] : -INFINITY;
  }
}

}  // namespace

hipError_t launch_utterance(const UttArgs& a, int num_workgroups, hipStream_t stream) {
  if (a.num_items <= 0 || num_workgroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(utterance_kernel, dim3(num_workgroups), dim3(64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace klat
