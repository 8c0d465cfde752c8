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

// src of every arc as a one-byte step from the arc before it (batch creation, once)
__global__ __launch_bounds__(256) void posterior_prep_kernel(PosteriorPrepArgs a) {
  const PostItem item = a.items[blockIdx.x];
  const int count = item.count & (kPostItemFirst - 1);
  if (count > 0 && threadIdx.x == 0) a.item_src[blockIdx.x] = a.src[item.arc_begin];
  bool over = false;
  for (int t = threadIdx.x; t < count; t += 256) {
    const int64_t i = item.arc_begin + t;
    const int s = a.src[i];
    const int step = t == 0 ? 0 : s - a.src[i - 1];
    const int span = a.dst[i] - s;
    over |= (unsigned)step > 255u || (unsigned)span > 255u;
    a.src_delta[i] = (uint8_t)step;
    a.span[i] = (uint8_t)span;
  }
  if (__any(over) && (threadIdx.x & 63) == 0) atomicOr(a.overflow, 1);
}

typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef double double2u __attribute__((ext_vector_type(2), aligned(8)));

// The same pass with src from the steps and dst from the arcs' spans (dst - src, one byte): a lane owns FOUR CONSECUTIVE arcs (one 16-byte load per
// array, two 16-byte stores), a wave 256 consecutive arcs; the sources come from a prefix sum of the
// steps (shuffles inside the wave, one LDS word per wave across the block).
template <bool PEN>
__global__ __launch_bounds__(256) void posterior_delta_kernel(PosteriorArgs a) {
  __shared__ int wave_steps[4];
  const PostItem item = a.items[blockIdx.x];
  const int64_t s0 = a.state_off[item.lattice];
  const int64_t S = a.state_off[item.lattice + 1] - s0;
  const int64_t base = padded_state_base(s0, item.lattice);
  const double* __restrict__ alpha = a.alpha + base;
  const double* __restrict__ beta = a.beta + base;
  const double tf = S > 0 ? alpha[S] : 0.0, tb = S > 0 ? beta[0] : 0.0;
  const double tot = 0.5 * (tb + tf);
  if ((item.count & kPostItemFirst) && threadIdx.x == 0) {
    a.tot_fwd[item.lattice] = tf;
    a.tot_bwd[item.lattice] = tb;
    a.total[item.lattice] = tot;
  }
  const int count = item.count & (kPostItemFirst - 1);
  if (count == 0) return;  // (workgroup-uniform)
  const ScaleParams sp = a.sp;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t0 = wave * 256 + lane * 4;
  const int64_t i0 = item.arc_begin + t0;
  int d[4], dst[4], il[4] = {0, 0, 0, 0};
  float g[4], ac[4];
  if (t0 + 3 < count) {
    const uint32_t d4 = *reinterpret_cast<const uint32_t __attribute__((aligned(1)))*>(a.src_delta + i0);
    const uint32_t s4 = *reinterpret_cast<const uint32_t __attribute__((aligned(1)))*>(a.span + i0);
    const float4u gv = *reinterpret_cast<const float4u*>(a.graph + i0);
    const float4u av = *reinterpret_cast<const float4u*>(a.acoustic + i0);
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = (int)((d4 >> (8 * q)) & 255u), dst[q] = (int)((s4 >> (8 * q)) & 255u), g[q] = gv[q], ac[q] = av[q];
    if (PEN) {
      const int4u iv = *reinterpret_cast<const int4u*>(a.ilabel + i0);
#pragma unroll
      for (int q = 0; q < 4; ++q) il[q] = iv[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = t0 + q < count;
      const int64_t i = ok ? i0 + q : item.arc_begin;
      d[q] = ok ? (int)a.src_delta[i] : 0;
      dst[q] = (int)a.span[i];
      g[q] = a.graph[i];
      ac[q] = a.acoustic[i];
      if (PEN) il[q] = a.ilabel[i];
    }
  }
  // sources: inclusive prefix sums of the steps, item-wide
  int run = d[0] + d[1] + d[2] + d[3];
  int inc = run;
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_steps[wave] = inc;
  __syncthreads();
  int src = a.item_src[blockIdx.x] + inc - run;
  for (int w = 0; w < wave; ++w) src += wave_steps[w];
  double out[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    src += d[q];
    const bool ok = t0 + q < count;
    const double c = arc_cost(g[q], ac[q], PEN ? il[q] != 0 : false, sp);
    const double through = (ok ? alpha[src] : 0.0) + (ok ? beta[src + dst[q]] : 0.0) - c;
    out[q] = (through > -INFINITY) ? through - tot : -INFINITY;
  }
  if (t0 + 3 < count) {
    double2u* o = reinterpret_cast<double2u*>(a.logpost + i0);
    double2u lo, hi;
    lo[0] = out[0], lo[1] = out[1], hi[0] = out[2], hi[1] = out[3];
    o[0] = lo;
    o[1] = hi;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (t0 + q < count) a.logpost[i0 + q] = out[q];
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

// The arcs of a compact descriptor: state s owns the arcs [row[s], row[s + 1]), row = prefix sums of the
// out-degrees; src = s, dst = s + span.
__global__ __launch_bounds__(256) void expand_compact_kernel(ExpandArgs a) {
  __shared__ long long sh[4];
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s0 = a.state_off[u], a0 = a.arc_off[u];
  const int S = (int)(a.state_off[u + 1] - s0);
  const long long A = (long long)(a.arc_off[u + 1] - a0);
  long long carry = 0;  // 64 bits: no sum of 32-bit degrees wraps round into the lattice's range
  bool bad = false;
  for (int base = 0; base < S && !bad; base += 256) {
    const int s = base + tid;
    const int deg_in = s < S ? a.out_degree[s0 + s] : 0;
    const long long deg = deg_in > 0 ? deg_in : 0;
    long long incl = deg;
    for (int d = 1; d < 64; d <<= 1) {
      const long long t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    // a negative degree ends the lattice before anything of this round is written (the barrier's vote is uniform)
    bad = __syncthreads_or(deg_in < 0) != 0;
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    long long off = carry, tot = 0;
    for (int w = 0; w < 4; ++w) {
      if (w < wave) off += sh[w];
      tot += sh[w];
    }
    const long long first = off + incl - deg;
    if (!bad && s < S && first >= 0 && first + deg <= A)
      for (long long i = first; i < first + deg; ++i) {
        a.src[a0 + i] = s;
        a.dst[a0 + i] = s + (int)a.span[a0 + i];
      }
    carry += tot;
  }
  if (tid == 0 && (bad || carry != A)) atomicCAS(&a.status[0], 0, u + 1);
}

hipError_t launch_expand_compact(const ExpandArgs& a, int num_lattices, hipStream_t stream) {
  if (num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(expand_compact_kernel, dim3(num_lattices), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_totals(const TotalsArgs& a, hipStream_t stream) {
  if (a.num_lattices <= 0) return hipSuccess;
  hipLaunchKernelGGL(totals_kernel, dim3((a.num_lattices + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_posterior_prep(const PosteriorPrepArgs& a, int num_items, hipStream_t stream) {
  if (num_items <= 0) return hipSuccess;
  hipLaunchKernelGGL(posterior_prep_kernel, dim3(num_items), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_posterior(const PosteriorArgs& a, int num_items, hipStream_t stream) {
  if (num_items <= 0) return hipSuccess;
  if (a.src_delta != nullptr) {
    if (a.sp.penalized)
      hipLaunchKernelGGL(posterior_delta_kernel<true>, dim3(num_items), dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL(posterior_delta_kernel<false>, dim3(num_items), dim3(256), 0, stream, a);
    return hipGetLastError();
  }
  if (a.sp.penalized)
    hipLaunchKernelGGL(posterior_kernel<true>, dim3(num_items), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(posterior_kernel<false>, dim3(num_items), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace klat
