// klat_kernels.hip -- hand-written gfx950 kernels of the lattice forward-backward path.
//
// Replaces the CPU graph walk of ComputeLatticeAlphasAndBetas [upstream Kaldi], called at
// latbin/lattice-to-word-frame-post.cc:88-90, kwsbin2/lattice-word-index-segment.cc:63 and
// kwsbin2/lattice-word-index-position.cc:66, plus the per-arc through-likelihood of
// kwsbin2/lattice-word-index-segment.cc:160-163.
//
// Number representation inside the sweep.  The reference keeps log-likelihoods as doubles and
// pays one exp + one log1p per arc *on the serial dependency chain* (LogAdd).  On the GPU the
// chain latency, not the flop count, bounds the sweep, so a state's value is carried as
// (mantissa m in [0.5,1) or 0, binary exponent e):  value = m * 2^e, log-value = ln2*e + ln m.
// An arc weight exp(-cost) is split the same way (2^k * wm, wm = 2^frac) off the chain; the
// chain itself is ldexp + fma + frexp.  Natural-log doubles are produced once per state / arc
// by the fully parallel kernels below.  The result equals the log-semiring sum the reference
// computes to ~1e-13 (the reference's own LogAdd drops terms below exp(-36), we keep them).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_kernels.h"

namespace klat {

static constexpr double kLn2 = 0.693147180559945309417232121458;
static constexpr double kLog2e = 1.442695040888963407359924681002;

// a2 + a3 fused into the cost load: ScaleLattice forms gs*Value1, as*Value2 in double and stores
// float; AddWordInsPenToCompactLattice adds pen (float) to Value1 of word arcs; ConvertToCost
// sums the two floats in double.
__device__ __forceinline__ double arc_cost(float g, float a, bool word, const ScaleParams& sp) {
  if (sp.scaled) {
    if (g == INFINITY) {
      a = INFINITY;
    } else {
      g = (float)(sp.gs * (double)g);
      a = (float)(sp.as * (double)a);
    }
  }
  if (sp.penalized && word) g = g + sp.pen;
  return (double)g + (double)a;
}

// exp(-c) = wm * 2^k with wm = 2^f, |f| <= 1/2.  dead = weight is exactly zero (cost +inf/NaN).
__device__ __forceinline__ void arc_weight(double c, double& wm, int& k, bool& dead) {
  dead = !(c < 1.0e30);
  double cc = dead ? 0.0 : fmax(c, -1.0e8);
  double t = -cc * kLog2e;
  double kf = rint(t);
  double f = fma(-cc, kLog2e, -kf);
  k = (int)kf;
  double x = f * kLn2;  // |x| <= 0.3466; Taylor degree 12: error < 2e-16 relative
  double p = 2.08767569878680989792e-09;          // 1/12!
  p = fma(p, x, 2.50521083854417187751e-08);      // 1/11!
  p = fma(p, x, 2.75573192239858906526e-07);      // 1/10!
  p = fma(p, x, 2.75573192239858906526e-06);      // 1/9!
  p = fma(p, x, 2.48015873015873015873e-05);      // 1/8!
  p = fma(p, x, 1.98412698412698412698e-04);      // 1/7!
  p = fma(p, x, 1.38888888888888888889e-03);      // 1/6!
  p = fma(p, x, 8.33333333333333333333e-03);      // 1/5!
  p = fma(p, x, 4.16666666666666666667e-02);      // 1/4!
  p = fma(p, x, 1.66666666666666666667e-01);      // 1/3!
  p = fma(p, x, 0.5);
  p = fma(p, x, 1.0);
  p = fma(p, x, 1.0);
  wm = p;
}

__device__ __forceinline__ double me_to_ln(double m, int e) {
  return (m == 0.0) ? -INFINITY : fma((double)e, kLn2, log(m));
}

// (M,E) += (tm,te) in the scaled-linear domain.
__device__ __forceinline__ void me_accumulate(double& M, int& E, double tm, int te) {
  int hi = max(E, te);
  M = ldexp(M, E - hi) + ldexp(tm, te - hi);
  E = hi;
}

// ---------------------------------------------------------------------------------------------
// Lockstep sweep.  One wave per job; a job is 64/G lattices, G lanes each, walking the forward
// (dir 0) or backward (dir 1) arc stream of those lattices round by round.  The values of the
// last W visited states of each lattice live in an LDS ring ([slot][lattice] so that the 64
// lanes of a ds_read_b64 fall in distinct banks for G = 1); older sources (FAR) are re-read
// from the per-state arrays this kernel itself wrote.  The arc stream is the only HBM stream:
// 12 B per term, read once, as 1 KiB dwordx4 wave loads, prefetched kPrefetchChunks deep.
template <int G>
__global__ __launch_bounds__(64) void sweep_kernel(SweepArgs args) {
  extern __shared__ double smem[];
  constexpr int LPW = 64 / G;
  constexpr int D = kPrefetchChunks;
  const int W = args.ring_slots;
  double* ring_m = smem;                           // [W][LPW]
  int* ring_e = reinterpret_cast<int*>(smem + (size_t)W * LPW);  // [W][LPW]

  const int lane = threadIdx.x;
  const int slot = lane / G;
  const int sub = lane % G;
  const SweepJob job = args.jobs[blockIdx.x];
  const LaneInfo li = args.lanes[job.lane_tab + slot];
  const int S = li.num_states;
  const bool bwd = job.dir != 0;
  double* vm = bwd ? args.bm : args.fm;
  int* ve = bwd ? args.be : args.fe;
  const int64_t base = li.state_base;
  const ScaleParams sp = args.sp;

  const int4* __restrict__ sw = reinterpret_cast<const int4*>(args.sw) + job.chunk_off * 64 + lane;
  const float4* __restrict__ sg = reinterpret_cast<const float4*>(args.sg) + job.chunk_off * 64 + lane;
  const float4* __restrict__ sa = reinterpret_cast<const float4*>(args.sa) + job.chunk_off * 64 + lane;

  int p = 0;  // position of the state this lattice is currently accumulating
  double accM = 0.0;
  int accE = kExpMin;

  int4 wb[D];
  float4 gb[D], ab[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const int64_t cn = min(j, job.nchunks - 1);
    wb[j] = sw[cn * 64];
    gb[j] = sg[cn * 64];
    ab[j] = sa[cn * 64];
  }

  auto round = [&](int wi, float g, float a) {
    const uint32_t w = (uint32_t)wi;
    const double c = arc_cost(g, a, (w & kWord) != 0, sp);
    double wm;
    int k;
    bool dead;
    arc_weight(c, wm, k, dead);
    const int rel = (int)(w & kRelMask);
    const int rs = (p - rel) & (W - 1);
    double sm = ring_m[rs * LPW + slot];
    int se = ring_e[rs * LPW + slot];
    if (rel == 0) {
      sm = 0.5;  // constant one = 0.5 * 2^1
      se = 1;
    }
    if (__any((w & kFar) != 0)) {
      if (w & kFar) {
        const int q = p - rel;
        const int64_t idx = base + (bwd ? (int64_t)(S - 1 - q) : (int64_t)q);
        sm = __hip_atomic_load(&vm[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        se = __hip_atomic_load(&ve[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const bool zero = ((w & kNil) != 0) | dead | (sm == 0.0);
    const double tm = zero ? 0.0 : sm * wm;
    const int te = zero ? kExpMin : se + k;
    me_accumulate(accM, accE, tm, te);
    if (w & kLast) {
      if (G > 1) {
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
          const double oM = __shfl_xor(accM, off, 64);
          const int oE = __shfl_xor(accE, off, 64);
          me_accumulate(accM, accE, oM, oE);
        }
      }
      const double m = __builtin_amdgcn_frexp_mant(accM);
      const int e = (accM == 0.0) ? kExpMin : accE + __builtin_amdgcn_frexp_exp(accM);
      if (sub == 0) {
        const int ws = p & (W - 1);
        ring_m[ws * LPW + slot] = m;
        ring_e[ws * LPW + slot] = e;
        if (!bwd && p == S) {  // virtual super-final state: tot_forward
          args.totf_m[li.lattice] = m;
          args.totf_e[li.lattice] = e;
        } else {
          const int64_t idx = base + (bwd ? (int64_t)(S - 1 - p) : (int64_t)p);
          vm[idx] = m;
          ve[idx] = e;
        }
      }
      p += 1;
      accM = 0.0;
      accE = kExpMin;
    }
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
      round(w4.x, g4.x, a4.x);
      round(w4.y, g4.y, a4.y);
      round(w4.z, g4.z, a4.z);
      round(w4.w, g4.w, a4.w);
    }
  }
}

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
    tf = me_to_ln(a.totf_m[u], a.totf_e[u]);
    tb = me_to_ln(a.bm[s0], a.be[s0]);
  }
  a.tot_fwd[u] = tf;
  a.tot_bwd[u] = tb;
  a.total[u] = 0.5 * (tb + tf);
}

// a8 + normalisation: logpost[i] = alpha[src] - cost + beta[dst] - total, fully parallel over
// arcs in caller order.  Reads src,dst,graph,acoustic (16 B/arc, coalesced), gathers the
// (m,e) pairs of the two end states (L2-resident: arcs are grouped by source), writes 8 B/arc.
template <bool PEN>
__global__ __launch_bounds__(256) void posterior_kernel(PosteriorArgs a) {
  const PostItem item = a.items[blockIdx.x];
  const int64_t base = a.state_off[item.lattice];
  const double tot = a.total[item.lattice];
  const ScaleParams sp = a.sp;
  for (int t = threadIdx.x; t < item.count; t += 256) {
    const int64_t i = item.arc_begin + t;
    const int s = a.src[i];
    const int d = a.dst[i];
    const bool word = PEN ? (a.ilabel[i] != 0) : false;
    const double c = arc_cost(a.graph[i], a.acoustic[i], word, sp);
    const double prod = a.fm[base + s] * a.bm[base + d];
    const int e = a.fe[base + s] + a.be[base + d];
    double r = -INFINITY;
    if (prod != 0.0 && c < 1.0e30) r = fma((double)e, kLn2, log(prod)) - c - tot;
    a.logpost[i] = r;
  }
}

// alpha/beta in natural log for the B2 interface (vector<double>* alpha, *beta).
__global__ __launch_bounds__(256) void convert_kernel(ConvertArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.num_states; j += stride) {
    a.alpha[j] = me_to_ln(a.fm[j], a.fe[j]);
    a.beta[j] = me_to_ln(a.bm[j], a.be[j]);
  }
}

// ---------------------------------------------------------------------------------------------
hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream) {
  if (num_jobs <= 0) return hipSuccess;
  const int LPW = 64 / G;
  const size_t lds = (size_t)args.ring_slots * LPW * (sizeof(double) + sizeof(int));
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

hipError_t launch_convert(const ConvertArgs& a, hipStream_t stream) {
  if (a.num_states <= 0) return hipSuccess;
  int64_t blocks = (a.num_states + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(convert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace klat
