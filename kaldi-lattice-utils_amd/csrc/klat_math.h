// klat_math.h -- device-side arithmetic shared by the sweep and the parallel kernels.
//
// Number representation inside the sweep.  The reference keeps log-likelihoods as doubles and
// pays one exp + one log1p per arc *on the serial dependency chain* (LogAdd).
// On the GPU the chain latency, not the flop count, bounds the sweep, so a state's value is
// carried as (mantissa m in [0.5,1) or 0, binary exponent e): value = m * 2^e,
// log-value = ln2*e + ln m.  An arc weight exp(-cost) is split the same way (2^k * wm,
// wm = 2^frac) off the chain; the chain itself is multiply + ldexp + add + frexp.  Natural-log
// doubles are produced once per state / arc by the fully parallel kernels.  The result equals
// the log-semiring sum the reference computes to ~1e-12 (the reference's own LogAdd drops
// terms below exp(-36.04); we keep them).
#ifndef KLAT_MATH_H_
#define KLAT_MATH_H_
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "klat_device.h"
#include "klat_kernels.h"

namespace klat {

static constexpr double kLn2 = 0.693147180559945309417232121458;
static constexpr double kLog2e = 1.442695040888963407359924681002;

// a2 + a3 fused into the cost load: ScaleLattice forms
// (double)scale * (double)x and stores float -- the double product of two floats is exact, so
// that single rounding is one IEEE float multiply; Value1 == +inf stays Zero;
// AddWordInsPenToCompactLattice adds pen (float) to Value1 of word arcs; ConvertToCost sums the
// two floats in double.
__device__ __forceinline__ void arc_cost_parts(float g, float a, bool word, const ScaleParams& sp,
                                               float& g_out, float& a_out) {
  if (sp.scaled) {
    a = (g == INFINITY) ? INFINITY : sp.as_f * a;
    g = (g == INFINITY) ? INFINITY : sp.gs_f * g;
  }
  if (sp.penalized && word) g = g + sp.pen;
  g_out = g;
  a_out = a;
}

__device__ __forceinline__ double arc_cost(float g, float a, bool word, const ScaleParams& sp) {
  float g2, a2;
  arc_cost_parts(g, a, word, sp, g2, a2);
  return (double)g2 + (double)a2;
}

// v_fma_f64 with the addend taken from a scalar register pair: keeps the polynomial a pure
// 3-address FMA chain (hipcc otherwise emits v_mov_b64 + v_fmac_f64 per step for a live
// VGPR constant).
__device__ __forceinline__ double fma_sconst(double a, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
}

// exp(-c) = wm * 2^k with wm = 2^f, |f| <= 1/2.  dead = the weight is exactly zero
// (cost +inf or NaN, or a padding term), returned as wm = 0.  2^f by a degree-12 Taylor polynomial in f (coefficients ln2^n/n!):
// |f ln2| <= 0.3466 -> truncation error < 2e-16 relative.
__device__ __forceinline__ void arc_weight(double c, bool nil, double& wm, int& k, bool& dead) {
  dead = !(c < 1.0e30) | nil;
  const double cc = dead ? 0.0 : fmax(c, -1.0e8);
  const double kf = rint(-cc * kLog2e);
  const double f = fma(-cc, kLog2e, -kf);
  k = (int)kf;
  double p = 2.5678435993488205142e-11;    // ln2^12/12!
  p = fma_sconst(p, f, 4.4455382718708114976e-10);  // ln2^11/11!
  p = fma_sconst(p, f, 7.05491162080112333e-09);   // ln2^10/10!
  p = fma_sconst(p, f, 1.01780860092396997e-07);   // ln2^9/9!
  p = fma_sconst(p, f, 1.32154867901443095e-06);   // ln2^8/8!
  p = fma_sconst(p, f, 1.52527338040598403e-05);   // ln2^7/7!
  p = fma_sconst(p, f, 1.54035303933816099e-04);   // ln2^6/6!
  p = fma_sconst(p, f, 1.33335581464284434e-03);   // ln2^5/5!
  p = fma_sconst(p, f, 9.61812910762847716e-03);   // ln2^4/4!
  p = fma_sconst(p, f, 5.55041086648215800e-02);   // ln2^3/3!
  p = fma_sconst(p, f, 2.40226506959100712e-01);   // ln2^2/2!
  p = fma_sconst(p, f, 6.93147180559945309e-01);   // ln2
  p = fma(p, f, 1.0);
  wm = dead ? 0.0 : p;
}

__device__ __forceinline__ double me_to_ln(double m, int e) {
  return (m == 0.0) ? -INFINITY : fma((double)e, kLn2, log(m));
}

// (M,E) += (tm,te) in the scaled-linear domain, safe for any exponent gap.
__device__ __forceinline__ void me_accumulate(double& M, int& E, double tm, int te) {
  const int hi = max(E, te);
  M = ldexp(M, E - hi) + ldexp(tm, te - hi);
  E = hi;
}

}  // namespace klat
#endif
