// klat_math.h -- device-side arithmetic shared by the sweep and the parallel kernels.
//
// Number representation inside the sweep.  The reference keeps log-likelihoods as doubles and
// pays one exp + one log1p per arc *on the serial dependency chain* (LogAdd).
// On the GPU the chain latency, not the flop count, bounds the sweep, so a state's value is
// carried as (mantissa m in [0.5,1) or 0, binary exponent e): value = m * 2^e,
// log-value = ln2*e + ln m.  An arc weight exp(-cost) is split the same way (2^k * wm,
// wm = 2^frac) off the chain; the chain itself is multiply + ldexp + add + frexp.  Natural-log
// doubles are produced once per state by the sweep's producer wave.  The result equals the
// log-semiring sum the reference computes to ~1e-11 (the reference's own LogAdd drops terms
// below exp(-36.04); we keep them).
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

// |cost| rounded to a float, for the batch's largest cost (pack_maxcost_kernel, pack_count_lean_kernel: only its binary
// exponent is used, and both passes must agree on it)
__device__ __forceinline__ float arc_cost_abs_f(float g, float a, bool word, const ScaleParams& sp) {
  float g2, a2;
  arc_cost_parts(g, a, word, sp, g2, a2);
  return fabsf(g2 + a2);
}

// The same cost for a stream term in seven instructions, valid whenever graph_scale >= 0 (see
// StreamCost::general): with a non-negative scale, scale * inf is inf or NaN and the arc stays
// dead either way, so ScaleLattice's special case for Zero needs no test; multiplying by 1 is the
// identity, so "not scaled" needs none; and the word bit of the stream word is bit 30 -- the
// float 2.0f -- so the penalty is fma(2 or 0, pen/2, g): one rounding of g + pen, like the
// reference's float add, and g itself when pen is 0.  (StreamCost: klat_kernels.h)
__device__ __forceinline__ double stream_cost(uint32_t w, float g, float a, const StreamCost& sc) {
  const float g2 = __fmaf_rn(__uint_as_float(w & kWord), sc.half_pen, sc.gs * g);
  return (double)g2 + (double)(sc.as * a);
}

// Per-lane selects with the lane mask in a scalar register pair and the VOP3 encoding of
// v_cndmask_b32.  hipcc prefers the VOP2 encoding with the mask in VCC; runs of those cost ~16
// cycles each once several waves share a CU (tools/ubench5.hip), the VOP3 form costs ~4.6.
__device__ __forceinline__ uint64_t lane_mask(bool cond) { return __builtin_amdgcn_ballot_w64(cond); }
__device__ __forceinline__ int select32(uint64_t mask, int if_set, int if_clear) {
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
  return r;
}
__device__ __forceinline__ double select64(uint64_t mask, double if_set, double if_clear) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, %2, %3, %6\n\t"
      "v_cndmask_b32_e64 %1, %4, %5, %6"
      : "=&v"(lo), "=v"(hi)
      : "v"(__double2loint(if_clear)), "v"(__double2loint(if_set)), "v"(__double2hiint(if_clear)),
        "v"(__double2hiint(if_set)), "s"(mask));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double select64_or_zero(uint64_t mask, double if_set) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, 0, %2, %4\n\t"
      "v_cndmask_b32_e64 %1, 0, %3, %4"
      : "=&v"(lo), "=v"(hi)
      : "v"(__double2loint(if_set)), "v"(__double2hiint(if_set)), "s"(mask));
  return __hiloint2double(hi, lo);
}

// exp(-c) = wm * 2^k with wm = 2^f, |f| <= 1/2.  dead = the weight is exactly zero (cost +inf or
// NaN; padding terms carry an infinite cost), returned as wm = 0 with an unspecified k.
// 2^f by the degree-9 polynomial of tools/exp2_poly.py (Chebyshev interpolant, relative error
// < 1.9e-14), evaluated as one block of nine v_fma_f64 with the coefficients in scalar
// registers: hipcc otherwise keeps them in VGPRs and emits v_mov_b64 + v_fmac_f64 per step.
// `live`: lanes whose term counts at all (a lane beyond a job's stored lanes carries its lattice's
// first term a second time and must weigh nothing)
// `c9v`: the polynomial's leading coefficient in a VECTOR register (arc_weight_c9()): the first fma
// has two constant operands and only one may come from a scalar register; handed in by the caller so
// that it is made once per kernel, not once per term
__device__ __forceinline__ double arc_weight_c9() {
  double v;
  asm volatile("v_mov_b64 %0, %1" : "=v"(v) : "s"(1.02059054137655954707e-07));
  return v;
}
__device__ __forceinline__ void arc_weight(double c, double& wm, int& k, bool& dead, uint64_t live, double c9v) {
  dead = !(c < 1.0e30);
  // NaN -> -1e8; +inf stays (wm is forced to 0 below).  The bare instruction: fmax() first quiets a
  // possible signalling NaN with a second v_max_f64, and c comes out of arithmetic (always quiet)
  double cc;
  asm("v_max_f64 %0, %1, %2" : "=v"(cc) : "v"(c), "s"(-1.0e8));
  const double kf = rint(-cc * kLog2e);
  const double f = fma(-cc, kLog2e, -kf);
  // the instruction saturates (kf is -inf for a dead term); a C++ cast would be undefined there
  asm("v_cvt_i32_f64 %0, %1" : "=v"(k) : "v"(kf));
  double p;
  asm("v_fma_f64 %0, %1, %2, %3\n\t"
      "v_fma_f64 %0, %0, %2, %4\n\t"
      "v_fma_f64 %0, %0, %2, %5\n\t"
      "v_fma_f64 %0, %0, %2, %6\n\t"
      "v_fma_f64 %0, %0, %2, %7\n\t"
      "v_fma_f64 %0, %0, %2, %8\n\t"
      "v_fma_f64 %0, %0, %2, %9\n\t"
      "v_fma_f64 %0, %0, %2, %10\n\t"
      "v_fma_f64 %0, %0, %2, %11"
      : "=&v"(p)
      : "v"(c9v), "v"(f), "s"(1.32596452448871106013e-06), "s"(1.52526729248507371552e-05),
        "s"(1.54034337512510096551e-04), "s"(1.33335582007945924535e-03),
        "s"(9.61812919393431262072e-03), "s"(5.55041086646516704239e-02),
        "s"(2.40226506956403407234e-01), "s"(6.93147180559946174405e-01),
        "s"(1.00000000000001354472e+00));
  wm = select64_or_zero(lane_mask(c < 1.0e30) & live, p);
}

// ln(m * 2^e) for a normalised mantissa m in [0.5,1) (or 0 = log-zero) with a 16-entry table in
// LDS instead of the library log (~50 instructions): 2m = c_i * (1 + r), |r| <= 1/32, and
// log1p(r) by its degree-10 series (next term < 3e-18).  The series runs as one block of
// v_fma_f64 with its constants in scalar registers or inline (see arc_weight).
// The table read is a separate step so that callers can overlap its LDS latency with other work.
__device__ __forceinline__ int ln_table_index(double m) {
  return (__double2hiint(m) >> 16) & (kLnTabSize - 1);  // top four mantissa bits
}
__device__ __forceinline__ double table_ln(double m, int e, const LnTabEntry& t) {
  const double r = fma(m, t.inv2, -1.0);
  const double c10 = -0.1;
  double p, v;
  asm("v_fma_f64 %0, %1, %2, %3\n\t"
      "v_fma_f64 %0, %0, %2, %4\n\t"
      "v_fma_f64 %0, %0, %2, %5\n\t"
      "v_fma_f64 %0, %0, %2, %6\n\t"
      "v_fma_f64 %0, %0, %2, %7\n\t"
      "v_fma_f64 %0, %0, %2, %8\n\t"
      "v_fma_f64 %0, %0, %2, %9\n\t"
      "v_fma_f64 %0, %0, %2, -0.5\n\t"
      "v_fma_f64 %0, %0, %2, 1.0"
      : "=&v"(p)
      : "v"(c10), "v"(r), "s"(1.0 / 9.0), "s"(-0.125), "s"(1.0 / 7.0), "s"(-1.0 / 6.0), "s"(0.2), "s"(-0.25), "s"(1.0 / 3.0));
  const double lnm = fma(p, r, t.lnc);
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(v) : "v"((double)e), "s"(kLn2), "v"(lnm));
  return select64(lane_mask(m == 0.0), -INFINITY, v);
}

// (M,E) += (tm,te) in the scaled-linear domain, safe for any exponent gap.
__device__ __forceinline__ void me_accumulate(double& M, int& E, double tm, int te) {
  const int hi = max(E, te);
  M = ldexp(M, E - hi) + ldexp(tm, te - hi);
  E = hi;
}

}  // namespace klat
#endif
