// t2fit_lane.h -- per-lane (one voxel) building blocks of the T2 fit kernels.
//
// Everything here is plain arithmetic on one voxel's echo samples so that the very same source
// is compiled (a) by hipcc into the gfx950 kernels in t2fit_kernels.hip and (b) by g++ into the
// host-side lane simulator that tests/ uses to check solver logic where no GPU is present.  The
// simulator is test infrastructure; the product library contains device code only.
//
// Reference formulas: run_t2mapping.py:129-177 (models/objectives), utils/t2map_utils.py:62-89
// (residual map).  T2, TE in milliseconds.
#pragma once

#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/t2fit.h"

#if defined(__HIPCC__)
#define T2_HD __host__ __device__ __forceinline__
#define T2_DEVICE_COMPILE 1
#define T2_UNROLL _Pragma("unroll")
#else
#define T2_HD inline
#define T2_UNROLL
#endif

namespace t2fit {

// Device-side copy of t2fit_config with both precisions of the echo times.  Passed by value as a
// kernel argument: it lives in the kernarg segment and uniform-index reads become scalar loads.
struct LaneParams {
  int32_t model, solver, precision, n_te;
  int32_t no_prior, norm, maxls, maxiter, maxfun, numpy_legacy;
  double te[T2FIT_MAX_TE];
  float te_f[T2FIT_MAX_TE];
  double x0[3], lb[3], ub[3];
  double ftol, gtol, fd_step, lm_xtol;
  double np_k_ub, np_t2_lb, np_t2_ub;
  // LM works in R = 1/T2: reciprocals of the T2 box in force (table or no-prior) and of the start point,
  // formed once on the host instead of by three float64 divisions per voxel
  double lm_r_lo, lm_r_hi, lm_r_x0;
  // Taylor coefficients 1/11! .. 1/2! of t2_exp_res(), read from the kernel-argument segment into scalar
  // registers: as literals the compiler re-materialises each one with two v_mov per use (a third of the
  // vector instructions of the residual pass)
  double exp_c[10];
  double inv_n;  // 1 / nTE (the objectives are means over the echoes)
  double te_max;  // largest echo time: bounds how far exp(-2 te / T2) moves with T2 (Lbfgsb::eval, shared square-root seed)
  double lbfgsb_tol;  // factr * epsmch = (ftol / eps) * eps, the relative-reduction stop of L-BFGS-B
};

// One voxel's samples: element i lives at p[i*stride].  In the kernels p points into LDS (one
// column per lane, stride = padded block width); in the host simulator stride is 1.
struct EchoView {
  const float* p;
  int stride;
  T2_HD float operator[](int i) const { return p[i * stride]; }
};

struct LaneResult {
  double x[3];  // k, T2, sigma
  double fun;
  int32_t nit;
  int32_t nfev;  // objective evaluations (scipy result.nfev)
  uint8_t status;
};

// ---- small math wrappers ---------------------------------------------------------------------
T2_HD double t2_exp(double x) { return exp(x); }
T2_HD double t2_sqrt(double x) { return sqrt(x); }
T2_HD double t2_log(double x) { return log(x); }
#if defined(__HIP_DEVICE_COMPILE__)
T2_HD float t2_exp(float x) { return __expf(x); }        // v_exp_f32 path
T2_HD float t2_log(float x) { return __logf(x); }        // v_log_f32 path (LM seed only)
T2_HD float t2_rsqrt(float x) { return __frsqrt_rn(x); }  // v_rsq_f32
T2_HD float t2_rcp(float x) { return __frcp_rn(x); }
#else
T2_HD float t2_log(float x) { return logf(x); }
T2_HD float t2_exp(float x) { return expf(x); }
T2_HD float t2_rsqrt(float x) { return 1.0f / sqrtf(x); }
T2_HD float t2_rcp(float x) { return 1.0f / x; }
#endif
// reciprocal good to ~1 ulp without the IEEE division sequence (v_rcp_f64 seed, two Newton steps);
// used only where the last bit is immaterial (scalings inside the quasi-Newton matrix rebuild)
#if defined(__HIP_DEVICE_COMPILE__)
T2_HD double t2_fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
#else
T2_HD double t2_fast_rcp(double x) { return 1.0 / x; }
#endif
// a / b through the 1-ulp reciprocal and one residual correction: the correctly rounded quotient in all but
// rare last-bit cases, with a dependent chain about half as long as the IEEE division sequence.  Used
// between objective evaluations (step lengths, breakpoints, the 3x3 solve), where a wave at one wave per
// SIMD waits on exactly these chains.
#if defined(__HIP_DEVICE_COMPILE__)
T2_HD double t2_fdiv(double a, double b) {
  const double r = t2_fast_rcp(b);
  const double q = a * r;
  return fma(fma(-q, b, a), r, q);
}
T2_HD double t2_rcp_for_div(double b) { return t2_fast_rcp(b); }  // r for t2_div_by_rcp(a, b, r) below
#else
T2_HD double t2_fdiv(double a, double b) { return a / b; }
T2_HD double t2_rcp_for_div(double b) { return 1.0 / b; }
#endif
// a / b from r = 1/b (correctly rounded) with one residual correction: the correctly rounded quotient
// except for rare last-bit cases, at three FMA-class operations instead of an IEEE division sequence
T2_HD double t2_div_by_rcp(double a, double b, double r) {
  const double q = a * r;
  return fma(fma(-q, b, a), r, q);
}
T2_HD double t2_rsqrt(double x) { return 1.0 / sqrt(x); }
// 1 / sqrt(x) to ~1 ulp for x > 0 (v_rsq_f64 seed, two Newton steps), same use as t2_fast_rcp
#if defined(__HIP_DEVICE_COMPILE__)
T2_HD double t2_fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = fma(fma(-hx * y, y, 0.5), y, y);
  y = fma(fma(-hx * y, y, 0.5), y, y);
  return y;
}
#else
T2_HD double t2_fast_rsqrt(double x) { return 1.0 / sqrt(x); }
#endif
T2_HD double t2_rcp(double x) { return 1.0 / x; }
T2_HD float t2_sqrt(float x) { return sqrtf(x); }

// exp() and sqrt() of the objective evaluations, device side: the device library's own float64 sequences
// (same constants, same operations, same order: bit-identical results) without their range handling.
// exp: arguments here are -te/T2 or -2te/T2 with T2 >= the lower bound, never positive: the overflow select is
// dead code, and ldexp() already delivers the denormal and zero results of the underflow select (checked against
// the library down to -1.4e9; config_check keeps the argument above that).  sqrt: arguments are
// k^2 E + sigma^2 in [0, 1e10]: the 2^-767 rescaling never triggers; zero stays exact (see below).
#if defined(__HIP_DEVICE_COMPILE__)
T2_HD double t2_exp_core(double x) {
  const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
  double r = fma(-0x1.62e42fefa39efp-1, n, x);
  r = fma(-0x1.abc9e3b39803fp-56, n, r);
  double p = fma(0x1.ade156a5dcb37p-26, r, 0x1.28af3fca7ab0cp-22);
  p = fma(r, p, 0x1.71dee623fde64p-19);
  p = fma(r, p, 0x1.a01997c89e6b0p-16);
  p = fma(r, p, 0x1.a01a014761f6ep-13);
  p = fma(r, p, 0x1.6c16c1852b7b0p-10);
  p = fma(r, p, 0x1.1111111122322p-7);
  p = fma(r, p, 0x1.55555555502a1p-5);
  p = fma(r, p, 0x1.5555555555511p-3);
  p = fma(r, p, 0x1.000000000000bp-1);
  p = fma(r, p, 1.0);
  p = fma(r, p, 1.0);
  return __builtin_ldexp(p, (int)n);
}
T2_HD double t2_sqrt_core(double x) {
  // rsq(0) = inf would turn the iteration into NaN; capped at 1e300 (far above rsq of any normal number, so
  // nothing else changes) every step below yields an exact 0 for x = 0: one v_min instead of a compare and
  // two selects behind each of the 32 square roots of an evaluation
  const double y = fmin(__builtin_amdgcn_rsq(x), 1e300);
  double g = x * y, h = y * 0.5;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  double d = fma(-g, g, x);
  h = fma(h, r, h);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  return g;
}
#else
T2_HD double t2_exp_core(double x) { return exp(x); }
T2_HD double t2_sqrt_core(double x) { return sqrt(x); }
#endif

// The square root of x together with h = 1 / (2 sqrt(x)) (to ~2^-51), and the square root of a value NEAR x from that h:
// the four radicands of one echo of a forward-difference evaluation differ by the step, parts in 1e8 or less, so the three
// displaced ones start from the first one's refined reciprocal root (one more coupled Newton step brings g and h to ~1e-15 of
// the new argument's, then the same residual correction as above) -- 6 instead of 11 instructions each.  Both sequences end in
// g + (a - g^2) h with g within an ulp and h good to 2^-50: the correctly rounded root, the same bits (Markstein).
// x > 0 and |a - x| <= 2^-20 x are the caller's business (Lbfgsb::eval checks both once per evaluation).
// The sequences are plain FMA arithmetic (`_seq`, host and device: tests/test_lane_solver_hostsim.py runs them on the CPU
// from a seed as coarse as the hardware's, against the correctly rounded sqrt); the device entry points feed them v_rsq_f64.
T2_HD double t2_sqrt_from_seed_seq(double x, double y, double& h_out) {  // y ~ 1 / sqrt(x) to 2^-23 or better, x > 0
  double g = x * y, h = y * 0.5;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  double d = fma(-g, g, x);
  h = fma(h, r, h);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  h_out = h;
  return g;
}
T2_HD double t2_rsqrt_half_near_seq(double a, double h) {  // 1 / (2 sqrt(a)) from h = 1 / (2 sqrt(x)) of a nearby x
  const double g = a * (h + h);
  return fma(h, fma(-h, g, 0.5), h);
}
T2_HD double t2_sqrt_near_seq(double a, double h) {
  double g = a * (h + h);
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  const double hi = fma(h, r, h);
  const double d = fma(-g, g, a);
  return fma(d, hi, g);
}
// sqrt(a) from h = 1 / (2 sqrt(a)) good to 2^-50 (the Rician lane keeps only h across its Chebyshev loop)
T2_HD double t2_sqrt_from_h_seq(double a, double h) {
  const double g0 = a * (h + h);
  return fma(fma(-g0, g0, a), h, g0);
}
#if defined(__HIP_DEVICE_COMPILE__)
T2_HD double t2_sqrt_core_h(double x, double& h_out) { return t2_sqrt_from_seed_seq(x, __builtin_amdgcn_rsq(x), h_out); }
T2_HD double t2_sqrt_near(double a, double h) { return t2_sqrt_near_seq(a, h); }
T2_HD double t2_rsqrt_half_near(double a, double h) { return t2_rsqrt_half_near_seq(a, h); }
T2_HD double t2_sqrt_from_h(double a, double h) { return t2_sqrt_from_h_seq(a, h); }
#else
T2_HD double t2_sqrt_core_h(double x, double& h_out) { h_out = 0.5 / sqrt(x); return sqrt(x); }
T2_HD double t2_sqrt_near(double a, double) { return sqrt(a); }
T2_HD double t2_rsqrt_half_near(double a, double) { return 0.5 / sqrt(a); }
T2_HD double t2_sqrt_from_h(double a, double) { return sqrt(a); }
#endif

template <typename T> T2_HD T t2_min(T a, T b) { return a < b ? a : b; }
template <typename T> T2_HD T t2_max(T a, T b) { return a > b ? a : b; }
// a NaN bound is ignored (a voxel whose S(TE0) is NaN keeps the table start point, as scipy reports it)
template <typename T> T2_HD T t2_clip(T x, T lo, T hi) {
  T r = x;
  if (lo > r) r = lo;
  if (hi < r) r = hi;
  return r;
}
template <typename T> T2_HD T t2_abs(T a) { return a < T(0) ? -a : a; }
T2_HD bool t2_finite(double x) { return (x - x) == 0.0; }
T2_HD bool t2_finite(float x) { return (x - x) == 0.0f; }

template <typename T> struct TeOf;
template <> struct TeOf<double> {
  T2_HD static double at(const LaneParams& P, int i) { return P.te[i]; }
};
template <> struct TeOf<float> {
  T2_HD static float at(const LaneParams& P, int i) { return P.te_f[i]; }
};

// compile-time loop: f(std::integral_constant<int, J>) for J in [J0, JN)
template <int J, int JN, class F> T2_HD void static_for(F&& f) {
  if constexpr (J < JN) {
    f(std::integral_constant<int, J>{});
    static_for<J + 1, JN>(f);
  }
}

// ---- exponentially scaled modified Bessel function I0 (scipy.special.i0e = Cephes i0e) ---------
// Chebyshev expansions from Cephes i0.c (public domain, Moshier): [0,8] and (8,inf).
//
// The coefficient tables are read from memory with a wave-uniform index -- scalar loads into scalar registers, one
// operand of the recurrence's add -- and the recurrence is a real loop.  (Round 2 had them as literals in fully
// unrolled code: every coefficient was re-materialised by two moves per use and the Rician-likelihood kernel for
// eight echoes was 138 KB of instructions, twice the instruction cache two CUs share.)  Device side the tables are
// __constant__ and NOT const, so the compiler cannot fold the loads back into literals.
#if defined(__HIPCC__)
#define T2_TABLE __device__ __constant__
#else
#define T2_TABLE static const
#endif
T2_TABLE double t2_i0e_A[30] = {
    -4.41534164647933937950E-18, 3.33079451882223809783E-17,  -2.43127984654795469359E-16,
    1.71539128555513303061E-15,  -1.16853328779934516808E-14, 7.67618549860493561688E-14,
    -4.85644678311192946090E-13, 2.95505266312963983461E-12,  -1.72682629144155570723E-11,
    9.67580903537323691224E-11,  -5.18979560163526290666E-10, 2.65982372468238665035E-9,
    -1.30002500998624804212E-8,  6.04699502254191894932E-8,   -2.67079385394061173391E-7,
    1.11738753912010371815E-6,   -4.41673835845875056359E-6,  1.64484480707288970893E-5,
    -5.75419501008210370398E-5,  1.88502885095841655729E-4,   -5.76375574538582365885E-4,
    1.63947561694133579842E-3,   -4.32430999505057594430E-3,  1.05464603945949983183E-2,
    -2.37374148058994688156E-2,  4.93052842396707084878E-2,   -9.49010970480476444210E-2,
    1.71620901522208775349E-1,   -3.04682672343198398683E-1,  6.76795274409476084995E-1};
T2_TABLE double t2_i0e_B[25] = {
    -7.23318048787475395456E-18, -4.83050448594418207126E-18, 4.46562142029675999901E-17,
    3.46122286769746109310E-17,  -2.82762398051658348494E-16, -3.42548561967721913462E-16,
    1.77256013305652638360E-15,  3.81168066935262242075E-15,  -9.55484669882830764870E-15,
    -4.15056934728722208663E-14, 1.54008621752140982691E-14,  3.85277838274214270114E-13,
    7.18012445138366623367E-13,  -1.79417853150680611778E-12, -1.32158118404477131188E-11,
    -3.14991652796324136454E-11, 1.18891471078464383424E-11,  4.94060238822496958910E-10,
    3.39623202570838634515E-9,   2.26666899049817806459E-8,   2.04891858946906374183E-7,
    2.89137052083475648297E-6,   6.88975834691682398426E-5,   3.36911647825569408990E-3,
    8.04490411014108831608E-1};

// Cephes' recurrence as its C source evaluates it without fused multiply-add (scipy's wheels: baseline x86-64): a product, a
// difference, a sum.  -DT2_I0E_FMA (experiment only, tools/experiments/r03_exp13.sh) fuses the product into the difference.
#if defined(T2_I0E_FMA)
#define T2_CLENSHAW(z, b1, b2, c) (fma(z, b1, -(b2)) + (c))
#else
#define T2_CLENSHAW(z, b1, b2, c) ((z) * (b1) - (b2) + (c))
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define T2_NOUNROLL _Pragma("nounroll")
#define T2_WAVE_ANY(p) (__ballot(p) != 0ull)
#else
#define T2_NOUNROLL
#define T2_WAVE_ANY(p) (p)
#endif

// Cephes chbevl, one value (the reference form: tests compare the four-wide loop below with it bit for bit)
T2_HD double t2_chbevl(double x, const double* c, int n) {
  double b0 = c[0], b1 = 0.0, b2 = 0.0;
  for (int i = 1; i < n; ++i) {
    b2 = b1;
    b1 = b0;
    b0 = x * b1 - b2 + c[i];
  }
  return 0.5 * (b0 - b2);
}

T2_HD double t2_i0e(double x) {
  if (x < 0) x = -x;
  if (x <= 8.0) return t2_chbevl(x * 0.5 - 2.0, t2_i0e_A, 30);
  return t2_fdiv(t2_chbevl(t2_fdiv(32.0, x) - 2.0, t2_i0e_B, 25), t2_sqrt_core(x));
}

// Four Chebyshev series at once (the four objective values of one forward-difference evaluation, same echo): four
// independent recurrences are the instruction-level parallelism a lane needs, one scalar-loaded coefficient feeds all
// four.  `chunks` x 5 coefficients; the coefficients of the next chunk are requested before the current chunk is
// worked through.  Same operations in the same order as t2_chbevl: the recurrence is entered one step earlier with
// b0 = b1 = 0, whose first step yields b0 = c[0] exactly for a finite argument.
T2_HD void t2_chbevl4(const double* z, const double* tab, int chunks, double* out) {
  double b0[4], b1[4], b2[4];
  T2_UNROLL
  for (int j = 0; j < 4; ++j) { b0[j] = 0.0; b1[j] = 0.0; b2[j] = 0.0; }
  double cn[5];
  T2_UNROLL
  for (int q = 0; q < 5; ++q) cn[q] = tab[q];
  T2_NOUNROLL
  for (int it = 0; it < chunks; ++it) {
    double cc[5];
    T2_UNROLL
    for (int q = 0; q < 5; ++q) cc[q] = cn[q];
    if (it + 1 < chunks) {
      T2_UNROLL
      for (int q = 0; q < 5; ++q) cn[q] = tab[(it + 1) * 5 + q];
    }
    T2_UNROLL
    for (int q = 0; q < 5; ++q) {
      T2_UNROLL
      for (int j = 0; j < 4; ++j) {
        b2[j] = b1[j];
        b1[j] = b0[j];
        b0[j] = T2_CLENSHAW(z[j], b1[j], b2[j], cc[q]);
      }
    }
  }
  T2_UNROLL
  for (int j = 0; j < 4; ++j) out[j] = 0.5 * (b0[j] - b2[j]);
}

// The (8, inf) table behind five zeros: entered with b0 = b1 = 0 the recurrence stays at +0 through the five padding
// steps (z * 0 - 0 + 0) and arrives at b0 = B[0] exactly, so both series take 30 steps and can share one loop.
T2_TABLE double t2_i0e_Bp[30] = {
    0.0, 0.0, 0.0, 0.0, 0.0,
    -7.23318048787475395456E-18, -4.83050448594418207126E-18, 4.46562142029675999901E-17,
    3.46122286769746109310E-17,  -2.82762398051658348494E-16, -3.42548561967721913462E-16,
    1.77256013305652638360E-15,  3.81168066935262242075E-15,  -9.55484669882830764870E-15,
    -4.15056934728722208663E-14, 1.54008621752140982691E-14,  3.85277838274214270114E-13,
    7.18012445138366623367E-13,  -1.79417853150680611778E-12, -1.32158118404477131188E-11,
    -3.14991652796324136454E-11, 1.18891471078464383424E-11,  4.94060238822496958910E-10,
    3.39623202570838634515E-9,   2.26666899049817806459E-8,   2.04891858946906374183E-7,
    2.89137052083475648297E-6,   6.88975834691682398426E-5,   3.36911647825569408990E-3,
    8.04490411014108831608E-1};

// i0e of four arguments of ONE lane that all need the same series (`lane_small`: all four <= 8, else all four > 8),
// in a wave whose lanes differ: one 30-step loop, each lane's coefficient picked from the two scalar-loaded tables.
// On the synthetic brain volumes 61 % of the (lane, echo) evaluations of a Rician fit need the [0, 8] series and 39 %
// the other one, so practically every wave needs both: run one after the other (for all lanes) that is 55 steps of
// 12 float64 operations; here it is 30 steps of 12 + a select.  Every lane performs exactly the operations of its own
// series (the padding steps of t2_i0e_Bp leave +0): same bits as t2_i0e.
#ifndef T2_I0E_CHUNK
#define T2_I0E_CHUNK 5  // coefficients per trip of the shared loop (30 = 2 x 15 = 3 x 10 = 5 x 6 = 6 x 5 = 10 x 3)
#endif
// `near` (wave-uniform; the caller's guarantee that ax[1..3] lie within 2^-20, relative, of ax[0]): the reciprocals of the
// (8, inf) series -- 32 / x before the recurrence, the division by sqrt(x) after it -- come from ONE reciprocal square root
// (t2_sqrt_core_h of ax[0], a coupled Newton step to each of the others): h = 1 / (2 sqrt(x)) to 2^-50 gives 1 / x = (2h)^2
// and 1 / sqrt(x) = 2h as the multipliers of the same quotient-and-residual steps, and sqrt(x) itself after the
// recurrence as x * 2h and one residual correction.  Only the four h stay live across the loop.
T2_HD void t2_i0e4_by_lane(const double* ax, bool lane_small, double* r, bool near = false) {
  constexpr int K = T2_I0E_CHUNK;
  static_assert(30 % K == 0, "the shared loop walks both 30-entry tables in whole chunks");
  double z[4], b0[4], b1[4], b2[4], hh[4];
  if (near) {
    (void)t2_sqrt_core_h(ax[0], hh[0]);
    T2_UNROLL
    for (int j = 1; j < 4; ++j) hh[j] = t2_rsqrt_half_near(ax[j], hh[0]);
    T2_UNROLL
    for (int j = 0; j < 4; ++j) {
      const double y = hh[j] + hh[j];
      const double za = ax[j] * 0.5 - 2.0, zb = t2_div_by_rcp(32.0, ax[j], y * y) - 2.0;
      z[j] = lane_small ? za : zb;
      b0[j] = 0.0; b1[j] = 0.0; b2[j] = 0.0;
    }
  } else {
    T2_UNROLL
    for (int j = 0; j < 4; ++j) {
      const double za = ax[j] * 0.5 - 2.0, zb = t2_fdiv(32.0, ax[j]) - 2.0;
      z[j] = lane_small ? za : zb;
      b0[j] = 0.0; b1[j] = 0.0; b2[j] = 0.0;
      hh[j] = 0.0;
    }
  }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(T2_I0E_SCALAR_COEFF)
  // Each lane fetches ITS series' coefficients with vector loads from its own table pointer (two addresses per wave: two
  // cache lines per load), a chunk ahead of their use.  Scalar loads of both tables and a per-lane pick cost six vector
  // instructions per coefficient -- a 64-bit select between two scalar-register pairs is two moves and a v_cndmask per half,
  // the constant bus feeds one scalar operand per instruction -- beside the twelve of the four recurrences
  // (-DT2_I0E_SCALAR_COEFF: that form).
  const double* tab = lane_small ? t2_i0e_A : t2_i0e_Bp;
  // two register sets for the coefficient chunks, taken in turn (the loads of the next chunk land where they will be read:
  // no copies), so the loop makes two chunks per trip
  static_assert((30 / K) % 2 == 0, "the shared loop takes the chunks two at a time");
  double c0[K], c1[K];
  T2_UNROLL
  for (int q = 0; q < K; ++q) c0[q] = tab[q];
  auto chunk = [&](const double* cc) {
    T2_UNROLL
    for (int q = 0; q < K; ++q) {
      T2_UNROLL
      for (int j = 0; j < 4; ++j) {
        b2[j] = b1[j];
        b1[j] = b0[j];
        b0[j] = T2_CLENSHAW(z[j], b1[j], b2[j], cc[q]);
      }
    }
  };
  T2_NOUNROLL
  for (int it = 0; it < 30 / K; it += 2) {
    T2_UNROLL
    for (int q = 0; q < K; ++q) c1[q] = tab[(it + 1) * K + q];
    chunk(c0);
    if (it + 2 < 30 / K) {
      T2_UNROLL
      for (int q = 0; q < K; ++q) c0[q] = tab[(it + 2) * K + q];
    }
    chunk(c1);
  }
#else
  double an[K], bn[K];
  T2_UNROLL
  for (int q = 0; q < K; ++q) { an[q] = t2_i0e_A[q]; bn[q] = t2_i0e_Bp[q]; }
  T2_NOUNROLL
  for (int it = 0; it < 30 / K; ++it) {
    double cc[K];
    T2_UNROLL
    for (int q = 0; q < K; ++q) cc[q] = lane_small ? an[q] : bn[q];
    if (it + 1 < 30 / K) {
      T2_UNROLL
      for (int q = 0; q < K; ++q) { an[q] = t2_i0e_A[(it + 1) * K + q]; bn[q] = t2_i0e_Bp[(it + 1) * K + q]; }
    }
    T2_UNROLL
    for (int q = 0; q < K; ++q) {
      T2_UNROLL
      for (int j = 0; j < 4; ++j) {
        b2[j] = b1[j];
        b1[j] = b0[j];
        b0[j] = T2_CLENSHAW(z[j], b1[j], b2[j], cc[q]);
      }
    }
  }
#endif
  if (near) {
    T2_UNROLL
    for (int j = 0; j < 4; ++j) {
      const double ra = 0.5 * (b0[j] - b2[j]);
      const double y = hh[j] + hh[j];
      const double rb = t2_div_by_rcp(ra, t2_sqrt_from_h(ax[j], hh[j]), y);
      r[j] = lane_small ? ra : rb;
    }
    return;
  }
  T2_UNROLL
  for (int j = 0; j < 4; ++j) {
    const double ra = 0.5 * (b0[j] - b2[j]);
    const double rb = t2_fdiv(ra, t2_sqrt_core(ax[j]));
    r[j] = lane_small ? ra : rb;
  }
}

// log() of i0e's values (positive normal numbers in (0, 1]) the way fdlibm's __ieee754_log computes it: argument reduction to
// [sqrt(2)/2, sqrt(2)), s = f / (2 + f), a degree-7 polynomial in s^2, the exponent times ln 2 in two pieces; error below
// 1 ulp, about 35 instructions.  The device library's log is a double-double evaluation of about 75 instructions, and the
// Rician likelihood takes four logs per echo and evaluation: 30 % of its evaluation.  Neither is the reference's log (numpy's)
// bit for bit -- log is one of the functions the one-ulp yardstick perturbs -- and measured on 20 000 voxels against the
// live oracle the leaner one agrees at least as often: T2 within 1 ms 98.31 % against 98.26 %, iteration count equal 99.37 %
// against 99.30 %, the numpy_legacy form 99.90 % both, the stable sets of the twelve rician fixtures 0 of 809 rows off both
// (profiles/r03_exp12_lean_log.txt); kernel 18.07 -> 16.20 ms at 256 x 256 x 180 x 6 TE.  -DT2_OCML_LOG restores the library call.
T2_HD double t2_log_lean(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  int e = __builtin_amdgcn_frexp_exp(x);           // x = m 2^e, m in [0.5, 1)
  double m = __builtin_amdgcn_frexp_mant(x);
#else
  int e;
  double m = frexp(x, &e);
#endif
  const bool low = m < 0x1.6a09e667f3bcdp-1;       // sqrt(2) / 2
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double f = m - 1.0, k = (double)e;
  const double s = t2_fdiv(f, 2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                            6.666666666666735130e-01);
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return fma(k, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, k * 1.90821492927058770002e-10)) - f));
}
#if defined(T2_OCML_LOG)
#define T2_LOG_I0E(v) t2_log(v)
#else
#define T2_LOG_I0E(v) t2_log_lean(v)
#endif

// log(i0e(x)) for four arguments (the four points of one forward-difference evaluation at one echo: they differ by
// 1e-8 relative, so a lane's four arguments need the same series except within 1e-8 of the boundary 8).  Branches are
// wave-uniform.  A wave whose lanes all need the same series runs that series with scalar coefficients; a wave whose
// lanes differ (practically always, see above) runs the shared 30-step loop; a wave in which some LANE has arguments on
// both sides of 8 falls back to both series for everybody and a per-argument pick.  Whatever the path, every value is
// produced by the operations of t2_i0e for its argument: the paths give identical bits.  `near`: see t2_i0e4_by_lane.
T2_HD void t2_log_i0e4(const double* x, double* out, bool near = false) {
  double ax[4], z[4], ra[4], rb[4], r[4];
  bool small[4];
  T2_UNROLL
  for (int j = 0; j < 4; ++j) {
    ax[j] = x[j] < 0 ? -x[j] : x[j];
    small[j] = ax[j] <= 8.0;
    ra[j] = 0.0;
    rb[j] = 0.0;
  }
  const bool any_small = small[0] || small[1] || small[2] || small[3];
  const bool any_large = !(small[0] && small[1] && small[2] && small[3]);
#if defined(T2_I0E_STATS) && !defined(__HIPCC__)  // host-side statistics build only (how often does a lane need which series?)
  ++g_calls; g_small += any_small; g_large += any_large;
#endif
  const bool wave_small = T2_WAVE_ANY(any_small), wave_large = T2_WAVE_ANY(any_large);
  if (wave_small && wave_large && !T2_WAVE_ANY(any_small && any_large)) {
    t2_i0e4_by_lane(ax, any_small, r, near);
  } else {
    if (wave_small) {
      T2_UNROLL
      for (int j = 0; j < 4; ++j) z[j] = ax[j] * 0.5 - 2.0;
      t2_chbevl4(z, t2_i0e_A, 6, ra);
    }
    if (wave_large) {
      T2_UNROLL
      for (int j = 0; j < 4; ++j) z[j] = t2_fdiv(32.0, ax[j]) - 2.0;
      t2_chbevl4(z, t2_i0e_B, 5, rb);
      T2_UNROLL
      for (int j = 0; j < 4; ++j) rb[j] = t2_fdiv(rb[j], t2_sqrt_core(ax[j]));
    }
    T2_UNROLL
    for (int j = 0; j < 4; ++j) r[j] = small[j] ? ra[j] : rb[j];
  }
  T2_UNROLL
  for (int j = 0; j < 4; ++j) out[j] = T2_LOG_I0E(r[j]);
}

// ---- objective values exactly in the reference's operation order (float64) ---------------------
// `y` holds the samples as the objective sees them: when cfg.norm they have already been divided by
// the voxel's maximum in float32 (run_t2mapping.py:237-238), see prepare_samples().
struct ObjCtx {
  const LaneParams* P;
  EchoView y;
  double* trace = nullptr;  // optional: (k, T2, sigma, f) after each iteration, 4 doubles each
  int trace_cap = 0;
  int* trace_n = nullptr;
#if defined(T2_PHASE_STAMPS)
  unsigned long long* diag = nullptr;  // diagnostic build: this wave's block counters (t2fit_lbfgsb.h T2_BLK_END)
#endif
  T2_HD float sample(int i) const { return y[i]; }
};

// (the objectives themselves live with the solver that evaluates them: t2fit_lbfgsb.h ObjTerm / eval(),
// t2fit_lm.h lm_eval())

T2_HD int n_params(int model) { return model == T2FIT_MODEL_GAUSSIAN ? 2 : 3; }

// Per-voxel bounds (run_t2mapping.py:243-245).  Returns false when some lb > ub.
T2_HD bool lane_bounds(const LaneParams& P, float y0_raw, double* lb, double* ub) {
  for (int j = 0; j < 3; ++j) {
    lb[j] = P.lb[j];
    ub[j] = P.ub[j];
  }
  if (P.no_prior) {
    lb[0] = (double)y0_raw;
    ub[0] = P.np_k_ub;
    lb[1] = P.np_t2_lb;
    ub[1] = P.np_t2_ub;
  }
  const int np = n_params(P.model);
  bool ok = true;
  for (int j = 0; j < np; ++j) ok = ok && !(lb[j] > ub[j]);
  return ok;
}

// exp(x) for the residual map, where the float64 prediction is rounded to float32 straight away
// (utils/t2map_utils.py:77-80 stores it into a float32 array): round-to-nearest range reduction by
// ln 2 in two pieces, degree-11 Taylor polynomial on |r| <= 0.347 (truncation 6e-15 relative), scale by
// 2^n.  A few ulp of float64 instead of the library's < 1 ulp at about a third of its instructions; the
// float32 rounding of the prediction flips for about one sample in 10^7.  NaN propagates; arguments
// below -746 (including -inf) give 0.  Not used inside any solver.
T2_HD double t2_exp_res(double x, const double* cc) {
  const double n = nearbyint(x * 1.4426950408889634);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = cc[0];
#if defined(T2_DEVICE_COMPILE)
#pragma unroll
#endif
  for (int j = 1; j < 10; ++j) p = fma(p, r, cc[j]);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double nc = fmin(fmax(n, -2000.0), 2000.0);  // keeps the int conversion defined (NaN -> -2000, p is NaN then)
  return x < -746.0 ? 0.0 : ldexp(p, (int)nc);  // exp(-inf) = 0 (a T2 of 0 ms handed to the residual entry point)
}
inline void t2_exp_res_coefficients(double* cc) {  // 1/11!, 1/10!, ..., 1/2!
  double f = 1.0;
  for (int j = 2; j <= 11; ++j) { f *= j; cc[11 - j] = 1.0 / f; }
}

// Mean signed residual from the float32 maps (utils/t2map_utils.py:62-89): float64 prediction
// stored as float32, float32 residuals, numpy's pairwise float32 row sum, divided by nTE.
// This is the arithmetic of numpy >= 2 (NEP 50: the np.float64 echo time promotes the prediction to float64), which
// the default fixtures were generated with.  Under the numpy 1.26 the reference freezes, value-based casting turns the
// echo time into a float32 and the whole prediction is float32 (utils/t2map_utils.py:74-80; about 1e-4 absolute
// difference): cfg.numpy_legacy selects that form (fixtures tests/golden/frozen_voxels_*.npz).
T2_HD float residual_mean(const ObjCtx& c, float k32, float t232, float s32) {
  const int n = c.P->n_te;
  const bool gauss = c.P->model == T2FIT_MODEL_GAUSSIAN;
  const bool legacy = c.P->numpy_legacy != 0;
  const double k = (double)k32, t2 = (double)t232;
  const double k2 = (double)(k32 * k32), s2 = (double)(s32 * s32);  // float32 squares, as numpy
  const double rt2 = 1.0 / t2;
  auto resid = [&](int i) {
    const double te = c.P->te[i];
    if (legacy) {  // float32 throughout: k * exp(float32(-te) / t2), sqrt(k^2 * exp(float32(-2 te) / t2) + s^2)
      const float pf = gauss ? k32 * expf((float)(-te) / t232)
                             : sqrtf((k32 * k32) * expf((float)(-2.0 * te) / t232) + s32 * s32);
      return c.sample(i) - pf;
    }
    double pred;
    if (gauss) pred = k * t2_exp_res(t2_div_by_rcp(-te, t2, rt2), c.P->exp_c);
    else pred = t2_sqrt(k2 * t2_exp_res(t2_div_by_rcp(-2.0 * te, t2, rt2), c.P->exp_c) + s2);
    return c.sample(i) - (float)pred;
  };
  // numpy float32 add.reduce over a contiguous row: n < 8 sequential from 0; otherwise eight
  // interleaved partial sums over the first n - n%8 items, combined as a balanced tree, then the
  // remaining items added one by one.
  if (n == 8) {  // branch-free block: the eight exp() chains interleave and share their constants
    float r[8];
#if defined(T2_DEVICE_COMPILE)
#pragma unroll
#endif
    for (int i = 0; i < 8; ++i) r[i] = resid(i);
    return (((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))) / 8.0f;
  }
  if (n < 8) {  // sequential sum from 0, as numpy does below eight items
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) sum += resid(i);
    return sum / (float)n;
  }
  // longer trains: r8 is indexed through selects so it stays in registers
  float r8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float sum = 0.0f;
  const int n8 = n & ~7;
  for (int i = 0; i < n; ++i) {
    const float r = resid(i);
    if (i >= n8) {
      if (i == n8) sum = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
      sum += r;
    } else {
      const int slot = i & 7;
#if defined(T2_DEVICE_COMPILE)
#pragma unroll
#endif
      for (int j = 0; j < 8; ++j) r8[j] = (j == slot) ? r8[j] + r : r8[j];
    }
  }
  if (n8 == n) sum = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
  return sum / (float)n;
}

}  // namespace t2fit
