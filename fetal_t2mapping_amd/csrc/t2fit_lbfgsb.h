// t2fit_lbfgsb.h -- per-lane restatement of the reference's optimiser.
//
// The reference fits each voxel with scipy.optimize.minimize(method="L-BFGS-B", jac=False)
// (run_t2mapping.py:260-286).  scipy is a third-party dependency that is not part of the reference
// tree (pinned 1.11.3, requirements_frozen.txt:144); this file restates its published algorithm for
// the 2- and 3-parameter problems of this path so that a lane reproduces the reference's iterates,
// stop decisions and therefore its (deliberately early-stopped) maps:
//
//   * driver and stop rules: scipy _minimize_lbfgsb (x0 clipped into the bounds; factr = ftol/eps;
//     pgtol = gtol; maxiter / maxfun checked after every iteration)
//   * gradient: scipy approx_derivative('2-point', abs_step=1e-8, bounds=...): forward difference
//     with the step flipped or shortened at a bound, dx recomputed as (x+h)-x
//   * L-BFGS-B 3.0 (Byrd, Lu, Nocedal, Zhu 1995; Zhu, Byrd, Lu, Nocedal 1997; Morales, Nocedal
//     2011): projected-gradient test, generalized Cauchy point, subspace minimisation with the
//     projected-Newton refinement and backtracking, More'-Thuente line search (dcsrch/dcstep,
//     ftol 1e-3, gtol 0.9, xtol 0.1, <= maxls evaluations), curvature test s'y > eps*(-g's),
//     theta = y'y / s'y, m = 10 correction pairs.
//
// What is different from the library code, and why: with n <= 3 variables the limited-memory
// matrix B = theta*I - W M W' is formed explicitly (n x n) by applying the stored pairs as BFGS
// updates to theta*I (Byrd-Nocedal-Schnabel 1994, thm 2.3: identical in exact arithmetic), so the
// 2m x 2m middle matrices of the general code never exist and the whole state of a lane -- 10 pairs
// of 3-vectors -- stays in registers.  Consequently the library's numerical-breakdown restarts
// (Cholesky failure inside formk/formt) are replaced by a positive-definiteness check of the reduced
// n x n system.  Everything is float64, as in the reference.
//
// Arithmetic contract (the library is compiled with -ffp-contract=off, so nothing is fused behind the source's back
// and a restructured kernel gives the same bits): the objective (eval, objective_t) performs numpy's operations
// one by one -- a product and the sum it feeds are two roundings, as in the reference -- and so do the parts that
// restate library code statement by statement (projected gradient, the forward-difference quotient, dcsrch /
// dcstep, the step bound and trial point of lnsrlb; scipy's wheels are built for baseline x86-64: no fused
// multiply-add).  fma() is written out where this file has its own formulation anyway: the explicit-B recursion
// and the Cauchy / subspace steps on it, and the correctly rounded quotient / exp / sqrt sequences of t2fit_lane.h.
#pragma once

#include <type_traits>

#include "t2fit_lane.h"

// Diagnostic build (-DT2_PHASE_STAMPS) only: wave-level cost of each block of the lane solver.  T2_BLK_END adds, for
// block i, the shader cycles since the matching T2_BLK_T0, the number of lanes that were active in it and one
// entry to three counters of the wave's own LDS block (c.diag); the kernel adds them up over the grid at exit.
#if defined(T2_PHASE_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define T2_BLK_T0(name) const unsigned long long name = __builtin_amdgcn_s_memtime();
#define T2_BLK_END(c, i, t0) t2_blk_end((c).diag, i, t0);
__device__ __forceinline__ void t2_blk_end(unsigned long long* dg, int i, unsigned long long t0) {
  const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0;
  const unsigned long long ex = __ballot(true);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)ex) - 1) {
    dg[3 * i] += dt;
    dg[3 * i + 1] += (unsigned long long)__popcll(ex);
    dg[3 * i + 2] += 1ull;
  }
}
#else
#define T2_BLK_T0(name)
#define T2_BLK_END(c, i, t0)
#endif

namespace t2fit {

// ---- objective with numpy's summation order --------------------------------------------------
// np.sum over a contiguous float64 vector: sequential for n < 8, otherwise eight interleaved
// partial sums combined as a balanced tree plus a sequential tail (numpy pairwise_sum).
template <int MODEL> struct ObjTerm;
// np.log(signal) - np.log(sigma**2) of run_t2mapping.py:169: a float32 array minus a float64 scalar.  numpy >= 2
// (NEP 50) promotes to float64; numpy < 2 -- the reference freezes 1.26.0, requirements_frozen.txt:103 -- casts the
// scalar to float32 and subtracts in float32 (cfg.numpy_legacy).  With the forward-difference step of 1e-8 the
// float32 form almost never sees sigma move: the two trajectories differ on most voxels.
T2_HD double rician_log_term(float log_y, double log_s2, bool legacy) {
  return legacy ? (double)(log_y - (float)log_s2) : (double)log_y - log_s2;
}
template <> struct ObjTerm<T2FIT_MODEL_GAUSSIAN> {
  double k, t2;
  T2_HD void set(const ObjCtx&, const double* x) { k = x[0]; t2 = x[1]; }
  T2_HD double at(const ObjCtx& c, int i) const {
    const double r = (double)c.sample(i) - k * t2_exp_core(t2_fdiv(-c.P->te[i], t2));
    return r * r;
  }
  T2_HD double finish(const ObjCtx& c, double s) const { return s / c.P->n_te; }
};
template <> struct ObjTerm<T2FIT_MODEL_GAUSSIAN_RICIAN> {
  double k2, t2, s2;
  T2_HD void set(const ObjCtx&, const double* x) { k2 = x[0] * x[0]; t2 = x[1]; s2 = x[2] * x[2]; }
  T2_HD double at(const ObjCtx& c, int i) const {
    const double r = (double)c.sample(i) - t2_sqrt_core(k2 * t2_exp_core(t2_fdiv(-2.0 * c.P->te[i], t2)) + s2);
    return r * r;
  }
  T2_HD double finish(const ObjCtx& c, double s) const { return s / c.P->n_te; }
};
template <> struct ObjTerm<T2FIT_MODEL_RICIAN> {
  double k, t2, s2, ls2;
  bool legacy;
  T2_HD void set(const ObjCtx& c, const double* x) {
    k = x[0]; t2 = x[1]; s2 = x[2] * x[2]; ls2 = t2_log(s2);
    legacy = c.P->numpy_legacy != 0;
  }
  T2_HD double at(const ObjCtx& c, int i) const {
    const float yf = c.sample(i);
    const double m = k * t2_exp_core(t2_fdiv(-c.P->te[i], t2));
    const double xx = t2_fdiv(m * (double)yf, s2);
    const double a = rician_log_term(logf(yf), ls2, legacy);
    const double b = t2_fdiv((double)(yf * yf) + m * m, 2.0 * s2);
    const double d = (xx < 0 ? -xx : xx) + T2_LOG_I0E(t2_i0e(xx));
    return (a - b) + d;
  }
  T2_HD double finish(const ObjCtx&, double s) const { return -s; }
};

template <int MODEL>
T2_HD double objective_t(const ObjCtx& c, const double* x) {
  ObjTerm<MODEL> t;
  t.set(c, x);
  const int n = c.P->n_te;
  double s;
  if (n < 8) {
    s = 0.0;
    for (int i = 0; i < n; ++i) s += t.at(c, i);
  } else {
    double r[8];
    T2_UNROLL
    for (int j = 0; j < 8; ++j) r[j] = t.at(c, j);
    int i = 8;
    for (; i + 8 <= n; i += 8) {
      T2_UNROLL
      for (int j = 0; j < 8; ++j) r[j] += t.at(c, i + j);
    }
    s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) s += t.at(c, i);
  }
  return t.finish(c, s);
}

// ---- More'-Thuente line search (MINPACK-2 dcsrch / dcstep) ------------------------------------
enum { LS_START = 0, LS_FG = 1, LS_CONV = 2, LS_WARN = 3, LS_ERROR = 4 };

struct LsState {
  int task, stage;
  bool brackt;
  double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
};

// min / max / abs of this solver: one instruction each (v_max_f64 / v_min_f64, |x| as an operand modifier) instead of a
// compare and two selects.  They differ from the generic a > b ? a : b only when an operand is NaN (the non-NaN one
// is returned) or a zero's sign matters; neither occurs between two evaluations of a finite objective.
T2_HD double lb_max(double a, double b) { return __builtin_fmax(a, b); }
T2_HD double lb_min(double a, double b) { return __builtin_fmin(a, b); }
T2_HD double lb_abs(double a) { return __builtin_fabs(a); }
T2_HD double t2_max3(double a, double b, double c) { return lb_max(a, lb_max(b, c)); }

// One safeguarded cubic / secant step (MINPACK-2 dcstep).  The four cases of the original share
// their cubic-interpolation arithmetic, so it is computed once on operands selected by case: the
// lanes of a wave usually sit in different cases, and four inlined copies of the divisions and the
// square root would otherwise run one after the other.  Every value is formed by the same
// operations, in the same order, as in the case-by-case form (3*(fp-fy)/(sty-stp) and
// 3*(fy-fp)/(stp-sty) are the same floating-point number).  There is no branch in it: every choice is
// a value select, so that the caller's other work (the new-iterate bookkeeping of the lanes whose line
// search has ended) can be scheduled into the gaps of this long dependent chain.
T2_HD void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
                  double fp, double dp, bool& brackt, double stpmin, double stpmax) {
  const double sgnd = !(dx < 0.0 || dx > 0.0) ? (double)NAN : (dx < 0.0 ? -dp : dp);  // dp * (dx / |dx|): the factor is exactly +-1
  const bool c1 = fp > fx;                              // higher function value: minimum bracketed
  const bool c2 = !c1 && sgnd < 0.0;                    // derivatives of opposite sign: bracketed
  const bool c3 = !c1 && !c2 && lb_abs(dp) < lb_abs(dx);  // derivative magnitude decreases
  const bool c4 = !c1 && !c2 && !c3;                    // derivative does not decrease
  // cubic through (sta, fa, da) and (stp, fp, dp): a = y in case 4, a = x otherwise
  const double sta = c4 ? sty : stx, fa = c4 ? fy : fx, da = c4 ? dy : dx;
  const double theta = t2_fdiv(3.0 * (fa - fp), stp - sta) + da + dp;
  const double s = t2_max3(lb_abs(theta), lb_abs(da), lb_abs(dp));
  const double rs = t2_rcp_for_div(s);  // three quotients by s share its reciprocal
  const double ts = t2_div_by_rcp(theta, s, rs);
  double arg = ts * ts - t2_div_by_rcp(da, s, rs) * t2_div_by_rcp(dp, s, rs);
  arg = c3 ? lb_max(0.0, arg) : arg;
  double gamma = s * t2_sqrt_core(arg);
  const bool flip = c1 ? stp < stx : (c4 ? stp > sty : stp > stx);
  gamma = flip ? -gamma : gamma;
  const double gd1 = c1 ? dx : dp;  // the slope subtracted from gamma
  const double p = (gamma - gd1) + theta;
  const double q = c3 ? (gamma + (dx - dp)) + gamma : ((gamma - gd1) + gamma) + (c1 ? dp : (c2 ? dx : dy));
  const double r = t2_fdiv(p, q);
  const double far_end = stp > stx ? stpmax : stpmin;
  // cubic step
  double stpc = c1 ? stx + r * (stp - stx) : stp + r * (sta - stp);
  stpc = (c3 && !(r < 0.0 && gamma != 0.0)) ? far_end : stpc;
  // quadratic (case 1) or secant (cases 2, 3) step: one division chain on operands selected by case
  //   case 1: ((dx / ((fx - fp) / (stp - stx) + dx)) / 2) ; otherwise dp / (dp - dx)
  const double qden = c1 ? t2_fdiv(fx - fp, stp - stx) + dx : dp - dx;
  const double qq = t2_fdiv(c1 ? dx : dp, qden);
  const double quad = c1 ? qq * 0.5 : qq;
  const double stpq = c1 ? stx + quad * (stp - stx) : stp + quad * (stx - stp);
  // the step taken
  const double dc = lb_abs(stpc - stp), dq = lb_abs(stpq - stp);
  const double f1 = lb_abs(stpc - stx) < lb_abs(stpq - stx) ? stpc : stpc + (stpq - stpc) * 0.5;
  const double farther = dc > dq ? stpc : stpq, nearer = dc < dq ? stpc : stpq;
  const double lim = stp + 0.66 * (sty - stp);
  const double f3b = stp > stx ? lb_min(lim, nearer) : lb_max(lim, nearer);
  const double f3u = lb_max(stpmin, lb_min(stpmax, farther));
  const double f4 = brackt ? stpc : far_end;
  const double stpf = c1 ? f1 : (c2 ? farther : (c3 ? (brackt ? f3b : f3u) : f4));
  brackt = brackt || c1 || c2;
  // interval update
  const bool swap = !c1 && sgnd < 0.0;
  const double nsty = c1 ? stp : (swap ? stx : sty);
  const double nfy = c1 ? fp : (swap ? fx : fy);
  const double ndy = c1 ? dp : (swap ? dx : dy);
  const double nstx = c1 ? stx : stp;
  const double nfx = c1 ? fx : fp;
  const double ndx = c1 ? dx : dp;
  sty = nsty; fy = nfy; dy = ndy;
  stx = nstx; fx = nfx; dx = ndx;
  stp = stpf;
}

// START call of dcsrch: validates the first step and initialises the search state.
T2_HD void dcsrch_start(double f, double g, double stp, double ftol, double stpmin, double stpmax, LsState& s) {
  const double xtrapu = 4.0;
  const bool bad = stp < stpmin || stp > stpmax || g >= 0.0;  // (the state below is dead after an error)
  s.brackt = false;
  s.stage = 1;
  s.finit = f; s.ginit = g; s.gtest = ftol * s.ginit;
  s.width = stpmax - stpmin; s.width1 = s.width * 2.0;  // width / 0.5
  s.stx = 0.0; s.fx = s.finit; s.gx = s.ginit;
  s.sty = 0.0; s.fy = s.finit; s.gy = s.ginit;
  s.stmin = 0.0; s.stmax = stp + xtrapu * stp;
  s.task = bad ? LS_ERROR : LS_FG;
}

// Every later call: f, g are the objective and directional derivative at the trial step stp.  Two halves, so
// that the caller can tell the cheap stop tests (every lane, every evaluation) from the safeguarded
// step (only lanes whose search continues): dcsrch_tests() returns the task; if it is LS_FG,
// dcsrch_update() computes the next trial step.
T2_HD int dcsrch_tests(double f, double g, double stp, double gtol, double xtol, double stpmin, double stpmax,
                       LsState& s) {
  const double ftest = s.finit + stp * s.gtest;
  if (s.stage == 1 && f <= ftest && g >= 0.0) s.stage = 2;
  int task = LS_FG;
  if (s.brackt && (stp <= s.stmin || stp >= s.stmax)) task = LS_WARN;   // rounding errors prevent progress
  if (s.brackt && s.stmax - s.stmin <= xtol * s.stmax) task = LS_WARN;  // xtol test satisfied
  if (stp == stpmax && f <= ftest && g <= s.gtest) task = LS_WARN;      // stp = stpmax
  if (stp == stpmin && (f > ftest || g >= s.gtest)) task = LS_WARN;     // stp = stpmin
  if (f <= ftest && lb_abs(g) <= gtol * (-s.ginit)) task = LS_CONV;
  s.task = task;
  return task;
}

T2_HD void dcsrch_update(double f, double g, double& stp, double xtol, double stpmin, double stpmax, LsState& s) {
  const double xtrapl = 1.1, xtrapu = 4.0, p5 = 0.5, p66 = 0.66;
  const double ftest = s.finit + stp * s.gtest;
  // stage 1 works on the modified function psi(stp) = f(stp) - f(0) - ftol*stp*f'(0)
  const bool modified = s.stage == 1 && f <= s.fx && f > ftest;
  const double gt = modified ? s.gtest : 0.0;
  double fm = modified ? f - stp * gt : f, gm = modified ? g - gt : g;
  double fxm = modified ? s.fx - s.stx * gt : s.fx, fym = modified ? s.fy - s.sty * gt : s.fy;
  double gxm = modified ? s.gx - gt : s.gx, gym = modified ? s.gy - gt : s.gy;
  dcstep(s.stx, fxm, gxm, s.sty, fym, gym, stp, fm, gm, s.brackt, s.stmin, s.stmax);
  s.fx = modified ? fxm + s.stx * gt : fxm;
  s.fy = modified ? fym + s.sty * gt : fym;
  s.gx = modified ? gxm + gt : gxm;
  s.gy = modified ? gym + gt : gym;
  const double span = lb_abs(s.sty - s.stx);
  const double mid = s.stx + p5 * (s.sty - s.stx);
  stp = (s.brackt && span >= p66 * s.width1) ? mid : stp;
  s.width1 = s.brackt ? s.width : s.width1;
  s.width = s.brackt ? span : s.width;
  const double lo_b = lb_min(s.stx, s.sty), hi_b = lb_max(s.stx, s.sty);
  const double lo_u = stp + xtrapl * (stp - s.stx), hi_u = stp + xtrapu * (stp - s.stx);
  s.stmin = s.brackt ? lo_b : lo_u;
  s.stmax = s.brackt ? hi_b : hi_u;
  stp = lb_max(stp, stpmin);
  stp = lb_min(stp, stpmax);
  const bool stuck = s.brackt && ((stp <= s.stmin || stp >= s.stmax) || (s.stmax - s.stmin <= xtol * s.stmax));
  stp = stuck ? s.stx : stp;
  s.task = LS_FG;
}

T2_HD void dcsrch(double f, double g, double& stp, double gtol, double xtol, double stpmin, double stpmax,
                  LsState& s) {
  if (dcsrch_tests(f, g, stp, gtol, xtol, stpmin, stpmax, s) == LS_FG) dcsrch_update(f, g, stp, xtol, stpmin, stpmax, s);
}

// numpy's float64 add.reduce order for a vector of n < 16 items, fed in index order with the index
// of the first eight items known at compile time: n < 8 is a plain left-to-right sum from 0;
// otherwise ((t0+t1)+(t2+t3)) + ((t4+t5)+(t6+t7)) and then t8.. one by one.  Both forms are carried
// (two adds per item) so that no branch on n or on the item index is needed while accumulating.
struct NpSum {
  double seq = 0.0, p0 = 0.0, p1 = 0.0, p2 = 0.0, s = 0.0;
  template <int J> T2_HD void add(double t) {
    seq = J == 0 ? 0.0 + t : seq + t;
    if constexpr (J == 0) p0 = t;
    else if constexpr (J == 1) p0 += t;
    else if constexpr (J == 2) p1 = t;
    else if constexpr (J == 3) { p1 += t; p0 += p1; }
    else if constexpr (J == 4) p1 = t;
    else if constexpr (J == 5) p1 += t;
    else if constexpr (J == 6) p2 = t;
    else { p2 += t; p1 += p2; s = p0 + p1; }
  }
  T2_HD void tail(double t) { s += t; }
  T2_HD double total(int n) const { return n < 8 ? seq : s; }
};


// Four row sums (the four objective values of one forward-difference evaluation) in numpy's float64 add.reduce order,
// fed one echo at a time from a run-time loop over the echoes: the echo index is wave-uniform, so what to do with an
// item is decided by scalar branches.  NTE in 1..7: the plain left-to-right sum; NTE == 8: the balanced tree over
// pairs, quads and halves with three carried values per sum; NTE == 0 (run-time n, 2..32): the general pairwise form
// -- below eight items left to right, otherwise eight interleaved partial sums over the first n - n % 8 items, combined
// as a balanced tree, then the remaining items one by one.
template <int NTE> struct RowSums4 {
  static constexpr bool kGeneral = NTE == 0, kTree = NTE == 8;
  double s[kTree ? 1 : 4];              // left-to-right sum; general form: tree total + tail
  double p[kTree ? 4 : 1], q[kTree ? 4 : 1], h[kTree ? 4 : 1];  // NTE == 8: pair, quad, first half
  double r[kGeneral ? 8 : 1][4];        // general form: the eight interleaved partial sums
  T2_HD void init() {
    if constexpr (!kTree) {
      T2_UNROLL
      for (int j = 0; j < 4; ++j) s[j] = 0.0;
    }
    if constexpr (kGeneral) {
      T2_UNROLL
      for (int q8 = 0; q8 < 8; ++q8)
        T2_UNROLL
        for (int j = 0; j < 4; ++j) r[q8][j] = 0.0;
    }
  }
  T2_HD void add(int i, int n, const double* t) {
    if constexpr (kTree) {
      T2_UNROLL
      for (int j = 0; j < 4; ++j) {
        if ((i & 1) == 0) { p[j] = t[j]; continue; }
        p[j] += t[j];
        if ((i & 2) == 0) { q[j] = p[j]; continue; }
        q[j] += p[j];
        if ((i & 4) == 0) h[j] = q[j];
      }
    } else if constexpr (kGeneral) {
      const int n8 = n & ~7;
      if (n < 8 || i >= n8) {
        T2_UNROLL
        for (int j = 0; j < 4; ++j) s[j] += t[j];
        return;
      }
      // (every slot is rewritten, the one in turn with its new value: written as eight guarded stores the optimiser
      // turns the guard into an index and the array goes to scratch memory)
      const int slot = i & 7;
      static_for<0, 8>([&](auto SC) {
        constexpr int S = decltype(SC)::value;
        T2_UNROLL
        for (int j = 0; j < 4; ++j) {
          const double nv = i < 8 ? t[j] : r[S][j] + t[j];
          r[S][j] = slot == S ? nv : r[S][j];
        }
      });
      if (i == n8 - 1) {
        T2_UNROLL
        for (int j = 0; j < 4; ++j) s[j] = ((r[0][j] + r[1][j]) + (r[2][j] + r[3][j])) + ((r[4][j] + r[5][j]) + (r[6][j] + r[7][j]));
      }
    } else {
      T2_UNROLL
      for (int j = 0; j < 4; ++j) s[j] += t[j];
    }
  }
  T2_HD double total(int j) const {
    if constexpr (kTree) return h[j] + q[j];
    else return s[j];
  }
};

// ---- the solver -----------------------------------------------------------------------------------
// Resumable (reverse-communication) form, like the library's own driver loop: the caller evaluates
// the objective and its forward-difference gradient at `x` with eval(), then calls advance(), which
// runs the solver up to the next point it needs evaluated (or to the end).  A persistent kernel can
// therefore keep the expensive eval() uniform across the lanes of a wave while every lane sits in a
// different phase (or a different voxel) of its own fit.
// NTE > 0 fixes the number of echoes at compile time (the kernels instantiate the common train lengths): the
// evaluation then is one straight-line block -- no per-echo `i < n` tests between the echoes, so their
// exp / sqrt chains interleave -- and only the summation order that numpy uses for that length is carried.
// GSPLIT (three parameters only): the last ratio of every pair's s lives in a second array (`ghist`, global memory in
// the kernels) instead of the ring in LDS.  Four instead of five doubles per pair are then in LDS, 320 instead of 400
// bytes per lane, and 160 KiB hold the rings of EIGHT one-wave workgroups instead of six: every SIMD of a CU
// interleaves two waves.  Where a number is kept does not change it: results are the same bit for bit.
template <int MODEL, int NTE = 0, bool GSPLIT = false>
struct Lbfgsb {
  static constexpr int N = MODEL == T2FIT_MODEL_GAUSSIAN ? 2 : 3;
  static constexpr int kNte = NTE;
  static constexpr int M = 10;
  static constexpr int PAIR = 2 * N - 1;  // doubles per correction pair (see load_s / store_s)
  static constexpr bool kSplit = GSPLIT && N == 3;
  static constexpr int PAIR_L = kSplit ? PAIR - 1 : PAIR;  // of which in the ring `hist`

  double lb[N], ub[N];              // box
  double x[N], g[N], f;             // current evaluation point, objective, FD gradient
  double z[N], d[N], t[N], r[N];    // line-search frame: target, direction, x_old, g_old
  double fold, gd, gdold, stp, stpmx, sbgnrm, theta;
  LsState ls;
  // correction pairs live outside the lane's registers: a ring of M slots of PAIR doubles (the ratios of s, then
  // y / sqrt(y's): see store_s), element e of slot q at hist[(q*PAIR + e) * hstride].  In the kernel that is LDS (one
  // column per lane, 400 B per lane for n = 3), which is what keeps the solver state within the VGPR budget.
  double* hist;
  double* ghist;  // kSplit: element q * gstride holds the last ratio of the pair in ring slot q
  int hstride, gstride, head;
  int iwhere[N];
  int col, nit, nfev, ifun;
  uint8_t status;
  bool first;
  bool wn_stale;  // the library's WN1 matrix would be out of date (see begin())
  // echo-count specialisations keep the voxel's samples in registers (eval() indexes them with constants only), so
  // the kernel needs no LDS for them: what decides how many waves a CU holds is the correction-pair ring alone
  float ys[NTE > 0 ? NTE : 1];
  T2_HD float smp(const ObjCtx& c, int i) const {
    if constexpr (NTE > 0) return ys[i];
    else return c.sample(i);
  }
#if defined(T2_PHASE_STAMPS)
  unsigned long long* diag = nullptr;  // diagnostic build: the wave's block counters, for the const helpers
#endif
#if !defined(__HIPCC__)
  int n_reset = 0;  // host simulator only (debugging aid): how often the correction memory was dropped
#define T2_COUNT_RESET() ++n_reset
#else
#define T2_COUNT_RESET()
#endif

  // scipy ScalarFunction.fun_and_grad with approx_derivative('2-point', abs_step=h, bounds): all N+1
  // objective values in one pass over the echoes.  Evaluations that share T2 share their exp()
  // (same inputs, same bits); the N+1 chains are independent, which is the instruction-level
  // parallelism a lane needs at one wave per SIMD.
  T2_HD void eval(const ObjCtx& c) {
    const LaneParams& P = *c.P;
    const int n = NTE > 0 ? NTE : P.n_te;
    double x1[N], dx[N];
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      double h = P.fd_step;
      if ((x[i] + h) - x[i] == 0.0)  // step lost to rounding: scipy falls back to sqrt(eps) relative
        h = 1.4901161193847656e-08 * (x[i] >= 0 ? 1.0 : -1.0) * lb_max(1.0, lb_abs(x[i]));
      const double lower = x[i] - lb[i], upper = ub[i] - x[i];
      const double xs = x[i] + h;
      const bool violated = xs < lb[i] || xs > ub[i];
      const bool fitting = lb_abs(h) <= lb_max(lower, upper);
      if (violated && fitting) h = -h;
      if (!fitting) h = upper >= lower ? upper : -lower;
      x1[i] = x[i] + h;
      dx[i] = x1[i] - x[i];
    }
    nfev += 1 + N;
    if (MODEL != T2FIT_MODEL_RICIAN && n >= 16) {  // long echo trains: one objective at a time (general pairwise summation)
      f = objective_t<MODEL>(c, x);
      T2_UNROLL
      for (int i = 0; i < N; ++i) {
        double xt[N];
        T2_UNROLL
        for (int j = 0; j < N; ++j) xt[j] = j == i ? x1[j] : x[j];
        g[i] = (objective_t<MODEL>(c, xt) - f) / dx[i];
      }
      return;
    }
    // First eight echoes fully unrolled (index known at compile time: straight-line code, the
    // LDS / scalar loads of all samples and echo times can be issued up front, the summation
    // order needs no dispatch); echoes 8..15, if any, in a short run-time loop.
    NpSum s0, s1, s2, s3;
    const double nd = (double)n, inv_n = P.inv_n;
    auto mean = [&](double sum) { return t2_div_by_rcp(sum, nd, inv_n); };  // sum / n
    if constexpr (MODEL == T2FIT_MODEL_GAUSSIAN) {
      const double k = x[0], t2 = x[1], kp = x1[0], t2p = x1[1];
      const double rt2 = t2_rcp_for_div(t2), rt2p = t2_rcp_for_div(t2p);  // one reciprocal per T2, shared by every echo
      auto body = [&](int i, auto add) {
        const double y = (double)smp(c, i), te = P.te[i];
        const double E = t2_exp_core(t2_div_by_rcp(-te, t2, rt2)), Ep = t2_exp_core(t2_div_by_rcp(-te, t2p, rt2p));
        const double r0 = y - k * E, r1 = y - kp * E, r2 = y - k * Ep;
        add(r0 * r0, r1 * r1, r2 * r2, 0.0);
      };
      static_for<0, 8>([&](auto JC) {
        constexpr int J = decltype(JC)::value;
        if (J < n) body(J, [&](double a, double b2, double c2, double) { s0.add<J>(a); s1.add<J>(b2); s2.add<J>(c2); });
      });
      for (int i = 8; i < n; ++i) body(i, [&](double a, double b2, double c2, double) { s0.tail(a); s1.tail(b2); s2.tail(c2); });
      f = mean(s0.total(n));
      g[0] = t2_fdiv(mean(s1.total(n)) - f, dx[0]);
      g[1] = t2_fdiv(mean(s2.total(n)) - f, dx[1]);
    } else if constexpr (MODEL == T2FIT_MODEL_GAUSSIAN_RICIAN) {
      const double k2 = x[0] * x[0], kp2 = x1[0] * x1[0], t2 = x[1], t2p = x1[1];
      const double sg2 = x[N - 1] * x[N - 1], sgp2 = x1[N - 1] * x1[N - 1];
      const double rt2 = t2_rcp_for_div(t2), rt2p = t2_rcp_for_div(t2p);
      // The four radicands of an echo lie within a few steps' worth of each other: when every displaced point is within
      // 2^-22 (relative, in the radicand) of the base point -- k and sigma directly, exp(-2 te / T2) through the largest echo time -- the
      // three displaced square roots start from the base one's reciprocal root (t2_sqrt_near: same bits, 6 instead of 11
      // instructions).  Decided once per evaluation for the whole wave; a lane at a degenerate point (a zero parameter,
      // a step that was replaced by the distance to a bound) sends the wave through the independent roots.
      // (2^-22: the coupled Newton step of t2_sqrt_near then lands 2^-45 from the root and the correction 2^-39 ulp from it;
      // the reference's steps are 1e-8 on k >= 550, sigma >= 2 and T2 >= 10 ms: 2^-36, 2^-28 and, at 300 ms, 2^-24)
      const double tol = 0x1p-22;
      const bool near = 2.0 * lb_abs(dx[0]) <= tol * lb_abs(x[0]) && 2.0 * lb_abs(dx[N - 1]) <= tol * lb_abs(x[N - 1]) &&
                        2.0 * P.te_max * lb_abs(dx[1]) <= tol * (t2 * t2p) && sg2 > 0.0;
      if (T2_WAVE_ANY(!near)) {
        // the rare way round: independent roots, one echo at a time in a real loop (compact code beside the straight-line
        // block below, whose registers it must not add to); same operations, same summation order (RowSums4)
        RowSums4<NTE> sums;
        sums.init();
        T2_NOUNROLL
        for (int i = 0; i < n; ++i) {
          float yf;
          if constexpr (NTE > 0) {  // register-resident samples: always read ys[0], rotate by one (see the Rician loop below)
            yf = ys[0];
            T2_UNROLL
            for (int j = 0; j + 1 < NTE; ++j) ys[j] = ys[j + 1];
            ys[NTE - 1] = yf;
          } else {
            yf = c.sample(i);
          }
          const double y = (double)yf, te = P.te[i];
          const double E = t2_exp_core(t2_div_by_rcp(-2.0 * te, t2, rt2)), Ep = t2_exp_core(t2_div_by_rcp(-2.0 * te, t2p, rt2p));
          const double r0 = y - t2_sqrt_core(k2 * E + sg2), r1 = y - t2_sqrt_core(kp2 * E + sg2);
          const double r2 = y - t2_sqrt_core(k2 * Ep + sg2), r3 = y - t2_sqrt_core(k2 * E + sgp2);
          const double tm[4] = {r0 * r0, r1 * r1, r2 * r2, r3 * r3};
          sums.add(i, n, tm);
        }
        f = mean(sums.total(0));
        g[0] = t2_fdiv(mean(sums.total(1)) - f, dx[0]);
        g[1] = t2_fdiv(mean(sums.total(2)) - f, dx[1]);
        g[N - 1] = t2_fdiv(mean(sums.total(3)) - f, dx[N - 1]);
        return;
      }
      auto body = [&](int i, auto add) {
        const double y = (double)smp(c, i), te = P.te[i];
        const double E = t2_exp_core(t2_div_by_rcp(-2.0 * te, t2, rt2)), Ep = t2_exp_core(t2_div_by_rcp(-2.0 * te, t2p, rt2p));
        double h;
        const double q0 = t2_sqrt_core_h(k2 * E + sg2, h);
        const double r0 = y - q0, r1 = y - t2_sqrt_near(kp2 * E + sg2, h);
        const double r2 = y - t2_sqrt_near(k2 * Ep + sg2, h), r3 = y - t2_sqrt_near(k2 * E + sgp2, h);
        add(r0 * r0, r1 * r1, r2 * r2, r3 * r3);
      };
      static_for<0, 8>([&](auto JC) {
        constexpr int J = decltype(JC)::value;
        if (J < n) body(J, [&](double a, double b2, double c2, double d2) { s0.add<J>(a); s1.add<J>(b2); s2.add<J>(c2); s3.add<J>(d2); });
      });
      for (int i = 8; i < n; ++i) body(i, [&](double a, double b2, double c2, double d2) { s0.tail(a); s1.tail(b2); s2.tail(c2); s3.tail(d2); });
      f = mean(s0.total(n));
      g[0] = t2_fdiv(mean(s1.total(n)) - f, dx[0]);
      g[1] = t2_fdiv(mean(s2.total(n)) - f, dx[1]);
      g[N - 1] = t2_fdiv(mean(s3.total(n)) - f, dx[N - 1]);
    } else {
      // Rician likelihood (run_t2mapping.py:157-177).  A real loop over the echoes, and inside it real loops over the
      // Chebyshev coefficients of i0e (t2_log_i0e4): the evaluation is a few KB of code whatever the echo count, and its
      // working set is what one echo needs.  With a compile-time echo count the samples live in registers, which a
      // run-time index cannot address: the loop always reads ys[0] and rotates the array by one (NTE moves per echo;
      // after NTE echoes it is back in order).
      const double k = x[0], kp = x1[0], t2 = x[1], t2p = x1[1];
      const double sg2 = x[N - 1] * x[N - 1], sgp2 = x1[N - 1] * x1[N - 1];
      const double ls2 = t2_log(sg2), lsp2 = t2_log(sgp2);
      const double rt2 = t2_rcp_for_div(t2), rt2p = t2_rcp_for_div(t2p);
      // both quotients of a term share one reciprocal per noise level (1 / (2 sigma^2) is half of 1 / sigma^2, exactly)
      const double rs2 = t2_rcp_for_div(sg2), rsp2 = t2_rcp_for_div(sgp2);
      const bool legacy = P.numpy_legacy != 0;
      // the four i0e arguments of an echo, (k E y / sigma^2 at the four points), within 2^-21 of each other: as for the square
      // roots of the least-squares model above, checked on the parameters once per evaluation, for the whole wave
      const double tol = 0x1p-22;
      const bool near_lane = 2.0 * lb_abs(dx[0]) <= tol * lb_abs(x[0]) && 2.0 * lb_abs(dx[N - 1]) <= tol * lb_abs(x[N - 1]) &&
                             2.0 * P.te_max * lb_abs(dx[1]) <= tol * (t2 * t2p);
      const bool near = !T2_WAVE_ANY(!near_lane);
      RowSums4<NTE> sums;
      sums.init();
      T2_NOUNROLL
      for (int i = 0; i < n; ++i) {
        float yf;
        if constexpr (NTE > 0) {
          yf = ys[0];
          T2_UNROLL
          for (int j = 0; j + 1 < NTE; ++j) ys[j] = ys[j + 1];
          ys[NTE - 1] = yf;
        } else {
          yf = c.sample(i);
        }
        const double te = P.te[i];
        const double E = t2_exp_core(t2_div_by_rcp(-te, t2, rt2)), Ep = t2_exp_core(t2_div_by_rcp(-te, t2p, rt2p));
        const double yd = (double)yf, y2 = (double)(yf * yf);
        const float ly = logf(yf);
        // the four points: (k, T2, sigma), (k + h, ..), (.., T2 + h, ..), (.., .., sigma + h)
        const double m[4] = {k * E, kp * E, k * Ep, k * E};
        double xx[4], li[4], tm[4];
        T2_UNROLL
        for (int j = 0; j < 4; ++j) xx[j] = t2_div_by_rcp(m[j] * yd, j == 3 ? sgp2 : sg2, j == 3 ? rsp2 : rs2);
        t2_log_i0e4(xx, li, near);
        T2_UNROLL
        for (int j = 0; j < 4; ++j) {
          const double a = rician_log_term(ly, j == 3 ? lsp2 : ls2, legacy);
          const double b = t2_div_by_rcp(y2 + m[j] * m[j], 2.0 * (j == 3 ? sgp2 : sg2), 0.5 * (j == 3 ? rsp2 : rs2));
          const double dd = (xx[j] < 0 ? -xx[j] : xx[j]) + li[j];
          tm[j] = (a - b) + dd;
        }
        sums.add(i, n, tm);
      }
      f = -sums.total(0);
      // (k + h) - k and (T2 + h) - T2 once more: two subtractions instead of two values carried through the loop
      g[0] = t2_fdiv(-sums.total(1) - f, kp - k);
      g[1] = t2_fdiv(-sums.total(2) - f, t2p - t2);
      g[N - 1] = t2_fdiv(-sums.total(3) - f, dx[N - 1]);
    }
  }

  T2_HD double projgr(const double* x, const double* g) const {
    double nrm = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      double gi = g[i];
      if (gi < 0.0) gi = lb_max(x[i] - ub[i], gi);
      else gi = lb_min(x[i] - lb[i], gi);
      nrm = lb_max(nrm, lb_abs(gi));
    }
    return nrm;
  }

  // Ring slot of pair number p (0 = oldest): PAIR = 2N - 1 doubles.  The BFGS update does not change when s is
  // scaled, so s is kept as a DIRECTION, s / s_0 = (1, s_1/s_0, s_2/s_0): only the N - 1 ratios are stored, then the N
  // components of y / sqrt(y's).  Ten pairs are 400 instead of 480 bytes per lane with three parameters -- what lets
  // a CU hold six waves instead of five -- and the implicit 1 saves a third of the products of B s.  A first component
  // that is zero (the variable sat at its bound during the step) or vanishingly small is replaced by
  // +-2^-400 max|s_i|: the direction moves by 2^-400, the ratios stay below 2^400 and their squares times B finite.
  T2_HD int slot_of(int p) const { return (head + p) % M; }
  T2_HD double& hratio(int q, int e) const {
    if constexpr (kSplit) {
      if (e == N - 2) return ghist[q * gstride];
    }
    return hist[(q * PAIR_L + e) * hstride];
  }
  T2_HD double& hy(int q, int i) const { return hist[(q * PAIR_L + (kSplit ? N - 2 : N - 1) + i) * hstride]; }
  T2_HD void load_s(int q, double* sv) const {  // as a vector (tests; build_b() reads the ratios directly)
    sv[0] = 1.0;
    T2_UNROLL
    for (int i = 1; i < N; ++i) sv[i] = hratio(q, i - 1);
  }
  T2_HD void store_s(int q, const double* sv) {
    double big = lb_abs(sv[0]);
    T2_UNROLL
    for (int i = 1; i < N; ++i) big = lb_max(big, lb_abs(sv[i]));
    const double least = big * 0x1p-400;  // s is not zero here (y's > 0)
    const double p0 = lb_abs(sv[0]) < least ? (sv[0] < 0.0 ? -least : least) : sv[0];
    const double r = t2_rcp_for_div(p0);
    T2_UNROLL
    for (int i = 1; i < N; ++i) hratio(q, i - 1) = t2_div_by_rcp(sv[i], p0, r);
  }

  // B = theta*I updated by the stored pairs, oldest first (BFGS recursion; B stays symmetric, so
  // only the upper triangle is computed).  The next pair is fetched from the ring while the
  // current one is applied, and the scaling is a reciprocal applied by multiplication: at one
  // wave per SIMD this block is latency-bound, and 18 IEEE divisions per pair dominated it.
  T2_HD void build_b(double (*B)[N]) const {
    // the recursion carries the upper triangle only (U[i][j], j >= i): half the loop-carried registers and
    // none of the mirror copies; the full matrix is written once, after the last pair
    double U[N][N];
    T2_UNROLL
    for (int i = 0; i < N; ++i)
      T2_UNROLL
      for (int j = 0; j < N; ++j) U[i][j] = i == j ? theta : 0.0;
    // two register sets for the pairs, taken in turn: pair p is worked on in set p & 1 while pair p + 1 is fetched into
    // the other one, whose last reader was pair p - 1 -- so the fetch can land where it will be read and no copy is
    // needed (one set and a landing buffer cost five 64-bit moves per pair).  rp[.][1..N-1]: the pair's ratios
    // s_i / s_0 (rp[.][0] unused: it is the implicit 1)
    double rp[2][N], yp[2][N];
    T2_UNROLL
    for (int i = 0; i < N; ++i) { rp[0][i] = 0.0; yp[0][i] = 0.0; rp[1][i] = 0.0; yp[1][i] = 0.0; }
    int q = head;  // ring slot of pair p, stepped along instead of (head + p) % M per pair
    if (col > 0) {
      T2_UNROLL
      for (int i = 1; i < N; ++i) rp[0][i] = hratio(q, i - 1);
      T2_UNROLL
      for (int i = 0; i < N; ++i) yp[0][i] = hy(q, i);
    }
    static_for<0, M>([&](auto PC) {
      constexpr int p = decltype(PC)::value, cur = p & 1, nxt = cur ^ 1;
      if (p < col) {
        // (with the whole ring in LDS the slot is stepped past the last pair too: that fetch reads a slot nobody uses and
        // nobody uses what it returns, two instructions less per pair; with a global part it would pull a line that was
        // never written all the way from HBM, and the next pair's wait for it costs the 2.4 % the register sets bring:
        // profiles/r03_exp21_step_and_digest.txt)
        const int step = q + 1 == M ? 0 : q + 1;
        const int qn = (!kSplit || p + 1 < col) ? step : q;
        q = qn;
        T2_UNROLL
        for (int i = 1; i < N; ++i) rp[nxt][i] = hratio(qn, i - 1);
        T2_UNROLL
        for (int i = 0; i < N; ++i) yp[nxt][i] = hy(qn, i);
        // B s for s = (1, rp[1], rp[2]): the first column of B plus the ratios times the others
        double bs[N];
        T2_UNROLL
        for (int i = 0; i < N; ++i) {
          double a = U[0][i];  // = U[i][0] (symmetric; only j >= i is kept)
          T2_UNROLL
          for (int j = 1; j < N; ++j) a = fma(j >= i ? U[i][j] : U[j][i], rp[cur][j], a);
          bs[i] = a;
        }
        double sbs = bs[0];
        T2_UNROLL
        for (int i = 1; i < N; ++i) sbs = fma(rp[cur][i], bs[i], sbs);
        const double rsbs = t2_fast_rcp(sbs);
        double tb[N];
        T2_UNROLL
        for (int i = 0; i < N; ++i) tb[i] = bs[i] * rsbs;
        // (the ring holds y / sqrt(y's): the rank-one term y y' / (y's) needs no scaling here -- digest())
        T2_UNROLL
        for (int i = 0; i < N; ++i)
          T2_UNROLL
          for (int j = i; j < N; ++j) U[i][j] = fma(yp[cur][i], yp[cur][j], fma(-bs[i], tb[j], U[i][j]));
      }
    });
    T2_UNROLL
    for (int i = 0; i < N; ++i)
      T2_UNROLL
      for (int j = 0; j < N; ++j) B[i][j] = j >= i ? U[i][j] : U[j][i];
  }

  // Generalized Cauchy point along the projected steepest-descent path.
  T2_HD void cauchy(const double* x, const double* g, const double (*B)[N], double theta, double sbgnrm,
                    int* iwhere, double* xcp) const {
    const double epsmch = 2.220446049250313e-16;
    T2_UNROLL
    for (int i = 0; i < N; ++i) xcp[i] = x[i];
    // (no early exits in here either -- see begin_pass(): a lane with nothing to do computes along and keeps nothing)
    const bool live = sbgnrm > 0.0;
    double d[N], tbk[N], zfix[N];
    bool hasbk[N];
    int nbreak = 0;
    double f1 = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      const double neggi = -g[i];
      const double tl = x[i] - lb[i], tu = ub[i] - x[i];
      {
        const bool xlower = tl <= 0.0, xupper = tu <= 0.0;
        const int iw = xlower ? (neggi <= 0.0 ? 1 : 0) : (xupper ? (neggi >= 0.0 ? 2 : 0) : (lb_abs(neggi) <= 0.0 ? -3 : 0));
        iwhere[i] = (live && iwhere[i] != 3) ? iw : iwhere[i];
      }
      d[i] = 0.0; tbk[i] = 0.0; zfix[i] = 0.0; hasbk[i] = false;
      if (iwhere[i] == 0) {
        d[i] = neggi;
        f1 = fma(-neggi, neggi, f1);
        if (neggi != 0.0) {  // one division on the selected numerator instead of one per branch
          tbk[i] = t2_fdiv(neggi < 0.0 ? tl : tu, lb_abs(neggi));
          hasbk[i] = true;
          ++nbreak;
        }
      }
    }
    const bool start = live && nbreak > 0;  // nbreak == 0: every moving variable is box-bounded here, so d == 0
    const double f2_org = -theta * f1;
    auto dBd = [&](const double* dd) {
      double s = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i)
        T2_UNROLL
        for (int j = 0; j < N; ++j) s = fma(dd[i] * B[i][j], dd[j], s);
      return s;
    };
    double f2 = dBd(d);
    double dtm = t2_fdiv(-f1, f2);
    double tsum = 0.0, tj = 0.0;
    int nleft = nbreak;
    bool all_fixed = false;
    // at most N breakpoints: N copies of the segment step instead of a loop (a back-edge here costs scalar registers
    // across the whole begin-iteration block); `go` turns false where the loop would have been left
    bool go = start;
    T2_UNROLL
    for (int seg = 0; seg < N; ++seg) {
      if (!(go && nleft > 0)) continue;
      // next breakpoint: one-hot flags rather than an index, so that no local array is ever
      // indexed by a run-time value (the compiler would move it to scratch memory)
      bool pick[N];
      bool any = false;
      double tmin = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i) {
        pick[i] = hasbk[i] && (!any || tbk[i] < tmin);
        if (pick[i]) {
          T2_UNROLL
          for (int j = 0; j < N; ++j)
            if (j < i) pick[j] = false;
          any = true;
          tmin = tbk[i];
        }
      }
      const double tj0 = tj;
      tj = tmin;
      const double dt = tj - tj0;
      if (dtm < dt) { go = false; continue; }
      T2_BLK_T0(t_bp)
      T2_BLK_END((*this), 1, t_bp)
      tsum += dt;
      --nleft;
      double dibp = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i)
        if (pick[i]) {
          dibp = d[i];
          d[i] = 0.0;
          hasbk[i] = false;
          if (dibp > 0.0) { zfix[i] = ub[i] - x[i]; xcp[i] = ub[i]; iwhere[i] = 2; }
          else { zfix[i] = lb[i] - x[i]; xcp[i] = lb[i]; iwhere[i] = 1; }
        }
      if (nleft == 0 && nbreak == N) { all_fixed = true; go = false; continue; }
      // derivatives of the quadratic model along the remaining direction, z = xcp - x so far
      double z[N];
      T2_UNROLL
      for (int i = 0; i < N; ++i) z[i] = d[i] != 0.0 ? tsum * d[i] : zfix[i];
      f1 = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i) {
        double bz = 0.0;
        T2_UNROLL
        for (int j = 0; j < N; ++j) bz = fma(B[i][j], z[j], bz);
        f1 = fma(d[i], g[i] + bz, f1);
      }
      f2 = lb_max(epsmch * f2_org, dBd(d));
      if (nleft > 0) dtm = t2_fdiv(-f1, f2);
      else { f1 = 0.0; f2 = 0.0; dtm = 0.0; }  // all remaining variables are box-bounded
    }
    const bool commit = start && !all_fixed;
    dtm = lb_max(dtm, 0.0);
    tsum += dtm;
    T2_UNROLL
    for (int i = 0; i < N; ++i) xcp[i] = commit ? fma(tsum, d[i], xcp[i]) : xcp[i];
  }

  // Direct primal subspace minimisation over the variables free at the Cauchy point, followed by
  // the L-BFGS-B 3.0 projection / backtracking safeguard.  Returns false if the reduced matrix is
  // not positive definite.  z holds xcp on entry and the subspace minimiser on exit.
  T2_HD bool subsm(const double* x, const double* g, const double (*B)[N], const int* iwhere, double* z) const {
    bool fr[N];
    double r[N], du[N];
    int nfree = 0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) { fr[i] = iwhere[i] <= 0; nfree += fr[i]; du[i] = 0.0; }
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      double bz = 0.0;
      T2_UNROLL
      for (int j = 0; j < N; ++j) bz = fma(B[i][j], z[j] - x[j], bz);
      r[i] = fr[i] ? -(g[i] + bz) : 0.0;
    }
    // reduced system A du = r with A = B on free rows/cols, identity elsewhere; LDL^T, pivots must be > 0
    double A[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    T2_UNROLL
    for (int i = 0; i < N; ++i)
      T2_UNROLL
      for (int j = 0; j < N; ++j)
        if (fr[i] && fr[j]) A[i][j] = B[i][j];
    double rr[3] = {r[0], r[1], N == 3 ? r[N - 1] : 0.0};
    if (N == 2) rr[2] = 0.0;
    // (one reciprocal per pivot, applied by multiplication: six quotients of this block share three divisors)
    const double d0 = A[0][0], r0 = t2_fast_rcp(d0);
    const double l10 = A[1][0] * r0, l20 = A[2][0] * r0;
    const double d1 = fma(-l10, A[1][0], A[1][1]), r1 = t2_fast_rcp(d1);
    const double a21 = fma(-l20, A[1][0], A[2][1]);
    const double l21 = a21 * r1;
    const double d2 = fma(-l21, a21, fma(-l20, A[2][0], A[2][2])), r2 = t2_fast_rcp(d2);
    // (no early exit on a non-positive pivot: the step is computed anyway -- NaN or nonsense then, thrown away by the
    // caller, which begins the iteration again -- so that this is one straight block)
    const bool pos_def = d0 > 0.0 && d1 > 0.0 && d2 > 0.0;
    const double y0 = rr[0], y1 = fma(-l10, y0, rr[1]), y2 = fma(-l21, y1, fma(-l20, y0, rr[2]));
    const double u2 = y2 * r2, u1 = fma(-l21, u2, y1 * r1);
    const double u0 = fma(-l20, u2, fma(-l10, u1, y0 * r0));
    du[0] = u0; du[1] = u1;
    if (N == 3) du[N - 1] = u2;
    // projected Newton point
    double xp[N];
    bool projected = false;
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      xp[i] = z[i];
      if (fr[i]) {
        const double xk = lb_max(lb[i], z[i] + du[i]);
        z[i] = lb_min(ub[i], xk);
        if (z[i] == lb[i] || z[i] == ub[i]) projected = true;
      }
    }
    double ddp = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) ddp = fma(z[i] - x[i], g[i], ddp);
    if (projected && ddp > 0.0) {  // not a descent direction: backtrack along du from the Cauchy point instead
      T2_UNROLL
      for (int i = 0; i < N; ++i) z[i] = xp[i];
      double alpha = 1.0, temp1 = 1.0;
      bool hit[N];
      T2_UNROLL
      for (int i = 0; i < N; ++i) hit[i] = false;
      T2_UNROLL
      for (int i = 0; i < N; ++i) {
        if (fr[i]) {
          const double dk = du[i];
          if (dk < 0.0) {
            const double temp2 = lb[i] - z[i];
            if (temp2 >= 0.0) temp1 = 0.0;
            else if (dk * alpha < temp2) temp1 = t2_fdiv(temp2, dk);
          } else if (dk > 0.0) {
            const double temp2 = ub[i] - z[i];
            if (temp2 <= 0.0) temp1 = 0.0;
            else if (dk * alpha > temp2) temp1 = t2_fdiv(temp2, dk);
          }
          if (temp1 < alpha) {
            alpha = temp1;
            T2_UNROLL
            for (int j = 0; j < N; ++j) hit[j] = j == i;
          }
        }
      }
      if (alpha < 1.0) {
        T2_UNROLL
        for (int i = 0; i < N; ++i)
          if (hit[i]) {
            if (du[i] > 0.0) { z[i] = ub[i]; du[i] = 0.0; }
            else if (du[i] < 0.0) { z[i] = lb[i]; du[i] = 0.0; }
          }
      }
      T2_UNROLL
      for (int i = 0; i < N; ++i)
        if (fr[i]) z[i] = fma(alpha, du[i], z[i]);
    }
    return pos_def;
  }

  // Start a fit: x = x0 clipped into the box (scipy), empty memory.  Next: eval(), then advance().
  // `hist_` must hold M * PAIR_L doubles at stride `hstride_` (kSplit: and `ghist_` M doubles at stride `gstride_`).
  T2_HD void init(const double* x0_, const double* lb_, const double* ub_, double* hist_, int hstride_,
                  double* ghist_ = nullptr, int gstride_ = 0) {
    hist = hist_;
    hstride = hstride_;
    ghist = ghist_;
    gstride = gstride_;
    head = 0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      lb[i] = lb_[i];
      ub[i] = ub_[i];
      x[i] = t2_clip(x0_[i], lb[i], ub[i]);
      iwhere[i] = (ub[i] - lb[i] <= 0.0) ? 3 : 0;
    }
    col = 0; nit = 0; nfev = 0; ifun = 0;
    theta = 1.0;
    status = T2FIT_ST_NOT_CONV;
    first = true;
    wn_stale = false;
    // digest() computes the line-search branches for every lane, also after the first evaluation of a fit, when
    // there is no line search yet: give it defined values to chew on (its results are discarded there)
    T2_UNROLL
    for (int i = 0; i < N; ++i) { z[i] = 0.0; d[i] = 0.0; t[i] = 0.0; r[i] = 0.0; g[i] = 0.0; }
    f = 0.0; fold = 0.0; gd = 0.0; gdold = 0.0; stp = 1.0; stpmx = 1.0; sbgnrm = 0.0;
    ls.task = LS_START; ls.stage = 1; ls.brackt = false;
    ls.ginit = ls.gtest = ls.gx = ls.gy = ls.finit = ls.fx = ls.fy = 0.0;
    ls.stx = ls.sty = ls.stmin = ls.stmax = ls.width = ls.width1 = 0.0;
  }

  // Everything between two evaluations, in two halves.  digest() takes in the evaluation that just finished (line
  // search stop tests; then either the next safeguarded trial step, or a new iterate: stop tests and the
  // correction pair) and says what comes next; begin() (re)starts iterations -- B, Cauchy point, subspace step,
  // line-search set-up -- until one yields a trial point or the fit ends.  Control is arranged so that each
  // expensive block exists exactly once in the instruction stream, so the lanes of a wave that sit in different
  // phases serialise over little code.  advance() = both: true when the fit has ended, false when `x` holds the
  // next point to evaluate.
  enum { GO_BEGIN, GO_TRIAL, GO_FAIL, GO_DONE };

  T2_HD int digest(const ObjCtx& c) {
    const LaneParams& P = *c.P;
    const double epsmch = 2.220446049250313e-16;
    // The lanes of a wave arrive here in three states: first evaluation of a fit, line search continues (the
    // safeguarded step: a long chain of dependent divisions), line search ended (new iterate: stop tests and the
    // correction pair).  At one wave per SIMD a dependent float64 chain issues at half rate, so the three are
    // not branches: every lane computes all of them side by side in one block -- the compiler interleaves the
    // independent chains -- and then keeps, by value selects, the results of the state it is in.  (Arithmetic on
    // the state a lane is not in runs on stale or zero values and is thrown away; nothing here can trap.)
    T2_BLK_T0(t_dig)
    const bool was_first = first;
    first = false;
    // -- line search: directional derivative, stop tests --
    double gdn = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) gdn += g[i] * d[i];
    LsState lsn = ls;
    const int task = dcsrch_tests(f, gdn, stp, 0.9, 0.1, 0.0, stpmx, lsn);
    const bool cont = !was_first && task == LS_FG;   // line search continues
    const bool newit = !was_first && task != LS_FG;  // a new iterate
    // -- line search continues: next trial step --
    double stpn = stp;
    dcsrch_update(f, gdn, stpn, 0.1, 0.0, stpmx, lsn);
    // -- new iterate (or first evaluation): projected gradient, stop tests, correction pair --
    const double sb = projgr(x, g);
    const double tol = P.lbfgsb_tol;  // factr * epsmch = (ftol / epsmch) * epsmch, formed on the host
    const int nit1 = nit + 1;
    const bool out_of_budget = nit1 >= P.maxiter || nfev > P.maxfun;  // scipy: STOP, success False
    const bool converged = sb <= P.gtol || (!was_first && (fold - f) <= tol * t2_max3(lb_abs(fold), lb_abs(f), 1.0));
    double rn[N], dn[N], rr = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) { rn[i] = g[i] - r[i]; rr += rn[i] * rn[i]; }
    // (the library scales by stp only when stp != 1; multiplying by 1.0 is exact, so no test is needed)
    const double dr = (gdn - gdold) * stp, ddum = -gdold * stp;
    T2_UNROLL
    for (int i = 0; i < N; ++i) dn[i] = d[i] * stp;
    const double theta_n = t2_fdiv(rr, dr);
    // the pair is stored as (s, y / sqrt(y's)): B is rebuilt from the whole ring at every iteration, and this way the
    // rank-one term of each update is a plain outer product (three multiplications, a dot product and a reciprocal
    // less per pair and iteration); theta keeps the unscaled y
    const double rsy = t2_fast_rsqrt(dr);
    const bool iterate_on = newit && !out_of_budget && !converged;
    const bool store_pair = iterate_on && !(dr <= epsmch * ddum);  // else: curvature too small, skip the update
    // -- keep what belongs to this lane's state --
    if (!was_first) gd = gdn;
    if (cont) { ls = lsn; stp = stpn; ++ifun; }
    if (!cont && !was_first) ls.task = lsn.task;
    if (was_first || newit) sbgnrm = sb;
    if (newit) {
      nit = nit1;
      if (c.trace && *c.trace_n < c.trace_cap) {  // what scipy hands to the reference's callback (:180-234)
        double* tr = c.trace + 4 * (*c.trace_n)++;
        tr[0] = x[0]; tr[1] = x[1]; tr[2] = N == 3 ? x[N - 1] : 0.0; tr[3] = f;
      }
    }
    if (iterate_on) {
      T2_UNROLL
      for (int i = 0; i < N; ++i) { r[i] = rn[i]; d[i] = dn[i]; }
    }
    if (store_pair) {
      if (col == M) {  // ring is full: the oldest pair is dropped
        head = (head + 1) % M;
        col = M - 1;
      }
      const int q = slot_of(col);
      store_s(q, dn);
      T2_UNROLL
      for (int i = 0; i < N; ++i) hy(q, i) = rn[i] * rsy;
      ++col;
      theta = theta_n;
    }
    if ((was_first || newit) && converged && !(newit && out_of_budget)) status = T2FIT_ST_CONVERGED;
    int next;
    if (cont) next = ifun - 1 >= P.maxls ? GO_FAIL : GO_TRIAL;
    else if (was_first) next = converged ? GO_DONE : GO_BEGIN;
    else next = (out_of_budget || converged) ? GO_DONE : GO_BEGIN;
    T2_BLK_END(c, 0, t_dig)
    if (next == GO_TRIAL) set_trial();
    return next;
  }

  // x = the trial point of the running line search
  T2_HD void set_trial() {
    T2_UNROLL
    for (int i = 0; i < N; ++i) x[i] = stp == 1.0 ? z[i] : stp * d[i] + t[i];
  }

  // One pass of the begin-iteration work.  `next` is GO_BEGIN or GO_FAIL (what digest() or an earlier pass returned).
  // Returns GO_TRIAL (`x` holds the next point to evaluate), GO_DONE (the fit has ended), or GO_BEGIN / GO_FAIL: the
  // memory was dropped or the line search could not start, and the iteration has to be begun again -- by another
  // call.  There is deliberately no loop in here: the restarts are rare (none on a typical --prior volume, one per
  // 15 voxels under --no_prior), and a back-edge around B / Cauchy point / subspace step cost 7 % of the kernel in
  // scalar-register spills alone; the persistent kernel simply calls again in its next round.
  T2_HD int begin_pass(const ObjCtx& c, int next) {
    const LaneParams& P = *c.P;
    bool ended = false;
    if (next == GO_FAIL) {
      // line search could not be completed: back to the previous iterate.  With an empty memory
      // that is scipy's ABNORMAL termination (success False); otherwise drop the memory and redo.
      T2_UNROLL
      for (int i = 0; i < N; ++i) { x[i] = t[i]; g[i] = r[i]; }
      f = fold;
      ended = col == 0;  // (the rest of the pass still runs for this lane, on values nobody reads afterwards)
      col = 0; theta = 1.0; wn_stale = false;
      T2_COUNT_RESET();
    }
    double B[N][N];
    T2_BLK_T0(t_b)
    build_b(B);
    T2_BLK_END(c, 3, t_b)
    T2_BLK_T0(t_c)
    cauchy(x, g, B, theta, sbgnrm, iwhere, z);
    T2_BLK_END(c, 4, t_c)
    int nfree = 0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) nfree += iwhere[i] <= 0;
    // Two reasons to drop the memory and begin the iteration again.  (a) The reduced matrix of the subspace step is
    // not positive definite (numerical breakdown).  (b) The library's stale WN1: it keeps its 2m x 2m matrix WN1
    // (inner products of the pairs over the free and the active variables) up to date inside formk, which it only
    // calls when a subspace step is taken.  An iteration whose Cauchy point leaves no variable free skips formk, so
    // the changes of that iteration (the newest pair, the variables that left the free set) never reach WN1; at the
    // next iteration that takes a subspace step the factorisation of the then inconsistent matrix fails
    // ("nonpositive definiteness in Cholesky factorization in formk") and the library drops its memory and
    // restarts the iteration.  This is deterministic and frequent when bounds are active (5-8 % of the voxels of a
    // --no_prior fit; all 150 restarts the reference makes on the golden fixtures follow this rule, none is missed
    // by it), so it is part of the reference's trajectory and is restated here, without the matrix: a flag.
    bool restart = false;
    if (nfree != 0 && col != 0) {
      restart = wn_stale;
      if (!restart) {
        T2_BLK_T0(t_s)
        restart = !subsm(x, g, B, iwhere, z);
        T2_BLK_END(c, 5, t_s)
      }
    } else if (col != 0) {
      wn_stale = true;  // no variable is free at the Cauchy point: the library skips formk
    }
    if (restart) {
      wn_stale = false;
      col = 0; theta = 1.0;
      T2_COUNT_RESET();
    }
    T2_BLK_T0(t_l)
    // line search along d = z - x (lnsrlb)
    T2_UNROLL
    for (int i = 0; i < N; ++i) { d[i] = z[i] - x[i]; t[i] = x[i]; r[i] = g[i]; }
    stpmx = 1e10;
    T2_UNROLL
    for (int i = 0; i < N; ++i) {  // (selects, no branches; the quotient of a skipped variable is computed and dropped)
      const double a1 = d[i];
      // lower bound limits a decreasing variable, upper bound an increasing one
      const double a2 = (a1 < 0.0 ? lb[i] : ub[i]) - x[i];
      const bool at_bound = a1 < 0.0 ? a2 >= 0.0 : a2 <= 0.0;
      const bool limits = a1 < 0.0 ? a1 * stpmx < a2 : a1 * stpmx > a2;
      const double q = t2_fdiv(a2, a1);
      const double cand = at_bound ? 0.0 : (limits ? q : stpmx);
      stpmx = a1 != 0.0 ? cand : stpmx;
    }
    stpmx = nit == 0 ? 1.0 : stpmx;
    stp = 1.0;  // every variable is boxed, so the first step is not rescaled by 1/|d|
    fold = f;
    gd = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) gd += g[i] * d[i];
    gdold = gd;
    dcsrch_start(f, gd, stp, 1e-3, 0.0, stpmx, ls);  // ERROR also covers gd >= 0: not a descent direction
    ifun = 1;
    T2_BLK_END(c, 6, t_l)
    // what this pass amounts to, by value selects (no early exits above: the block is one straight line)
    const int out = ended ? GO_DONE : (restart ? GO_BEGIN : ((ls.task != LS_FG || ifun - 1 >= P.maxls) ? GO_FAIL : GO_TRIAL));
    if (out == GO_TRIAL) set_trial();
    return out;
  }

  // everything between two evaluations: true when the fit has ended, false when `x` holds the next point to evaluate
  T2_HD bool advance(const ObjCtx& c) {
    int next = digest(c);
    while (next == GO_BEGIN || next == GO_FAIL) next = begin_pass(c, next);
    return next == GO_DONE;
  }

  T2_HD void result(LaneResult& out) const {
    T2_UNROLL
    for (int i = 0; i < 3; ++i) out.x[i] = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) out.x[i] = x[i];
    out.fun = f;
    out.nit = nit;
    out.nfev = nfev;
    out.status = status;
  }
};

template <int MODEL>
T2_HD void lbfgsb_solve(const ObjCtx& c, const double* lb, const double* ub, LaneResult& out) {
  Lbfgsb<MODEL> s;
  double hist[Lbfgsb<MODEL>::M * Lbfgsb<MODEL>::PAIR] = {};
  s.init(c.P->x0, lb, ub, hist, 1);
  do {
    s.eval(c);
  } while (!s.advance(c));
  s.result(out);
}

}  // namespace t2fit
