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
#pragma once

#include <type_traits>

#include "t2fit_lane.h"

#if defined(__HIPCC__)
#define T2_UNROLL _Pragma("unroll")
#else
#define T2_UNROLL
#endif

namespace t2fit {

// ---- objective with numpy's summation order --------------------------------------------------
// np.sum over a contiguous float64 vector: sequential for n < 8, otherwise eight interleaved
// partial sums combined as a balanced tree plus a sequential tail (numpy pairwise_sum).
template <int MODEL> struct ObjTerm;
template <> struct ObjTerm<T2FIT_MODEL_GAUSSIAN> {
  double k, t2;
  T2_HD void set(const double* x) { k = x[0]; t2 = x[1]; }
  T2_HD double at(const ObjCtx& c, int i) const {
    const double r = (double)c.sample(i) - k * t2_exp_core(t2_fdiv(-c.P->te[i], t2));
    return r * r;
  }
  T2_HD double finish(const ObjCtx& c, double s) const { return s / c.P->n_te; }
};
template <> struct ObjTerm<T2FIT_MODEL_GAUSSIAN_RICIAN> {
  double k2, t2, s2;
  T2_HD void set(const double* x) { k2 = x[0] * x[0]; t2 = x[1]; s2 = x[2] * x[2]; }
  T2_HD double at(const ObjCtx& c, int i) const {
    const double r = (double)c.sample(i) - t2_sqrt_core(k2 * t2_exp_core(t2_fdiv(-2.0 * c.P->te[i], t2)) + s2);
    return r * r;
  }
  T2_HD double finish(const ObjCtx& c, double s) const { return s / c.P->n_te; }
};
template <> struct ObjTerm<T2FIT_MODEL_RICIAN> {
  double k, t2, s2, ls2;
  T2_HD void set(const double* x) { k = x[0]; t2 = x[1]; s2 = x[2] * x[2]; ls2 = t2_log(s2); }
  T2_HD double at(const ObjCtx& c, int i) const {
    const float yf = c.sample(i);
    const double m = k * t2_exp_core(t2_fdiv(-c.P->te[i], t2));
    const double xx = t2_fdiv(m * (double)yf, s2);
    const double a = (double)logf(yf) - ls2;
    const double b = t2_fdiv((double)(yf * yf) + m * m, 2.0 * s2);
    const double d = (xx < 0 ? -xx : xx) + t2_log(t2_i0e(xx));
    return (a - b) + d;
  }
  T2_HD double finish(const ObjCtx&, double s) const { return -s; }
};

template <int MODEL>
T2_HD double objective_t(const ObjCtx& c, const double* x) {
  ObjTerm<MODEL> t;
  t.set(x);
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

T2_HD double t2_max3(double a, double b, double c) { return t2_max(a, t2_max(b, c)); }

// One safeguarded cubic / secant step (MINPACK-2 dcstep).  The four cases of the original share
// their cubic-interpolation arithmetic, so it is computed once on operands selected by case: the
// lanes of a wave usually sit in different cases, and four inlined copies of the divisions and the
// square root would otherwise run one after the other.  Every value is formed by the same
// operations, in the same order, as in the case-by-case form (3*(fp-fy)/(sty-stp) and
// 3*(fy-fp)/(stp-sty) are the same floating-point number).
T2_HD void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
                  double fp, double dp, bool& brackt, double stpmin, double stpmax) {
  const double sgnd = !(dx < 0.0 || dx > 0.0) ? (double)NAN : (dx < 0.0 ? -dp : dp);  // dp * (dx / |dx|): the factor is exactly +-1
  const bool c1 = fp > fx;                              // higher function value: minimum bracketed
  const bool c2 = !c1 && sgnd < 0.0;                    // derivatives of opposite sign: bracketed
  const bool c3 = !c1 && !c2 && t2_abs(dp) < t2_abs(dx);  // derivative magnitude decreases
  const bool c4 = !c1 && !c2 && !c3;                    // derivative does not decrease
  // cubic through (sta, fa, da) and (stp, fp, dp): a = y in case 4, a = x otherwise
  const double sta = c4 ? sty : stx, fa = c4 ? fy : fx, da = c4 ? dy : dx;
  const double theta = t2_fdiv(3.0 * (fa - fp), stp - sta) + da + dp;
  const double s = t2_max3(t2_abs(theta), t2_abs(da), t2_abs(dp));
  const double ts = t2_fdiv(theta, s);
  double arg = ts * ts - t2_fdiv(da, s) * t2_fdiv(dp, s);
  if (c3) arg = t2_max(0.0, arg);
  double gamma = s * t2_sqrt_core(arg);
  const bool flip = c1 ? stp < stx : (c4 ? stp > sty : stp > stx);
  if (flip) gamma = -gamma;
  const double gd1 = c1 ? dx : dp;  // the slope subtracted from gamma
  const double p = (gamma - gd1) + theta;
  const double q = c3 ? (gamma + (dx - dp)) + gamma : ((gamma - gd1) + gamma) + (c1 ? dp : (c2 ? dx : dy));
  const double r = t2_fdiv(p, q);
  // cubic step
  double stpc = c1 ? stx + r * (stp - stx) : stp + r * (sta - stp);
  if (c3 && !(r < 0.0 && gamma != 0.0)) stpc = stp > stx ? stpmax : stpmin;
  // quadratic (case 1) or secant (cases 2, 3) step
  const double quad = c1 ? t2_fdiv(dx, t2_fdiv(fx - fp, stp - stx) + dx) * 0.5 : t2_fdiv(dp, dp - dx);
  const double stpq = c1 ? stx + quad * (stp - stx) : stp + quad * (stx - stp);
  double stpf;
  if (c1) {
    stpf = t2_abs(stpc - stx) < t2_abs(stpq - stx) ? stpc : stpc + (stpq - stpc) * 0.5;
  } else if (c2) {
    stpf = t2_abs(stpc - stp) > t2_abs(stpq - stp) ? stpc : stpq;
  } else if (c3) {
    if (brackt) {
      stpf = t2_abs(stpc - stp) < t2_abs(stpq - stp) ? stpc : stpq;
      stpf = stp > stx ? t2_min(stp + 0.66 * (sty - stp), stpf) : t2_max(stp + 0.66 * (sty - stp), stpf);
    } else {
      stpf = t2_abs(stpc - stp) > t2_abs(stpq - stp) ? stpc : stpq;
      stpf = t2_min(stpmax, stpf);
      stpf = t2_max(stpmin, stpf);
    }
  } else {
    stpf = brackt ? stpc : (stp > stx ? stpmax : stpmin);
  }
  if (c1 || c2) brackt = true;
  // interval update, written as value selects (conditional stores through the reference
  // parameters made the compiler keep the six values in scratch memory)
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
  if (stp < stpmin || stp > stpmax || g >= 0.0) { s.task = LS_ERROR; return; }
  s.brackt = false;
  s.stage = 1;
  s.finit = f; s.ginit = g; s.gtest = ftol * s.ginit;
  s.width = stpmax - stpmin; s.width1 = s.width * 2.0;  // width / 0.5
  s.stx = 0.0; s.fx = s.finit; s.gx = s.ginit;
  s.sty = 0.0; s.fy = s.finit; s.gy = s.ginit;
  s.stmin = 0.0; s.stmax = stp + xtrapu * stp;
  s.task = LS_FG;
}

// Every later call: f, g are the objective and directional derivative at the trial step stp.
T2_HD void dcsrch(double f, double g, double& stp, double gtol, double xtol, double stpmin, double stpmax,
                  LsState& s) {
  const double xtrapl = 1.1, xtrapu = 4.0, p5 = 0.5, p66 = 0.66;
  const double ftest = s.finit + stp * s.gtest;
  if (s.stage == 1 && f <= ftest && g >= 0.0) s.stage = 2;
  int task = LS_FG;
  if (s.brackt && (stp <= s.stmin || stp >= s.stmax)) task = LS_WARN;   // rounding errors prevent progress
  if (s.brackt && s.stmax - s.stmin <= xtol * s.stmax) task = LS_WARN;  // xtol test satisfied
  if (stp == stpmax && f <= ftest && g <= s.gtest) task = LS_WARN;      // stp = stpmax
  if (stp == stpmin && (f > ftest || g >= s.gtest)) task = LS_WARN;     // stp = stpmin
  if (f <= ftest && t2_abs(g) <= gtol * (-s.ginit)) task = LS_CONV;
  if (task != LS_FG) { s.task = task; return; }
  // stage 1 works on the modified function psi(stp) = f(stp) - f(0) - ftol*stp*f'(0)
  const bool modified = s.stage == 1 && f <= s.fx && f > ftest;
  double fm = f, gm = g, fxm = s.fx, fym = s.fy, gxm = s.gx, gym = s.gy;
  if (modified) {
    fm = f - stp * s.gtest;
    fxm = s.fx - s.stx * s.gtest;
    fym = s.fy - s.sty * s.gtest;
    gm = g - s.gtest;
    gxm = s.gx - s.gtest;
    gym = s.gy - s.gtest;
  }
  dcstep(s.stx, fxm, gxm, s.sty, fym, gym, stp, fm, gm, s.brackt, s.stmin, s.stmax);
  if (modified) {
    s.fx = fxm + s.stx * s.gtest;
    s.fy = fym + s.sty * s.gtest;
    s.gx = gxm + s.gtest;
    s.gy = gym + s.gtest;
  } else {
    s.fx = fxm; s.fy = fym; s.gx = gxm; s.gy = gym;
  }
  if (s.brackt) {
    if (t2_abs(s.sty - s.stx) >= p66 * s.width1) stp = s.stx + p5 * (s.sty - s.stx);
    s.width1 = s.width;
    s.width = t2_abs(s.sty - s.stx);
  }
  if (s.brackt) {
    s.stmin = t2_min(s.stx, s.sty);
    s.stmax = t2_max(s.stx, s.sty);
  } else {
    s.stmin = stp + xtrapl * (stp - s.stx);
    s.stmax = stp + xtrapu * (stp - s.stx);
  }
  stp = t2_max(stp, stpmin);
  stp = t2_min(stp, stpmax);
  if ((s.brackt && (stp <= s.stmin || stp >= s.stmax)) || (s.brackt && s.stmax - s.stmin <= xtol * s.stmax))
    stp = s.stx;
  s.task = LS_FG;
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


// ---- the solver -----------------------------------------------------------------------------------
// Resumable (reverse-communication) form, like the library's own driver loop: the caller evaluates
// the objective and its forward-difference gradient at `x` with eval(), then calls advance(), which
// runs the solver up to the next point it needs evaluated (or to the end).  A persistent kernel can
// therefore keep the expensive eval() uniform across the lanes of a wave while every lane sits in a
// different phase (or a different voxel) of its own fit.
template <int MODEL>
struct Lbfgsb {
  static constexpr int N = MODEL == T2FIT_MODEL_GAUSSIAN ? 2 : 3;
  static constexpr int M = 10;

  double lb[N], ub[N];              // box
  double x[N], g[N], f;             // current evaluation point, objective, FD gradient
  double z[N], d[N], t[N], r[N];    // line-search frame: target, direction, x_old, g_old
  double fold, gd, gdold, stp, stpmx, sbgnrm, theta;
  LsState ls;
  // correction pairs live outside the lane's registers: a ring of M slots of (s, y), element e of
  // slot q at hist[(q*2N + e) * hstride].  In the kernel that is LDS (one column per lane, 480 B per
  // lane for n = 3), which is what keeps the solver state within the VGPR budget.
  double* hist;
  int hstride, head;
  int iwhere[N];
  int col, nit, nfev, ifun;
  uint8_t status;
  bool first;
#if defined(T2_PHASE_STAMPS)
  unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0}, stamp_t = 0;  // diagnostic build: cycles per advance() block
#endif
#if defined(T2_PHASE_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define T2_LSTAMP0() stamp_t = __builtin_amdgcn_s_memtime();
#define T2_LSTAMP(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); stamp[i] += n_ - stamp_t; stamp_t = n_; }
#else
#define T2_LSTAMP0()
#define T2_LSTAMP(i)
#endif

  // scipy ScalarFunction.fun_and_grad with approx_derivative('2-point', abs_step=h, bounds): all N+1
  // objective values in one pass over the echoes.  Evaluations that share T2 share their exp()
  // (same inputs, same bits); the N+1 chains are independent, which is the instruction-level
  // parallelism a lane needs at one wave per SIMD.
  T2_HD void eval(const ObjCtx& c) {
    const LaneParams& P = *c.P;
    const int n = P.n_te;
    double x1[N], dx[N];
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      double h = P.fd_step;
      if ((x[i] + h) - x[i] == 0.0)  // step lost to rounding: scipy falls back to sqrt(eps) relative
        h = 1.4901161193847656e-08 * (x[i] >= 0 ? 1.0 : -1.0) * t2_max(1.0, t2_abs(x[i]));
      const double lower = x[i] - lb[i], upper = ub[i] - x[i];
      const double xs = x[i] + h;
      const bool violated = xs < lb[i] || xs > ub[i];
      const bool fitting = t2_abs(h) <= t2_max(lower, upper);
      if (violated && fitting) h = -h;
      if (!fitting) h = upper >= lower ? upper : -lower;
      x1[i] = x[i] + h;
      dx[i] = x1[i] - x[i];
    }
    nfev += 1 + N;
    if (n >= 16) {  // long echo trains: one objective at a time (general pairwise summation)
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
        const double y = (double)c.sample(i), te = P.te[i];
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
      auto body = [&](int i, auto add) {
        const double y = (double)c.sample(i), te = P.te[i];
        const double E = t2_exp_core(t2_div_by_rcp(-2.0 * te, t2, rt2)), Ep = t2_exp_core(t2_div_by_rcp(-2.0 * te, t2p, rt2p));
        const double r0 = y - t2_sqrt_core(k2 * E + sg2), r1 = y - t2_sqrt_core(kp2 * E + sg2);
        const double r2 = y - t2_sqrt_core(k2 * Ep + sg2), r3 = y - t2_sqrt_core(k2 * E + sgp2);
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
      const double k = x[0], kp = x1[0], t2 = x[1], t2p = x1[1];
      const double sg2 = x[N - 1] * x[N - 1], sgp2 = x1[N - 1] * x1[N - 1];
      const double ls2 = t2_log(sg2), lsp2 = t2_log(sgp2);
      const double rt2 = t2_rcp_for_div(t2), rt2p = t2_rcp_for_div(t2p);
      // both quotients of a term share one reciprocal per noise level (1 / (2 sigma^2) is half of 1 / sigma^2, exactly)
      const double rs2 = t2_rcp_for_div(sg2), rsp2 = t2_rcp_for_div(sgp2);
      auto term = [](double kk, double E, double s2v, double rs2v, double ls2v, float yf) {
        const double m = kk * E;
        const double xx = t2_div_by_rcp(m * (double)yf, s2v, rs2v);
        const double a = (double)logf(yf) - ls2v;
        const double b = t2_div_by_rcp((double)(yf * yf) + m * m, 2.0 * s2v, 0.5 * rs2v);
        const double dd = (xx < 0 ? -xx : xx) + t2_log(t2_i0e(xx));
        return (a - b) + dd;
      };
      auto body = [&](int i, auto add) {
        const float yf = c.sample(i);
        const double te = P.te[i];
        const double E = t2_exp_core(t2_div_by_rcp(-te, t2, rt2)), Ep = t2_exp_core(t2_div_by_rcp(-te, t2p, rt2p));
        add(term(k, E, sg2, rs2, ls2, yf), term(kp, E, sg2, rs2, ls2, yf), term(k, Ep, sg2, rs2, ls2, yf),
            term(k, E, sgp2, rsp2, lsp2, yf));
      };
      static_for<0, 8>([&](auto JC) {
        constexpr int J = decltype(JC)::value;
        if (J < n) body(J, [&](double a, double b2, double c2, double d2) { s0.add<J>(a); s1.add<J>(b2); s2.add<J>(c2); s3.add<J>(d2); });
      });
      for (int i = 8; i < n; ++i) body(i, [&](double a, double b2, double c2, double d2) { s0.tail(a); s1.tail(b2); s2.tail(c2); s3.tail(d2); });
      f = -s0.total(n);
      g[0] = t2_fdiv(-s1.total(n) - f, dx[0]);
      g[1] = t2_fdiv(-s2.total(n) - f, dx[1]);
      g[N - 1] = t2_fdiv(-s3.total(n) - f, dx[N - 1]);
    }
  }

  T2_HD double projgr(const double* x, const double* g) const {
    double nrm = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      double gi = g[i];
      if (gi < 0.0) gi = t2_max(x[i] - ub[i], gi);
      else gi = t2_min(x[i] - lb[i], gi);
      nrm = t2_max(nrm, t2_abs(gi));
    }
    return nrm;
  }

  T2_HD double& hs(int p, int i) const { return hist[((((head + p) % M) * 2) * N + i) * hstride]; }
  T2_HD double& hy(int p, int i) const { return hist[((((head + p) % M) * 2 + 1) * N + i) * hstride]; }

  // B = theta*I updated by the stored pairs, oldest first (BFGS recursion; B stays symmetric, so
  // only the upper triangle is computed).  The next pair is fetched from the ring while the
  // current one is applied, and the two scalings are reciprocals applied by multiplication: at one
  // wave per SIMD this block is latency-bound, and 18 IEEE divisions per pair dominated it.
  T2_HD void build_b(double (*B)[N]) const {
    T2_UNROLL
    for (int i = 0; i < N; ++i)
      T2_UNROLL
      for (int j = 0; j < N; ++j) B[i][j] = i == j ? theta : 0.0;
    double sp[N], yp[N];
    if (col > 0) {
      T2_UNROLL
      for (int i = 0; i < N; ++i) { sp[i] = hs(0, i); yp[i] = hy(0, i); }
    }
    for (int p = 0; p < col; ++p) {
      double sn[N], yn[N];
      const int pn = p + 1 < col ? p + 1 : p;
      T2_UNROLL
      for (int i = 0; i < N; ++i) { sn[i] = hs(pn, i); yn[i] = hy(pn, i); }
      double bs[N];
      double sbs = 0.0, ys = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i) {
        double a = 0.0;
        T2_UNROLL
        for (int j = 0; j < N; ++j) a += B[i][j] * sp[j];
        bs[i] = a;
      }
      T2_UNROLL
      for (int i = 0; i < N; ++i) { sbs += sp[i] * bs[i]; ys += yp[i] * sp[i]; }
      const double rys = t2_fast_rcp(ys), rsbs = t2_fast_rcp(sbs);
      double ty[N], tb[N];
      T2_UNROLL
      for (int i = 0; i < N; ++i) { ty[i] = yp[i] * rys; tb[i] = bs[i] * rsbs; }
      T2_UNROLL
      for (int i = 0; i < N; ++i)
        T2_UNROLL
        for (int j = i; j < N; ++j) {
          const double v = B[i][j] + (yp[i] * ty[j] - bs[i] * tb[j]);
          B[i][j] = v;
          B[j][i] = v;
        }
      T2_UNROLL
      for (int i = 0; i < N; ++i) { sp[i] = sn[i]; yp[i] = yn[i]; }
    }
  }

  // Generalized Cauchy point along the projected steepest-descent path.
  T2_HD void cauchy(const double* x, const double* g, const double (*B)[N], double theta, double sbgnrm,
                    int* iwhere, double* xcp) const {
    const double epsmch = 2.220446049250313e-16;
    T2_UNROLL
    for (int i = 0; i < N; ++i) xcp[i] = x[i];
    if (sbgnrm <= 0.0) return;
    double d[N], tbk[N], zfix[N];
    bool hasbk[N];
    int nbreak = 0;
    double f1 = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      const double neggi = -g[i];
      const double tl = x[i] - lb[i], tu = ub[i] - x[i];
      if (iwhere[i] != 3) {
        const bool xlower = tl <= 0.0, xupper = tu <= 0.0;
        iwhere[i] = 0;
        if (xlower) { if (neggi <= 0.0) iwhere[i] = 1; }
        else if (xupper) { if (neggi >= 0.0) iwhere[i] = 2; }
        else if (t2_abs(neggi) <= 0.0) iwhere[i] = -3;
      }
      d[i] = 0.0; tbk[i] = 0.0; zfix[i] = 0.0; hasbk[i] = false;
      if (iwhere[i] == 0) {
        d[i] = neggi;
        f1 -= neggi * neggi;
        if (neggi != 0.0) {  // one division on the selected numerator instead of one per branch
          tbk[i] = t2_fdiv(neggi < 0.0 ? tl : tu, t2_abs(neggi));
          hasbk[i] = true;
          ++nbreak;
        }
      }
    }
    if (nbreak == 0) return;  // every moving variable is box-bounded here, so d == 0
    const double f2_org = -theta * f1;
    auto dBd = [&](const double* dd) {
      double s = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i)
        T2_UNROLL
        for (int j = 0; j < N; ++j) s += dd[i] * B[i][j] * dd[j];
      return s;
    };
    double f2 = dBd(d);
    double dtm = t2_fdiv(-f1, f2);
    double tsum = 0.0, tj = 0.0;
    int nleft = nbreak;
    bool all_fixed = false;
    while (nleft > 0) {
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
      if (dtm < dt) break;
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
      if (nleft == 0 && nbreak == N) { all_fixed = true; break; }
      // derivatives of the quadratic model along the remaining direction, z = xcp - x so far
      double z[N];
      T2_UNROLL
      for (int i = 0; i < N; ++i) z[i] = d[i] != 0.0 ? tsum * d[i] : zfix[i];
      f1 = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i) {
        double bz = 0.0;
        T2_UNROLL
        for (int j = 0; j < N; ++j) bz += B[i][j] * z[j];
        f1 += d[i] * (g[i] + bz);
      }
      f2 = t2_max(epsmch * f2_org, dBd(d));
      if (nleft > 0) dtm = t2_fdiv(-f1, f2);
      else { f1 = 0.0; f2 = 0.0; dtm = 0.0; }  // all remaining variables are box-bounded
    }
    if (all_fixed) return;
    dtm = t2_max(dtm, 0.0);
    tsum += dtm;
    T2_UNROLL
    for (int i = 0; i < N; ++i) xcp[i] += tsum * d[i];
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
      for (int j = 0; j < N; ++j) bz += B[i][j] * (z[j] - x[j]);
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
    const double d0 = A[0][0];
    if (!(d0 > 0.0)) return false;
    const double l10 = t2_fdiv(A[1][0], d0), l20 = t2_fdiv(A[2][0], d0);
    const double d1 = A[1][1] - l10 * A[1][0];
    if (!(d1 > 0.0)) return false;
    const double l21 = t2_fdiv(A[2][1] - l20 * A[1][0], d1);
    const double d2 = A[2][2] - l20 * A[2][0] - l21 * (A[2][1] - l20 * A[1][0]);
    if (!(d2 > 0.0)) return false;
    const double y0 = rr[0], y1 = rr[1] - l10 * y0, y2 = rr[2] - l20 * y0 - l21 * y1;
    const double u2 = t2_fdiv(y2, d2), u1 = t2_fdiv(y1, d1) - l21 * u2, u0 = t2_fdiv(y0, d0) - l10 * u1 - l20 * u2;
    du[0] = u0; du[1] = u1;
    if (N == 3) du[N - 1] = u2;
    // projected Newton point
    double xp[N];
    bool projected = false;
    T2_UNROLL
    for (int i = 0; i < N; ++i) {
      xp[i] = z[i];
      if (fr[i]) {
        const double xk = t2_max(lb[i], z[i] + du[i]);
        z[i] = t2_min(ub[i], xk);
        if (z[i] == lb[i] || z[i] == ub[i]) projected = true;
      }
    }
    if (!projected) return true;
    double ddp = 0.0;
    T2_UNROLL
    for (int i = 0; i < N; ++i) ddp += (z[i] - x[i]) * g[i];
    if (ddp > 0.0) {  // not a descent direction: backtrack along du from the Cauchy point instead
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
        if (fr[i]) z[i] += alpha * du[i];
    }
    return true;
  }

  // Start a fit: x = x0 clipped into the box (scipy), empty memory.  Next: eval(), then advance().
  // `hist_` must hold 2*M*N doubles at stride `hstride_`.
  T2_HD void init(const double* x0_, const double* lb_, const double* ub_, double* hist_, int hstride_) {
    hist = hist_;
    hstride = hstride_;
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
  }

  // Everything between two evaluations.  Control is arranged so that each expensive block (the
  // end-of-iteration bookkeeping and the begin-iteration work: B, Cauchy point, subspace step,
  // line-search set-up) exists exactly once in the instruction stream, so the lanes of a wave that
  // sit in different phases serialise over little code.  Returns true when the fit has ended,
  // false when `x` holds the next point to evaluate.
  T2_HD bool advance(const ObjCtx& c) {
    const LaneParams& P = *c.P;
    const double epsmch = 2.220446049250313e-16;
    enum { GO_BEGIN, GO_TRIAL, GO_FAIL, GO_DONE };
    int next;
    T2_LSTAMP0()
    // ---- 1. digest the evaluation that just finished ----
    if (first) {
      first = false;
      sbgnrm = projgr(x, g);
      next = GO_BEGIN;
      if (sbgnrm <= P.gtol) { status = T2FIT_ST_CONVERGED; next = GO_DONE; }
    } else {
      gd = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i) gd += g[i] * d[i];
      dcsrch(f, gd, stp, 0.9, 0.1, 0.0, stpmx, ls);
      if (ls.task == LS_FG) {
        ++ifun;
        next = ifun - 1 >= P.maxls ? GO_FAIL : GO_TRIAL;
      } else {
        // ---- new iterate: stop tests, then the correction pair ----
        ++nit;
        if (c.trace && *c.trace_n < c.trace_cap) {  // what scipy hands to the reference's callback (:180-234)
          double* tr = c.trace + 4 * (*c.trace_n)++;
          tr[0] = x[0]; tr[1] = x[1]; tr[2] = N == 3 ? x[N - 1] : 0.0; tr[3] = f;
        }
        sbgnrm = projgr(x, g);
        const double tol = P.lbfgsb_tol;  // factr * epsmch = (ftol / epsmch) * epsmch, formed on the host
        if (nit >= P.maxiter || nfev > P.maxfun) {
          next = GO_DONE;  // scipy: STOP, success False
        } else if (sbgnrm <= P.gtol || (fold - f) <= tol * t2_max3(t2_abs(fold), t2_abs(f), 1.0)) {
          status = T2FIT_ST_CONVERGED;
          next = GO_DONE;
        } else {
          double rr = 0.0, dr, ddum;
          T2_UNROLL
          for (int i = 0; i < N; ++i) { r[i] = g[i] - r[i]; rr += r[i] * r[i]; }
          if (stp == 1.0) {
            dr = gd - gdold;
            ddum = -gdold;
          } else {
            dr = (gd - gdold) * stp;
            T2_UNROLL
            for (int i = 0; i < N; ++i) d[i] *= stp;
            ddum = -gdold * stp;
          }
          if (!(dr <= epsmch * ddum)) {  // else: curvature too small, skip the update
            if (col == M) {  // ring is full: the oldest pair is dropped
              head = (head + 1) % M;
              col = M - 1;
            }
            T2_UNROLL
            for (int i = 0; i < N; ++i) { hs(col, i) = d[i]; hy(col, i) = r[i]; }
            ++col;
            theta = t2_fdiv(rr, dr);
          }
          next = GO_BEGIN;
        }
      }
    }
    T2_LSTAMP(0)
    // ---- 2. (re)start iterations until one yields a trial point or the fit ends ----
    for (;;) {
      if (next == GO_FAIL) {
        // line search could not be completed: back to the previous iterate.  With an empty memory
        // that is scipy's ABNORMAL termination (success False); otherwise drop the memory and redo.
        T2_UNROLL
        for (int i = 0; i < N; ++i) { x[i] = t[i]; g[i] = r[i]; }
        f = fold;
        next = GO_DONE;
        if (col != 0) { col = 0; theta = 1.0; next = GO_BEGIN; }
      }
      if (next != GO_BEGIN) break;
      double B[N][N];
      T2_LSTAMP(5)
      build_b(B);
      T2_LSTAMP(1)
      cauchy(x, g, B, theta, sbgnrm, iwhere, z);
      T2_LSTAMP(2)
      int nfree = 0;
      T2_UNROLL
      for (int i = 0; i < N; ++i) nfree += iwhere[i] <= 0;
      if (nfree != 0 && col != 0) {
        if (!subsm(x, g, B, iwhere, z)) {  // numerical breakdown: drop the memory, redo the iteration
          col = 0; theta = 1.0;
          continue;
        }
      }
      T2_LSTAMP(3)
      // line search along d = z - x (lnsrlb)
      T2_UNROLL
      for (int i = 0; i < N; ++i) { d[i] = z[i] - x[i]; t[i] = x[i]; r[i] = g[i]; }
      stpmx = 1e10;
      if (nit == 0) {
        stpmx = 1.0;
      } else {
        T2_UNROLL
        for (int i = 0; i < N; ++i) {
          const double a1 = d[i];
          if (a1 != 0.0) {  // lower bound limits a decreasing variable, upper bound an increasing one
            const double a2 = (a1 < 0.0 ? lb[i] : ub[i]) - x[i];
            const bool at_bound = a1 < 0.0 ? a2 >= 0.0 : a2 <= 0.0;
            const bool limits = a1 < 0.0 ? a1 * stpmx < a2 : a1 * stpmx > a2;
            if (at_bound) stpmx = 0.0;
            else if (limits) stpmx = t2_fdiv(a2, a1);
          }
        }
      }
      stp = 1.0;  // every variable is boxed, so the first step is not rescaled by 1/|d|
      fold = f;
      gd = 0.0;
      T2_UNROLL
      for (int i = 0; i < N; ++i) gd += g[i] * d[i];
      gdold = gd;
      dcsrch_start(f, gd, stp, 1e-3, 0.0, stpmx, ls);  // ERROR also covers gd >= 0: not a descent direction
      ifun = 1;
      next = (ls.task != LS_FG || ifun - 1 >= P.maxls) ? GO_FAIL : GO_TRIAL;
      T2_LSTAMP(4)
    }
    T2_LSTAMP(5)
    if (next == GO_TRIAL) {
      T2_UNROLL
      for (int i = 0; i < N; ++i) x[i] = stp == 1.0 ? z[i] : stp * d[i] + t[i];
      return false;
    }
    return true;
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
  double hist[2 * Lbfgsb<MODEL>::M * Lbfgsb<MODEL>::N];
  s.init(c.P->x0, lb, ub, hist, 1);
  do {
    s.eval(c);
  } while (!s.advance(c));
  s.result(out);
}

}  // namespace t2fit
