// t2fit_lm.h -- per-lane bounded Levenberg-Marquardt for the two least-squares objectives.
//
// Minimises the reference's objective (run_t2mapping.py:141-155) under the reference's box bounds
// (:38-106, :243-245) to convergence.  The reference itself stops its quasi-Newton solver early
// (ftol = gtol = 1e-2 for the 3-parameter model), so this solver is the "converged" alternative
// to the trajectory-faithful L-BFGS-B lane solver in t2fit_lbfgsb.h, not a replica of it.
//
// Method: closed-form weighted log-linear seed for (k, T2) and a data-driven seed for the noise
// floor, then projected LM in (k, R = 1/T2) for the 2-parameter model and (k^2, R, sigma^2) for the
// 3-parameter model, with Marquardt scaling, an active set taken from the gradient sign at the
// bounds, trial points clipped to the box, Nielsen's gain-ratio damping update, and one restart from
// the table start point when the seeded run ends with T2 on a bound.  T is float or double.
#pragma once

#include "t2fit_lane.h"

// This solver is this library's own formulation, not a restatement of reference arithmetic: multiply-adds may fuse
// (the library as a whole is compiled with -ffp-contract=off, see t2fit_lbfgsb.h).
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

namespace t2fit {

template <typename T> struct LmEval {
  T f;      // sum of squared residuals (not yet divided by n)
  T a[6];   // J^T J, packed: kk, kR, ks, RR, Rs, ss
  T g[3];   // J^T r  (descent step solves (A + lambda D) d = g)
};

// Residuals and model Jacobian at q.  NP = 2: q = (k, R), m = k E with E = exp(-t R), R = 1/T2.
// NP = 3: q = (a, R, u) = (k^2, 1/T2, sigma^2), m = sqrt(a E^2 + u): m^2 is linear in (a, u), which
// removes the flat, badly scaled sigma direction the (k, T2, sigma) form has when sigma << k E
// (d m / d sigma = sigma / m -> 0) and halves the number of iterations.  The model is even in k and
// sigma, so squaring loses nothing; the box maps monotonically.
template <typename T, int NP, int NTE = 0>
T2_HD void lm_eval(const ObjCtx& c, const T* q, LmEval<T>& e) {
  const LaneParams& P = *c.P;
  const int n = NTE > 0 ? NTE : P.n_te;  // NTE > 0: echo count known at compile time (one straight block)
  const T k = q[0], R = q[1], s = (NP == 3) ? q[2] : T(0);
  T f = 0, akk = 0, akr = 0, aks = 0, arr = 0, ars = 0, ass = 0, gk = 0, gr = 0, gs = 0;
  auto echo = [&](int i) {
    const T t = TeOf<T>::at(P, i);
    const T y = (T)c.sample(i);
    T jk, jr, js = 0, m;
    if (NP == 2) {
      const T E = t2_exp(-t * R);
      m = k * E;
      jk = E;
      jr = -t * m;
    } else {
      const T E2 = t2_exp(T(-2) * t * R);
      const T v = k * E2 + s;
      const T rm = t2_rsqrt(v);
      m = v * rm;
      js = T(0.5) * rm;
      jk = E2 * js;
      jr = T(-2) * t * k * jk;
    }
    const T r = y - m;
    f += r * r;
    akk += jk * jk;
    akr += jk * jr;
    arr += jr * jr;
    gk += jk * r;
    gr += jr * r;
    if (NP == 3) {
      aks += jk * js;
      ars += jr * js;
      ass += js * js;
      gs += js * r;
    }
  };
  // the first eight echoes with compile-time indices: their echo times (scalar loads) and samples (LDS) are
  // all requested up front instead of one load-wait pair per loop trip; longer trains continue in a loop
  static_for<0, 8>([&](auto JC) {
    constexpr int J = decltype(JC)::value;
    if (J < n) echo(J);
  });
  for (int i = 8; i < n; ++i) echo(i);
  e.f = f;
  e.a[0] = akk; e.a[1] = akr; e.a[2] = aks; e.a[3] = arr; e.a[4] = ars; e.a[5] = ass;
  e.g[0] = gk; e.g[1] = gr; e.g[2] = gs;
}

// Solve the damped, active-set-reduced normal equations.  Fixed coordinates get d = 0.
template <typename T, int NP>
T2_HD void lm_step(const LmEval<T>& e, T lambda, const bool* fixed, T* d) {
  T a00 = e.a[0], a01 = e.a[1], a02 = e.a[2], a11 = e.a[3], a12 = e.a[4], a22 = e.a[5];
  T g0 = e.g[0], g1 = e.g[1], g2 = e.g[2];
  const T tiny = T(1e-30);
  a00 += lambda * t2_max(a00, tiny);
  a11 += lambda * t2_max(a11, tiny);
  a22 += lambda * t2_max(a22, tiny);
  if (fixed[0]) { a00 = 1; a01 = 0; a02 = 0; g0 = 0; }
  if (fixed[1]) { a11 = 1; a01 = 0; a12 = 0; g1 = 0; }
  if (NP == 2 || fixed[2]) { a22 = 1; a02 = 0; a12 = 0; g2 = 0; }
  // LDL^T of the symmetric 3x3; three reciprocals instead of six divisions (the step is only a
  // proposal: it is accepted or rejected on the objective, so its last bits do not matter)
  const T r0 = t2_rcp(a00);
  const T l10 = a01 * r0, l20 = a02 * r0;
  const T d1 = a11 - l10 * a01;
  const T r1 = t2_rcp(d1);
  const T l21 = (a12 - l20 * a01) * r1;
  const T d2 = a22 - l20 * a02 - l21 * (a12 - l20 * a01);
  const T r2 = t2_rcp(d2);
  const T z0 = g0;
  const T z1 = g1 - l10 * z0;
  const T z2 = g2 - l20 * z0 - l21 * z1;
  const T x2 = z2 * r2;
  const T x1 = z1 * r1 - l21 * x2;
  const T x0 = z0 * r0 - l10 * x1 - l20 * x2;
  d[0] = x0; d[1] = x1; d[2] = x2;
}

template <typename T> struct LmTol;
// xtol: relative step; ftiny: relative objective decrease (actual, or predicted by the damped model) below
// which the fit has converged; fnoise: how far rounding can push the objective of a trial point up
template <> struct LmTol<double> { static constexpr double xtol = 1e-6, ftiny = 1e-12, fnoise = 1e-13; };
template <> struct LmTol<float> { static constexpr float xtol = 1e-5f, ftiny = 1e-6f, fnoise = 4e-6f; };

// Weighted log-linear regression ln y = ln k - R t with weights y^2 (matches the least-squares
// objective to first order).  Returns false if fewer than two positive samples.
template <typename T>
T2_HD bool loglinear_seed(const ObjCtx& c, T& k, T& R) {
  const LaneParams& P = *c.P;
  T sw = 0, st = 0, stt = 0, sl = 0, stl = 0;
  int cnt = 0;
  for (int i = 0; i < P.n_te; ++i) {
    const T y = (T)c.sample(i);
    if (y > T(0)) {
      const T t = TeOf<T>::at(P, i);
      const T w = y * y, l = t2_log(y);
      sw += w; st += w * t; stt += w * t * t; sl += w * l; stl += w * t * l;
      ++cnt;
    }
  }
  const T det = sw * stt - st * st;
  if (cnt < 2 || !(det > T(0))) return false;
  const T slope = (sw * stl - st * sl) / det;  // = -R
  const T icpt = (sl - slope * st) / sw;
  R = -slope;
  k = t2_exp(t2_min(icpt, T(60)));
  return t2_finite(k) && t2_finite(R);
}

// Resumable form (same protocol as Lbfgsb): init() seeds the fit, eval() evaluates residuals and
// Jacobian at the pending point, advance() accepts or rejects it and proposes the next one.
template <typename T, int NP, int NTE = 0>
struct LmLane {
  T lo[3], hi[3];        // box in (k | k^2, R, sigma^2)
  T q[3], qn[3], d[3];   // accepted point, pending point, step that led to it
  LmEval<T> e, en;
  T lambda, nu, rel, pred, xtol;
  double lb0;            // lower bound of k: the only per-voxel bound (no-prior mode); the others are in LaneParams
  int it, maxit;
  uint8_t status;
  bool first;
  // second start: when the seeded run ends with T2 on a bound (decayed / flat / noise-only voxels,
  // where the objective has one minimum at each end of the T2 interval) the fit is repeated from the
  // table start point and the lower objective wins
  int stage;
  T qbest[3], fbest;
  uint8_t sbest;

  // the table start point in (k | k^2, R, sigma^2), clipped into the box
  T2_HD void table_start(const LaneParams& P, T* out) const {
    auto sq = [](double v) { return v > 0.0 ? v * v : 0.0; };
    out[0] = t2_clip(NP == 3 ? (T)sq(P.x0[0]) : (T)P.x0[0], lo[0], hi[0]);
    out[1] = t2_clip((T)P.lm_r_x0, lo[1], hi[1]);
    out[2] = (NP == 3) ? t2_clip((T)sq(P.x0[2]), lo[2], hi[2]) : T(0);
  }

  T2_HD void init(const ObjCtx& c, const double* x0, const double* lb_, const double* ub_) {
    const LaneParams& P = *c.P;
    auto sq = [](double v) { return v > 0.0 ? v * v : 0.0; };  // |.| >= 0 side of an even parameter
    lb0 = lb_[0];
    lo[0] = NP == 3 ? (T)sq(lb_[0]) : (T)lb_[0];
    hi[0] = NP == 3 ? (T)sq(ub_[0]) : (T)ub_[0];
    lo[1] = (T)P.lm_r_lo; hi[1] = (T)P.lm_r_hi;  // R = 1/T2 reverses the interval: 1/ub, 1/lb (make_lane_params)
    lo[2] = (T)sq(lb_[2]); hi[2] = (T)sq(ub_[2]);
    table_start(P, qn);
    stage = 0;
    T ks, Rs;
    if (loglinear_seed<T>(c, ks, Rs)) {
      qn[0] = t2_clip(NP == 3 ? ks * ks : ks, lo[0], hi[0]);
      qn[1] = t2_clip(Rs, lo[1], hi[1]);
      if (NP == 3) {
        // noise-floor seed: m^2 = a E^2 + u, so u ~ mean(y^2 - a E^2) at the seeded (a, R)
        T acc = 0;
        for (int i = 0; i < P.n_te; ++i) {
          const T y = (T)c.sample(i);
          acc += y * y - qn[0] * t2_exp(T(-2) * TeOf<T>::at(P, i) * qn[1]);
        }
        const T u = acc / (T)P.n_te;
        if (t2_finite(u)) qn[2] = t2_clip(u, lo[2], hi[2]);
      }
    }
    xtol = P.lm_xtol > 0 ? (T)P.lm_xtol : (T)LmTol<T>::xtol;
    lambda = T(1e-3);
    nu = T(2);
    it = 0;
    maxit = P.maxiter > 0 ? P.maxiter : 60;
    status = T2FIT_ST_NOT_CONV;
    first = true;
  }

  T2_HD void eval(const ObjCtx& c) { lm_eval<T, NP, NTE>(c, qn, en); }

  // the current run has ended with `status`; returns false if a second run was started instead
  T2_HD bool finish_run(const LaneParams& P) {
    const bool at_t2_bound = q[1] <= lo[1] || q[1] >= hi[1];
    if (stage == 0 && at_t2_bound && status != T2FIT_ST_NONFINITE && it < maxit) {
      T x0q[3];
      table_start(P, x0q);
      bool same = true;
      for (int j = 0; j < 3; ++j) same = same && x0q[j] == q[j];
      if (!same) {
        stage = 1;
        for (int j = 0; j < 3; ++j) { qbest[j] = q[j]; qn[j] = x0q[j]; }
        fbest = e.f;
        sbest = status;
        lambda = T(1e-3);
        nu = T(2);
        status = T2FIT_ST_NOT_CONV;
        first = true;
        return false;
      }
    }
    if (stage == 1 && !(e.f < fbest)) {  // the seeded run was at least as good: keep it
      for (int j = 0; j < 3; ++j) q[j] = qbest[j];
      e.f = fbest;
      status = sbest;
    }
    return true;
  }

  T2_HD bool advance(const ObjCtx& c) {
    if (!step(c)) return false;
    return finish_run(*c.P);
  }

  T2_HD bool step(const ObjCtx& c) {
    if (first) {
      first = false;
      for (int j = 0; j < 3; ++j) q[j] = qn[j];
      e = en;
      if (!t2_finite(e.f)) { status = T2FIT_ST_NONFINITE; return true; }
    } else {
      const T act = e.f - en.f;
      if (t2_finite(en.f) && act >= T(0)) {
        const T rho = pred > T(0) ? act * t2_rcp(pred) : T(1);
        T fac = T(2) * rho - T(1);
        fac = T(1) - fac * fac * fac;
        lambda *= t2_max(T(1) / T(3), fac);
        lambda = t2_max(lambda, T(1e-12));
        nu = T(2);
        for (int j = 0; j < 3; ++j) q[j] = qn[j];
        e = en;
        if (c.trace && *c.trace_n < c.trace_cap) {
          double* tr = c.trace + 4 * (*c.trace_n)++;
          tr[0] = NP == 3 ? sqrt((double)q[0]) : (double)q[0]; tr[1] = 1.0 / (double)q[1];
          tr[2] = NP == 3 ? sqrt((double)q[2]) : 0.0; tr[3] = (double)e.f / c.P->n_te;
        }
        if (rel <= xtol || act <= (T)LmTol<T>::ftiny * e.f) { status = T2FIT_ST_CONVERGED; ++it; return true; }
      } else {
        // rejected.  If the damped model itself promised less than the objective can resolve in T, the
        // rejection is rounding noise and no damping will do better (MINPACK's actred/prered test): stop
        // here instead of walking lambda up through a dozen more evaluations.
        if (t2_finite(en.f) && pred <= (T)LmTol<T>::ftiny * e.f && -act <= (T)LmTol<T>::fnoise * e.f) {
          status = T2FIT_ST_CONVERGED;
          ++it;
          return true;
        }
        lambda *= nu;
        nu *= T(2);
        if (lambda > T(1e14)) { status = T2FIT_ST_CONVERGED; return true; }  // no descent left at any damping
      }
      ++it;
      if (it >= maxit) return true;
    }
    // propose the next point: active set from the gradient sign at the bounds, damped step, clip
    bool fixed[3];
    for (int j = 0; j < 3; ++j)
      fixed[j] = (q[j] <= lo[j] && e.g[j] < T(0)) || (q[j] >= hi[j] && e.g[j] > T(0));
    lm_step<T, NP>(e, lambda, fixed, d);
    bool moved = false;
    rel = 0;
    for (int j = 0; j < 3; ++j) {
      qn[j] = t2_clip(q[j] + d[j], lo[j], hi[j]);
      if (j >= NP) qn[j] = 0;
      const T dj = qn[j] - q[j];
      moved = moved || dj != T(0);
      rel = t2_max(rel, t2_abs(dj) * t2_rcp(t2_abs(q[j]) + (j == 1 ? T(1e-6) : (NP == 3 ? T(1) : T(1e-3)))));
    }
    if (!moved) { status = T2FIT_ST_CONVERGED; return true; }
    // predicted reduction of the (unclipped) damped model: d^T (lambda D d + g)
    pred = 0;
    const T dd[3] = {e.a[0], e.a[3], e.a[5]};
    for (int j = 0; j < NP; ++j) pred += d[j] * (lambda * t2_max(dd[j], T(1e-30)) * d[j] + e.g[j]);
    return false;
  }

  T2_HD void result(const ObjCtx& c, LaneResult& out) const {
    // back to (k, T2, sigma); a coordinate sitting on a bound is snapped onto it exactly
    const LaneParams& P = *c.P;
    const double lbd[3] = {lb0, P.no_prior ? P.np_t2_lb : P.lb[1], P.lb[2]};
    const double ubd[3] = {P.no_prior ? P.np_k_ub : P.ub[0], P.no_prior ? P.np_t2_ub : P.ub[1], P.ub[2]};
    // (square root and reciprocal in the solver's own precision: the iterate carries no more than that)
    out.x[0] = NP == 3 ? (q[0] <= lo[0] && lbd[0] > 0.0 ? lbd[0] : (q[0] >= hi[0] ? ubd[0] : (double)t2_sqrt(q[0])))
                       : (double)q[0];
    out.x[1] = q[1] <= lo[1] ? ubd[1] : (q[1] >= hi[1] ? lbd[1] : (double)(T(1) / q[1]));
    out.x[2] = (NP == 3) ? (q[2] <= lo[2] && lbd[2] > 0.0 ? lbd[2] : (q[2] >= hi[2] ? ubd[2] : (double)t2_sqrt(q[2]))) : 0.0;
    out.fun = (double)e.f / c.P->n_te;
    out.nit = it;
    out.nfev = it + 1;
    out.status = status;
  }
};

template <typename T, int NP>
T2_HD void lm_solve(const ObjCtx& c, const double* lbd, const double* ubd, LaneResult& out) {
  LmLane<T, NP> s;
  s.init(c, c.P->x0, lbd, ubd);
  do {
    s.eval(c);
  } while (!s.advance(c));
  s.result(c, out);
}

}  // namespace t2fit

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
