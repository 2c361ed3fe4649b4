// t2fit_lm.h -- per-lane bounded Levenberg-Marquardt for the two least-squares objectives.
//
// Minimises the reference's objective (run_t2mapping.py:141-155) under the reference's box bounds
// (:38-106, :243-245) to convergence.  The reference itself stops its quasi-Newton solver early
// (ftol = gtol = 1e-2 for the 3-parameter model), so this solver is the "converged" alternative
// to the trajectory-faithful L-BFGS-B lane solver in t2fit_lbfgsb.h, not a replica of it.
//
// Method: closed-form weighted log-linear seed, then projected LM in (k, R = 1/T2, sigma) with
// Marquardt scaling, an active set taken from the gradient sign at the bounds, trial points
// clipped to the box, and Nielsen's gain-ratio damping update.  T is float or double.
#pragma once

#include "t2fit_lane.h"

namespace t2fit {

template <typename T> struct LmEval {
  T f;      // sum of squared residuals (not yet divided by n)
  T a[6];   // J^T J, packed: kk, kR, ks, RR, Rs, ss
  T g[3];   // J^T r  (descent step solves (A + lambda D) d = g)
};

// Residuals and model Jacobian at q = (k, R, s).  NP = 2: m = k E;  NP = 3: m = sqrt(k^2 E^2 + s^2).
template <typename T, int NP>
T2_HD void lm_eval(const ObjCtx& c, const T* q, LmEval<T>& e) {
  const LaneParams& P = *c.P;
  const int n = P.n_te;
  const T k = q[0], R = q[1], s = (NP == 3) ? q[2] : T(0);
  T f = 0, akk = 0, akr = 0, aks = 0, arr = 0, ars = 0, ass = 0, gk = 0, gr = 0, gs = 0;
  const T k2 = k * k, s2 = s * s;
  for (int i = 0; i < n; ++i) {
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
      const T v = k2 * E2 + s2;
      const T rm = t2_rsqrt(v);
      m = v * rm;
      jk = k * E2 * rm;
      jr = -t * k * jk;
      js = s * rm;
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
  }
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
  // LDL^T of the symmetric 3x3
  const T d0 = a00;
  const T l10 = a01 / d0, l20 = a02 / d0;
  const T d1 = a11 - l10 * a01;
  const T l21 = (a12 - l20 * a01) / d1;
  const T d2 = a22 - l20 * a02 - l21 * (a12 - l20 * a01);
  const T z0 = g0;
  const T z1 = g1 - l10 * z0;
  const T z2 = g2 - l20 * z0 - l21 * z1;
  const T x2 = z2 / d2;
  const T x1 = z1 / d1 - l21 * x2;
  const T x0 = z0 / d0 - l10 * x1 - l20 * x2;
  d[0] = x0; d[1] = x1; d[2] = x2;
}

template <typename T> struct LmTol;
template <> struct LmTol<double> { static constexpr double xtol = 1e-10, ftiny = 1e-15; };
template <> struct LmTol<float> { static constexpr float xtol = 2e-6f, ftiny = 1e-7f; };

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

template <typename T, int NP>
T2_HD void lm_solve(const ObjCtx& c, const double* lbd, const double* ubd, LaneResult& out) {
  const LaneParams& P = *c.P;
  const int n = P.n_te;
  // bounds in (k, R, s); R = 1/T2 reverses the T2 interval
  T lo[3], hi[3];
  lo[0] = (T)lbd[0]; hi[0] = (T)ubd[0];
  lo[1] = (T)(1.0 / ubd[1]); hi[1] = (T)(1.0 / lbd[1]);
  lo[2] = (T)lbd[2]; hi[2] = (T)ubd[2];
  T q[3];
  q[0] = t2_clip((T)P.x0[0], lo[0], hi[0]);
  q[1] = t2_clip((T)(1.0 / P.x0[1]), lo[1], hi[1]);
  q[2] = (NP == 3) ? t2_clip((T)P.x0[2], lo[2], hi[2]) : T(0);
  {
    T ks, Rs;
    if (loglinear_seed<T>(c, ks, Rs)) {
      q[0] = t2_clip(ks, lo[0], hi[0]);
      q[1] = t2_clip(Rs, lo[1], hi[1]);
    }
  }
  LmEval<T> e;
  lm_eval<T, NP>(c, q, e);
  const T xtol = P.lm_xtol > 0 ? (T)P.lm_xtol : (T)LmTol<T>::xtol;
  T lambda = T(1e-3), nu = T(2);
  int it = 0;
  uint8_t status = T2FIT_ST_NOT_CONV;
  const int maxit = P.maxiter > 0 ? P.maxiter : 60;
  if (!t2_finite(e.f)) {
    status = T2FIT_ST_NONFINITE;
  } else {
    for (; it < maxit; ++it) {
      bool fixed[3];
      for (int j = 0; j < 3; ++j)
        fixed[j] = (q[j] <= lo[j] && e.g[j] < T(0)) || (q[j] >= hi[j] && e.g[j] > T(0));
      T d[3];
      lm_step<T, NP>(e, lambda, fixed, d);
      T qn[3];
      bool moved = false;
      T rel = 0;
      for (int j = 0; j < 3; ++j) {
        qn[j] = t2_clip(q[j] + d[j], lo[j], hi[j]);
        if (j >= NP) qn[j] = 0;
        const T dj = qn[j] - q[j];
        moved = moved || dj != T(0);
        rel = t2_max(rel, t2_abs(dj) / (t2_abs(q[j]) + (j == 1 ? T(1e-6) : T(1e-3))));
      }
      if (!moved) { status = T2FIT_ST_CONVERGED; break; }
      LmEval<T> en;
      lm_eval<T, NP>(c, qn, en);
      // predicted reduction of the (unclipped) damped model: d^T (lambda D d + g)
      T pred = 0;
      {
        const T dd[3] = {e.a[0], e.a[3], e.a[5]};
        for (int j = 0; j < NP; ++j) pred += d[j] * (lambda * t2_max(dd[j], T(1e-30)) * d[j] + e.g[j]);
      }
      const T act = e.f - en.f;
      if (t2_finite(en.f) && act >= T(0)) {
        const T rho = pred > T(0) ? act / pred : T(1);
        T fac = T(2) * rho - T(1);
        fac = T(1) - fac * fac * fac;
        lambda *= t2_max(T(1) / T(3), fac);
        lambda = t2_max(lambda, T(1e-12));
        nu = T(2);
        for (int j = 0; j < 3; ++j) q[j] = qn[j];
        e = en;
        if (rel <= xtol || act <= (T)LmTol<T>::ftiny * e.f) { status = T2FIT_ST_CONVERGED; ++it; break; }
      } else {
        lambda *= nu;
        nu *= T(2);
        if (lambda > T(1e14)) { status = T2FIT_ST_CONVERGED; break; }  // no descent left at any damping
      }
    }
  }
  out.x[0] = (double)q[0];
  // snap T2 exactly onto a bound the rate sits on (1/(1/b) need not round-trip)
  out.x[1] = q[1] <= lo[1] ? ubd[1] : (q[1] >= hi[1] ? lbd[1] : 1.0 / (double)q[1]);
  out.x[2] = (NP == 3) ? (double)q[2] : 0.0;
  out.fun = (double)e.f / n;
  out.nit = it;
  out.nfev = it + 1;
  out.status = status;
}

}  // namespace t2fit
