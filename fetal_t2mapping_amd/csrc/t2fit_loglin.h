// t2fit_loglin.h -- closed-form weighted log-linear fit of the 2-parameter model S = k exp(-t/T2).
//
// BASELINE.json configuration 2 names a "2-param log-linear fit"; the reference itself has no such
// routine (its 2-parameter mode runs the same L-BFGS-B loop as the others, run_t2mapping.py:260-272),
// so this solver is an extension: parity with the reference is unpinned, it is checked against its
// own closed form (oracle/t2fit_oracle.py loglinear_fit) and, on noise-free decays inside the bounds,
// against the reference's converged 2-parameter result.
//
// ln y_i = ln k - t_i / T2, weighted by y_i^2 (to first order the weights of the reference's
// least-squares objective, run_t2mapping.py:141-147); samples <= 0 carry no weight.  One pass over
// the echoes, no iteration: this is the only fit on the path that is HBM-bound rather than ALU-bound.
// The logarithm is float32, the five sums and the 2x2 solve float64.  The result is clipped into the
// same box the iterative solvers use (table bounds, or the data-dependent no-prior bounds).
#pragma once

#include "t2fit_lane.h"

// This solver is this library's own formulation, not a restatement of reference arithmetic: multiply-adds may fuse
// (the library as a whole is compiled with -ffp-contract=off, see t2fit_lbfgsb.h).
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

namespace t2fit {

T2_HD float t2_logf_precise(float x) { return logf(x); }  // the ~1 ulp library form, not v_log_f32 alone

// Returns false (and leaves k, T2 untouched) when fewer than two positive samples carry weight or the
// weighted echo times are degenerate.
T2_HD bool loglin_closed_form(const ObjCtx& c, double& k, double& t2) {
  const LaneParams& P = *c.P;
  // One pass over the echoes, numerically equal to the centred two-pass form: t and ln y are taken
  // relative to the heaviest sample (the pivot), which then contributes exactly zero to every sum.
  // The textbook one-pass form (sw*stl - st*sl) / (sw*stt - st^2) cancels catastrophically when one
  // sample carries nearly all the weight (a decay much faster than the echo spacing); with that
  // sample as the pivot the subtracted terms are second order in the other samples' weights.
  int piv = 0, cnt = 0;
  float ypiv = c.sample(0);
  for (int i = 0; i < P.n_te; ++i) {
    const float yf = c.sample(i);
    const bool up = yf > ypiv;
    ypiv = up ? yf : ypiv;
    piv = up ? i : piv;
    cnt += yf > 0.0f;
  }
  if (cnt < 2) return false;
  const double tp = P.te[piv], lp = (double)t2_logf_precise(ypiv);
  double sw = 0.0, st = 0.0, sl = 0.0, stt = 0.0, stl = 0.0;
  for (int i = 0; i < P.n_te; ++i) {
    // a sample <= 0 gets weight 0 and stands in as the pivot value (dl = 0): two selects instead of
    // five conditional accumulations
    const float yf = c.sample(i);
    const bool pos = yf > 0.0f;
    const double y = (double)(pos ? yf : 0.0f), w = y * y;
    const double dt = P.te[i] - tp, dl = (double)t2_logf_precise(pos ? yf : ypiv) - lp;
    const double wt = w * dt;
    sw += w; st += wt; sl += w * dl; stt += wt * dt; stl += wt * dl;
  }
  const double rsw = 1.0 / sw;
  const double sxx = stt - st * st * rsw, sxy = stl - st * sl * rsw;
  if (!(sxx > 0.0)) return false;
  const double slope = sxy / sxx;                                   // = -1/T2
  const double icpt = (lp + sl * rsw) - slope * (tp + st * rsw);    // = ln k
  if (!t2_finite(slope) || !t2_finite(icpt)) return false;
  // a flat or rising signal has no finite decay time: +inf, which the box clips to its upper end
  t2 = slope < 0.0 ? -1.0 / slope : (double)INFINITY;
  k = t2_exp(t2_min(icpt, 700.0));
  return true;
}

// mean squared residual of the 2-parameter model (the reference's objective, :141-147)
T2_HD double loglin_objective(const ObjCtx& c, double k, double t2) {
  const LaneParams& P = *c.P;
  double s = 0.0;
  for (int i = 0; i < P.n_te; ++i) {
    const double r = (double)c.sample(i) - k * t2_exp(-P.te[i] / t2);
    s += r * r;
  }
  return s / P.n_te;
}

T2_HD void loglin_solve(const ObjCtx& c, const double* lb, const double* ub, bool want_fun, LaneResult& r) {
  const LaneParams& P = *c.P;
  double k = 0.0, t2 = 0.0;
  const bool ok = loglin_closed_form(c, k, t2);
  if (ok) {
    r.x[0] = t2_clip(k, lb[0], ub[0]);
    r.x[1] = t2_clip(t2, lb[1], ub[1]);
    r.status = T2FIT_ST_CONVERGED;
  } else {  // nothing to regress on: the clipped table start point, flagged
    r.x[0] = t2_clip(P.x0[0], lb[0], ub[0]);
    r.x[1] = t2_clip(P.x0[1], lb[1], ub[1]);
    r.status = T2FIT_ST_NOT_CONV;
  }
  r.x[2] = 0.0;
  r.nit = 0;
  r.nfev = 1;
  r.fun = want_fun ? loglin_objective(c, r.x[0], r.x[1]) : NAN;
}

}  // namespace t2fit

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
