// t2fit_dispatch.h -- one voxel end to end: input checks, bounds, solver choice, epilogue values.
// Mirrors the control flow of fit_voxel (run_t2mapping.py:237-312) for a single lane.
//
// fit_lane_t<SOLVER, PREC, MODEL> is the compile-time specialised lane body each kernel
// instantiation uses, so its register allocation covers one path only.
#pragma once

#include "t2fit_lane.h"
#include "t2fit_lbfgsb.h"
#include "t2fit_lm.h"
#include "t2fit_loglin.h"

namespace t2fit {

// Host-side translation of the ABI struct into the kernel argument.
inline LaneParams make_lane_params(const t2fit_config& c) {
  LaneParams P;
  P.model = c.model; P.solver = c.solver; P.precision = c.precision; P.n_te = c.n_te;
  P.no_prior = c.no_prior; P.norm = c.norm; P.maxls = c.maxls; P.maxiter = c.maxiter;
  P.maxfun = c.maxfun; P.numpy_legacy = c.numpy_legacy != 0;
  for (int i = 0; i < T2FIT_MAX_TE; ++i) {
    P.te[i] = i < c.n_te ? c.te_ms[i] : 0.0;
    P.te_f[i] = (float)P.te[i];
  }
  for (int j = 0; j < 3; ++j) { P.x0[j] = c.x0[j]; P.lb[j] = c.lb[j]; P.ub[j] = c.ub[j]; }
  P.ftol = c.ftol; P.gtol = c.gtol; P.fd_step = c.fd_step; P.lm_xtol = c.lm_xtol;
  P.np_k_ub = c.noprior_k_ub; P.np_t2_lb = c.noprior_t2_lb; P.np_t2_ub = c.noprior_t2_ub;
  P.lm_r_lo = 1.0 / (c.no_prior ? c.noprior_t2_ub : c.ub[1]);
  P.lm_r_hi = 1.0 / (c.no_prior ? c.noprior_t2_lb : c.lb[1]);
  P.lm_r_x0 = 1.0 / c.x0[1];
  t2_exp_res_coefficients(P.exp_c);
  P.inv_n = 1.0 / c.n_te;
  P.te_max = 0.0;
  for (int i = 0; i < c.n_te; ++i) P.te_max = c.te_ms[i] > P.te_max ? c.te_ms[i] : P.te_max;
  P.lbfgsb_tol = (c.ftol / 2.220446049250313e-16) * 2.220446049250313e-16;
  return P;
}

struct LaneOutputs {
  float t2, k, sigma, res, r2, fun, se;
  int32_t nit;
  uint8_t status;
};

// Check one voxel's samples (stored at col[i*stride]) and turn them into what the objective sees:
// divided in place by their maximum when cfg.norm (float32 / float32, as numpy does).  y0_raw is
// the first sample before normalisation: the no-prior lower bound of k uses it (run_t2mapping.py:244).
T2_HD ObjCtx prepare_samples(const LaneParams& P, float* col, int stride, bool& finite, float& y0_raw) {
  const int n = P.n_te;
  finite = true;
  y0_raw = col[0];
  float ymax = col[0];
  for (int i = 0; i < n; ++i) {
    const float v = col[i * stride];
    finite = finite && t2_finite(v);
    ymax = v > ymax ? v : ymax;
  }
  if (P.norm) {
    for (int i = 0; i < n; ++i) {
      const float v = col[i * stride] / ymax;
      col[i * stride] = v;
      finite = finite && t2_finite(v);
    }
  }
  ObjCtx c;
  c.P = &P;
  c.y = EchoView{col, stride};
  return c;
}

template <int SOLVER, int PREC, int MODEL>
T2_HD void fit_lane_t(const LaneParams& P, const ObjCtx& c, bool finite, float y0_raw, LaneResult& r) {
  constexpr int NP = MODEL == T2FIT_MODEL_GAUSSIAN ? 2 : 3;
  double lb[3], ub[3];
  const bool feasible = lane_bounds(P, y0_raw, lb, ub);
  r.nit = 0;
  r.nfev = 0;
  r.fun = NAN;
  if (!feasible) {
    // scipy raises "one of the lower bounds is greater than an upper bound" (whole volume aborts)
    r.x[0] = r.x[1] = NAN;
    r.x[2] = NP == 2 ? 0.0 : NAN;
    r.status = T2FIT_ST_INFEASIBLE;
    return;
  }
  if (!finite) {
    // objective is NaN everywhere: scipy stops at the clipped start point, nit = 0, success False
    for (int j = 0; j < 3; ++j) r.x[j] = j < NP ? t2_clip(P.x0[j], lb[j], ub[j]) : 0.0;
    r.status = T2FIT_ST_NONFINITE;
    return;
  }
  if constexpr (SOLVER == T2FIT_SOLVER_LOGLIN) {
    static_assert(MODEL == T2FIT_MODEL_GAUSSIAN, "the log-linear closed form exists for the 2-parameter model only");
    loglin_solve(c, lb, ub, true, r);
  } else if constexpr (SOLVER == T2FIT_SOLVER_LM) {
    static_assert(MODEL != T2FIT_MODEL_RICIAN, "LM handles the least-squares models only");
    if constexpr (PREC == T2FIT_PREC_F32) lm_solve<float, NP>(c, lb, ub, r);
    else lm_solve<double, NP>(c, lb, ub, r);
  } else {
    lbfgsb_solve<MODEL>(c, lb, ub, r);
  }
}

// Coefficient of determination about the mean (formula of the reference's notebook,
// notebooks/20240924_ada_qmri_jmri_invitro.ipynb:345-347).  No reference map exists for it: extension.
T2_HD float r_squared(const ObjCtx& c, double k, double t2, double sg) {
  const LaneParams& P = *c.P;
  const int n = P.n_te;
  double mean = 0.0;
  for (int i = 0; i < n; ++i) mean += (double)c.sample(i);
  mean /= n;
  double ss_tot = 0.0, ss_res = 0.0;
  const bool sq = P.model == T2FIT_MODEL_GAUSSIAN_RICIAN;
  for (int i = 0; i < n; ++i) {
    const double yi = (double)c.sample(i);
    const double E = t2_exp(-P.te[i] / t2);
    const double m = sq ? t2_sqrt(k * k * E * E + sg * sg) : k * E;
    ss_res += (yi - m) * (yi - m);
    ss_tot += (yi - mean) * (yi - mean);
  }
  return (float)(1.0 - ss_res / ss_tot);
}

// Standard error of T2 from the Gauss-Newton covariance s^2 (J^T J)^-1 at (k, T2, sigma), with
// s^2 = SS_res / (nTE - n_par).  Extension (BASELINE.json config 3 asks for CI maps, the reference has
// none): 95 % confidence interval = T2 +- 1.96 * this.  NaN when under-determined or singular.
T2_HD float t2_std_error(const ObjCtx& c, double k, double t2, double sg) {
  const LaneParams& P = *c.P;
  const int n = P.n_te;
  const bool sq = P.model == T2FIT_MODEL_GAUSSIAN_RICIAN;
  const int np = P.model == T2FIT_MODEL_GAUSSIAN ? 2 : 3;
  if (n <= np) return NAN;
  double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, ss = 0;
  for (int i = 0; i < n; ++i) {
    const double t = P.te[i];
    const double E = t2_exp(-t / t2);
    double m, jk, jt, js;
    if (sq) {
      m = t2_sqrt(k * k * E * E + sg * sg);
      jk = k * E * E / m;
      jt = k * k * E * E * (t / (t2 * t2)) / m;
      js = sg / m;
    } else {
      m = k * E;
      jk = E;
      jt = m * t / (t2 * t2);
      js = 0.0;
    }
    const double r = (double)c.sample(i) - m;
    ss += r * r;
    a00 += jk * jk; a01 += jk * jt; a02 += jk * js; a11 += jt * jt; a12 += jt * js; a22 += js * js;
  }
  const double s2 = ss / (n - np);
  double var;
  if (sq) {
    // [(J^T J)^-1]_{11} = cofactor / determinant of the symmetric 3x3
    const double det = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
    var = s2 * (a00 * a22 - a02 * a02) / det;
  } else if (np == 3) {
    // Rician likelihood model: sigma is not a parameter of the mean model; covariance over (k, T2)
    var = s2 * a00 / (a00 * a11 - a01 * a01);
  } else {
    var = s2 * a00 / (a00 * a11 - a01 * a01);
  }
  return var > 0.0 ? (float)t2_sqrt(var) : NAN;
}

// Epilogue: float32 map values (run_t2mapping.py:456-458 casts), then the residual map and the
// optional R^2 evaluated from those float32 maps (utils/t2map_utils.py:62-89).
T2_HD void lane_epilogue(const ObjCtx& c, const LaneResult& r, LaneOutputs& o, bool want_r2, bool want_se) {
  o.k = (float)r.x[0];
  o.t2 = (float)r.x[1];
  o.sigma = (float)r.x[2];
  o.fun = (float)r.fun;
  o.nit = r.nit;
  o.status = r.status;
  o.res = residual_mean(c, o.k, o.t2, o.sigma);
  o.r2 = want_r2 ? r_squared(c, (double)o.k, (double)o.t2, (double)o.sigma) : 0.0f;
  o.se = want_se ? t2_std_error(c, (double)o.k, (double)o.t2, (double)o.sigma) : 0.0f;
}

}  // namespace t2fit
