// t2fit_config.h -- the reference's six fit tables (run_t2mapping.py:38-106) and argument checks.
#pragma once

#include <string.h>

#include "../../include/t2fit.h"

namespace t2fit {

// returns 0 or T2FIT_E_INVALID
inline int config_default_impl(t2fit_config* cfg, int model, int low_field) {
  if (!cfg || model < 0 || model > 2) return T2FIT_E_INVALID;
  memset(cfg, 0, sizeof(*cfg));
  cfg->abi_version = T2FIT_ABI_VERSION;
  cfg->model = model;
  cfg->solver = T2FIT_SOLVER_LBFGSB;
  cfg->precision = T2FIT_PREC_F64;
  cfg->n_te = 3;
  cfg->te_ms[0] = low_field ? 114.0 : 115.0;  // run_t2mapping.py:540-545
  cfg->te_ms[1] = 202.0;
  cfg->te_ms[2] = 299.0;
  cfg->maxls = 50;
  cfg->maxiter = 15000;  // scipy L-BFGS-B defaults
  cfg->maxfun = 15000;
  cfg->fd_step = 1e-8;
  cfg->noprior_k_ub = 10000.0;
  cfg->noprior_t2_lb = 10.0;
  cfg->noprior_t2_ub = 2000.0;
  const bool lf = low_field != 0;
  if (model == T2FIT_MODEL_GAUSSIAN) {
    cfg->x0[0] = lf ? 650.0 : 890.0; cfg->x0[1] = 165.0; cfg->x0[2] = 0.0;
    cfg->lb[0] = lf ? 600.0 : 850.0; cfg->ub[0] = lf ? 10000.0 : 30000.0;
    cfg->lb[1] = 10.0; cfg->ub[1] = 600.0;
    cfg->lb[2] = 0.0; cfg->ub[2] = 0.0;
    cfg->ftol = 1e-6;
    cfg->gtol = 1e-5;  // not in the table: scipy default
  } else {
    cfg->ftol = 1e-2;
    cfg->gtol = 1e-2;
    if (lf) {
      cfg->x0[0] = 650.0; cfg->x0[1] = 110.0; cfg->x0[2] = 40.0;
      cfg->lb[0] = 550.0; cfg->ub[0] = model == T2FIT_MODEL_RICIAN ? 900.0 : 10000.0;
      cfg->lb[1] = 10.0; cfg->ub[1] = 600.0;
      cfg->lb[2] = 2.0; cfg->ub[2] = 1000.0;
    } else if (model == T2FIT_MODEL_GAUSSIAN_RICIAN) {
      cfg->x0[0] = 890.0; cfg->x0[1] = 110.0; cfg->x0[2] = 40.0;
      cfg->lb[0] = 850.0; cfg->ub[0] = 30000.0;
      cfg->lb[1] = 30.0; cfg->ub[1] = 600.0;
      cfg->lb[2] = 2.0; cfg->ub[2] = 1000.0;
    } else {
      cfg->x0[0] = 17.0; cfg->x0[1] = 40.0; cfg->x0[2] = 0.15;
      cfg->lb[0] = 850.0; cfg->ub[0] = 30000.0;
      cfg->lb[1] = 30.0; cfg->ub[1] = 600.0;
      cfg->lb[2] = 7.0; cfg->ub[2] = 200.0;
    }
  }
  return T2FIT_OK;
}

// Argument validation shared by every entry point.  `why` receives a static message.
inline int config_check(const t2fit_config* c, const char** why) {
  *why = "";
  if (!c) { *why = "cfg is NULL"; return T2FIT_E_INVALID; }
  if (c->abi_version != T2FIT_ABI_VERSION) { *why = "cfg.abi_version mismatch"; return T2FIT_E_INVALID; }
  if (c->model < 0 || c->model > 2) { *why = "unknown model"; return T2FIT_E_INVALID; }
  if (c->solver != T2FIT_SOLVER_LBFGSB && c->solver != T2FIT_SOLVER_LM && c->solver != T2FIT_SOLVER_LOGLIN) {
    *why = "unknown solver";
    return T2FIT_E_INVALID;
  }
  if (c->solver == T2FIT_SOLVER_LOGLIN && c->model != T2FIT_MODEL_GAUSSIAN) {
    *why = "the log-linear closed form exists for the 2-parameter gaussian model only";
    return T2FIT_E_INVALID;
  }
  if (c->solver == T2FIT_SOLVER_LM && c->model == T2FIT_MODEL_RICIAN) {
    *why = "the LM solver handles the least-squares models only; use T2FIT_SOLVER_LBFGSB for rician";
    return T2FIT_E_INVALID;
  }
  if (c->precision != T2FIT_PREC_F64 && c->precision != T2FIT_PREC_F32) { *why = "unknown precision"; return T2FIT_E_INVALID; }
  if (c->n_te < 2 || c->n_te > T2FIT_MAX_TE) { *why = "n_te out of range [2, 32]"; return T2FIT_E_INVALID; }
  for (int i = 0; i < c->n_te; ++i)
    if (!(c->te_ms[i] > 0.0) || (i > 0 && !(c->te_ms[i] > c->te_ms[i - 1]))) {
      *why = "te_ms must be positive and ascending";
      return T2FIT_E_INVALID;
    }
  if (c->numpy_legacy != 0 && c->numpy_legacy != 1) { *why = "numpy_legacy must be 0 or 1"; return T2FIT_E_INVALID; }
  if (c->maxls <= 0) { *why = "maxls must be positive"; return T2FIT_E_INVALID; }  // scipy raises too
  const int np = c->model == T2FIT_MODEL_GAUSSIAN ? 2 : 3;
  for (int j = 0; j < np; ++j) {
    if (c->no_prior && j < 2) continue;  // replaced per voxel
    if (c->lb[j] > c->ub[j]) { *why = "table bounds have lb > ub"; return T2FIT_E_BOUNDS; }
  }
  if (!(c->lb[1] > 0.0) && !c->no_prior) { *why = "T2 lower bound must be positive"; return T2FIT_E_INVALID; }
  // the objective evaluations use exp() without the library's range selects (t2_exp_core): exact, denormals and
  // underflow to 0 included, while the argument's exponent fits an int, i.e. for -2 TE / T2 above -1.4e9
  const double t2_min = c->no_prior ? c->noprior_t2_lb : c->lb[1];
  if (!(t2_min > 0.0) || 2.0 * c->te_ms[c->n_te - 1] / t2_min > 1.0e9) {
    *why = "echo times absurdly long for the T2 lower bound (2 TE / T2 must not exceed 1e9)";
    return T2FIT_E_INVALID;
  }
  return T2FIT_OK;
}

}  // namespace t2fit
