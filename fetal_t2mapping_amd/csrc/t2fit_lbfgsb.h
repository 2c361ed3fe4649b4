// t2fit_lbfgsb.h -- placeholder, replaced by the per-lane L-BFGS-B solver.
#pragma once

#include "t2fit_lane.h"

namespace t2fit {

template <int MODEL>
T2_HD void lbfgsb_solve(const ObjCtx& c, const double* lb, const double* ub, LaneResult& out) {
  const int np = n_params(c.P->model);
  for (int j = 0; j < 3; ++j) out.x[j] = j < np ? t2_clip(c.P->x0[j], lb[j], ub[j]) : 0.0;
  out.fun = objective(c, out.x);
  out.nit = 0;
  out.status = T2FIT_ST_NOT_CONV;
}

}  // namespace t2fit
