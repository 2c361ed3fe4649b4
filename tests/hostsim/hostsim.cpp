// hostsim.cpp -- TEST INFRASTRUCTURE: compiles the per-lane solver headers for the host CPU so
// that solver logic can be checked against the golden fixtures where no GPU exists.  It is built
// by tests/hostsim/build.py into tests/hostsim/libt2fit_hostsim.so and loaded only by tests/;
// the product package never loads it and has no CPU execution path.
#include <stdlib.h>

#include "../../fetal_t2mapping_amd/csrc/t2fit_config.h"
#include "../../fetal_t2mapping_amd/csrc/t2fit_dispatch.h"

using namespace t2fit;

// runtime switch over the lane-solver instantiations (the kernels pick theirs at launch time)
static void fit_lane(const LaneParams& P, const ObjCtx& c, bool finite, float y0_raw, LaneResult& r) {
  if (P.solver == T2FIT_SOLVER_LOGLIN) {
    fit_lane_t<T2FIT_SOLVER_LOGLIN, T2FIT_PREC_F64, T2FIT_MODEL_GAUSSIAN>(P, c, finite, y0_raw, r);
  } else if (P.solver == T2FIT_SOLVER_LM) {
    if (P.precision == T2FIT_PREC_F32) {
      if (P.model == T2FIT_MODEL_GAUSSIAN) fit_lane_t<T2FIT_SOLVER_LM, T2FIT_PREC_F32, T2FIT_MODEL_GAUSSIAN>(P, c, finite, y0_raw, r);
      else fit_lane_t<T2FIT_SOLVER_LM, T2FIT_PREC_F32, T2FIT_MODEL_GAUSSIAN_RICIAN>(P, c, finite, y0_raw, r);
    } else {
      if (P.model == T2FIT_MODEL_GAUSSIAN) fit_lane_t<T2FIT_SOLVER_LM, T2FIT_PREC_F64, T2FIT_MODEL_GAUSSIAN>(P, c, finite, y0_raw, r);
      else fit_lane_t<T2FIT_SOLVER_LM, T2FIT_PREC_F64, T2FIT_MODEL_GAUSSIAN_RICIAN>(P, c, finite, y0_raw, r);
    }
  } else {
    if (P.model == T2FIT_MODEL_GAUSSIAN) fit_lane_t<T2FIT_SOLVER_LBFGSB, T2FIT_PREC_F64, T2FIT_MODEL_GAUSSIAN>(P, c, finite, y0_raw, r);
    else if (P.model == T2FIT_MODEL_GAUSSIAN_RICIAN) fit_lane_t<T2FIT_SOLVER_LBFGSB, T2FIT_PREC_F64, T2FIT_MODEL_GAUSSIAN_RICIAN>(P, c, finite, y0_raw, r);
    else fit_lane_t<T2FIT_SOLVER_LBFGSB, T2FIT_PREC_F64, T2FIT_MODEL_RICIAN>(P, c, finite, y0_raw, r);
  }
}


extern "C" int hostsim_config_default(t2fit_config* cfg, int model, int low_field) {
  return config_default_impl(cfg, model, low_field);
}

// rows: (n, nTE) voxel-major float32
extern "C" int hostsim_fit_rows(const t2fit_config* cfg, const float* rows, int64_t n, double* x,
                                double* fun, int32_t* nit, uint8_t* status, float* res, float* r2) {
  const char* why;
  int rc = config_check(cfg, &why);
  if (rc != T2FIT_OK) return rc;
  const LaneParams P = make_lane_params(*cfg);
  for (int64_t v = 0; v < n; ++v) {
    float buf[T2FIT_MAX_TE];
    for (int i = 0; i < cfg->n_te; ++i) buf[i] = rows[v * cfg->n_te + i];
    bool finite;
    float y0_raw;
    const ObjCtx c = prepare_samples(P, buf, 1, finite, y0_raw);
    LaneResult r;
    fit_lane(P, c, finite, y0_raw, r);
    LaneOutputs o;
    lane_epilogue(c, r, o, r2 != nullptr, false);
    for (int j = 0; j < 3; ++j) x[v * 3 + j] = r.x[j];
    fun[v] = r.fun;
    nit[v] = r.nit;
    status[v] = r.status;
    if (getenv("T2_HOSTSIM_NFEV")) nit[v] = r.nfev;  // debugging aid: report evaluations instead of iterations
    if (res) res[v] = o.res;
    if (r2) r2[v] = o.r2;
  }
  return 0;
}

// one voxel with a per-iteration trace (x0, x1, x2, f) for debugging solver trajectories
extern "C" int hostsim_trace_row(const t2fit_config* cfg, const float* row, double* trace, int cap, int* n_out,
                                 double* x, double* fun, int32_t* nit, uint8_t* status) {
  const char* why;
  int rc = config_check(cfg, &why);
  if (rc != T2FIT_OK) return rc;
  const LaneParams P = make_lane_params(*cfg);
  float buf[T2FIT_MAX_TE];
  for (int i = 0; i < cfg->n_te; ++i) buf[i] = row[i];
  bool finite;
  float y0_raw;
  ObjCtx c = prepare_samples(P, buf, 1, finite, y0_raw);
  *n_out = 0;
  c.trace = trace; c.trace_cap = cap; c.trace_n = n_out;
  double lb[3], ub[3];
  if (!lane_bounds(P, y0_raw, lb, ub) || !finite) return -10;
  LaneResult r;
  if (P.solver == T2FIT_SOLVER_LM) {
    if (P.model == T2FIT_MODEL_GAUSSIAN) lm_solve<double, 2>(c, lb, ub, r);
    else lm_solve<double, 3>(c, lb, ub, r);
  } else if (P.model == T2FIT_MODEL_GAUSSIAN) lbfgsb_solve<T2FIT_MODEL_GAUSSIAN>(c, lb, ub, r);
  else if (P.model == T2FIT_MODEL_GAUSSIAN_RICIAN) lbfgsb_solve<T2FIT_MODEL_GAUSSIAN_RICIAN>(c, lb, ub, r);
  else lbfgsb_solve<T2FIT_MODEL_RICIAN>(c, lb, ub, r);
  for (int j = 0; j < 3; ++j) x[j] = r.x[j];
  *fun = r.fun; *nit = r.nit; *status = r.status;
  return 0;
}

// residual map value per row from given float32 parameters (utils/t2map_utils.py:62-89 per lane)
extern "C" int hostsim_residuals(const t2fit_config* cfg, const float* rows, int64_t n, const float* k,
                                 const float* t2, const float* sigma, float* res) {
  const LaneParams P = make_lane_params(*cfg);
  for (int64_t v = 0; v < n; ++v) {
    float buf[T2FIT_MAX_TE];
    for (int i = 0; i < cfg->n_te; ++i) buf[i] = rows[v * cfg->n_te + i];
    bool finite;
    float y0_raw;
    const ObjCtx c = prepare_samples(P, buf, 1, finite, y0_raw);
    res[v] = residual_mean(c, k[v], t2[v], sigma[v]);
  }
  return 0;
}

// every point the reference-trajectory lane solver evaluates (line-search trials included), with f and the
// forward-difference gradient: pts[7 * i] = x0, x1, x2, f, g0, g1, g2.  Debugging aid for comparing the lane solver's
// control flow with scipy's step by step.
extern "C" int hostsim_eval_points(const t2fit_config* cfg, const float* row, double* pts, int cap, int* n_out) {
  const char* why;
  int rc = config_check(cfg, &why);
  if (rc != T2FIT_OK) return rc;
  const LaneParams P = make_lane_params(*cfg);
  float buf[T2FIT_MAX_TE];
  for (int i = 0; i < cfg->n_te; ++i) buf[i] = row[i];
  bool finite;
  float y0_raw;
  ObjCtx c = prepare_samples(P, buf, 1, finite, y0_raw);
  double lb[3], ub[3];
  if (!lane_bounds(P, y0_raw, lb, ub) || !finite) return -10;
  *n_out = 0;
  auto run = [&](auto& s) {
    double hist[60] = {};
    s.init(P.x0, lb, ub, hist, 1);
    do {
      s.eval(c);
      if (*n_out < cap) {
        double* p = pts + 7 * (*n_out)++;
        for (int j = 0; j < 3; ++j) { p[j] = j < s.N ? s.x[j] : 0.0; p[4 + j] = j < s.N ? s.g[j] : 0.0; }
        p[3] = s.f;
      }
    } while (!s.advance(c));
    if (*n_out < cap) pts[7 * (*n_out)] = (double)s.n_reset;  // one slot past the last point: memory resets
  };
  if (P.model == T2FIT_MODEL_GAUSSIAN) { Lbfgsb<T2FIT_MODEL_GAUSSIAN> s; run(s); }
  else if (P.model == T2FIT_MODEL_GAUSSIAN_RICIAN) { Lbfgsb<T2FIT_MODEL_GAUSSIAN_RICIAN> s; run(s); }
  else { Lbfgsb<T2FIT_MODEL_RICIAN> s; run(s); }
  return 0;
}

// The ring's storage form of a correction pair's s (t2fit_lbfgsb.h store_s / load_s): write `n` vectors of `dim`
// (2 or 3) components through every ring slot and read them back.
extern "C" int hostsim_pair_roundtrip(int dim, const double* s_in, int64_t n, double* s_out) {
  if (dim != 2 && dim != 3) return -1;
  for (int64_t v = 0; v < n; ++v) {
    double hist[64] = {};
    const int q = (int)(v % 10);
    if (dim == 3) {
      Lbfgsb<T2FIT_MODEL_GAUSSIAN_RICIAN> s;
      s.hist = hist; s.hstride = 1; s.head = 0;
      s.store_s(q, s_in + 3 * v);
      s.load_s(q, s_out + 3 * v);
    } else {
      Lbfgsb<T2FIT_MODEL_GAUSSIAN> s;
      s.hist = hist; s.hstride = 1; s.head = 0;
      s.store_s(q, s_in + 2 * v);
      s.load_s(q, s_out + 2 * v);
    }
  }
  return 0;
}

// log(i0e(x)) through the four-wide table-driven loop the Rician lane uses (out) and through the one-value Cephes
// form it restates (ref): n groups of four arguments.
extern "C" int hostsim_log_i0e4(const double* x, int64_t n, double* out, double* ref) {
  for (int64_t v = 0; v < n; ++v) {
    t2_log_i0e4(x + 4 * v, out + 4 * v);
    for (int j = 0; j < 4; ++j) ref[4 * v + j] = T2_LOG_I0E(t2_i0e(x[4 * v + j]));
  }
  return 0;
}

// i0e through the shared 30-step loop a wave with lanes on both sides of 8 runs (each lane picks its series'
// coefficients; a single simulated lane never gets there through t2_log_i0e4): groups of four arguments on ONE side
// of 8 each; out = that loop, ref = the one-value Cephes form.
extern "C" int hostsim_i0e4_by_lane(const double* x, int64_t n, double* out, double* ref) {
  for (int64_t v = 0; v < n; ++v) {
    double ax[4];
    for (int j = 0; j < 4; ++j) ax[j] = x[4 * v + j] < 0 ? -x[4 * v + j] : x[4 * v + j];
    const bool lane_small = ax[0] <= 8.0;
    for (int j = 1; j < 4; ++j)
      if ((ax[j] <= 8.0) != lane_small) return -1;
    t2_i0e4_by_lane(ax, lane_small, out + 4 * v);
    for (int j = 0; j < 4; ++j) ref[4 * v + j] = t2_i0e(x[4 * v + j]);
  }
  return 0;
}

// numpy's add.reduce order as the echo loop of the Rician evaluation accumulates it: terms[i * 4 + j] is item i of row
// sum j; nte_special != 0 uses the compile-time echo-count form (n must then be 3..8), 0 the run-time form.
extern "C" int hostsim_rowsums4(const double* terms, int n, int nte_special, double* out) {
  auto run = [&](auto& s) {
    s.init();
    for (int i = 0; i < n; ++i) s.add(i, n, terms + 4 * i);
    for (int j = 0; j < 4; ++j) out[j] = s.total(j);
  };
  if (!nte_special) { RowSums4<0> s; run(s); return 0; }
  switch (n) {
    case 3: { RowSums4<3> s; run(s); return 0; }
    case 4: { RowSums4<4> s; run(s); return 0; }
    case 5: { RowSums4<5> s; run(s); return 0; }
    case 6: { RowSums4<6> s; run(s); return 0; }
    case 7: { RowSums4<7> s; run(s); return 0; }
    case 8: { RowSums4<8> s; run(s); return 0; }
  }
  return -1;
}

// the Rician objective and its forward-difference gradient at `x` as the lane solver evaluates them (run-time echo
// count, or the echo-count specialisation when nte_special), next to the one-objective-at-a-time reference form
// (objective_t / ObjTerm: the statement-by-statement restatement of run_t2mapping.py:157-177): out = f, g0, g1, g2;
// ref = f(x), f(x + h e0), f(x + h e1), f(x + h e2) are not exposed -- ref[0] = objective_t at x only.
extern "C" int hostsim_rician_eval(const t2fit_config* cfg, const float* row, const double* x, int nte_special, double* out,
                                   double* ref) {
  const LaneParams P = make_lane_params(*cfg);
  float buf[T2FIT_MAX_TE];
  for (int i = 0; i < cfg->n_te; ++i) buf[i] = row[i];
  bool finite;
  float y0_raw;
  ObjCtx c = prepare_samples(P, buf, 1, finite, y0_raw);
  double lb[3], ub[3];
  lane_bounds(P, y0_raw, lb, ub);
  double hist[60] = {};
  auto run = [&](auto& s) {
    s.init(x, lb, ub, hist, 1);
    for (int j = 0; j < 3; ++j) s.x[j] = x[j];
    if constexpr (std::remove_reference_t<decltype(s)>::kNte > 0)
      for (int i = 0; i < std::remove_reference_t<decltype(s)>::kNte; ++i) s.ys[i] = buf[i];
    s.eval(c);
    out[0] = s.f; out[1] = s.g[0]; out[2] = s.g[1]; out[3] = s.g[2];
    if constexpr (std::remove_reference_t<decltype(s)>::kNte > 0)  // the sample rotation must have come full circle
      for (int i = 0; i < std::remove_reference_t<decltype(s)>::kNte; ++i)
        if (s.ys[i] != buf[i]) out[0] = NAN;
  };
  ref[0] = objective_t<T2FIT_MODEL_RICIAN>(c, x);
  if (!nte_special) { Lbfgsb<T2FIT_MODEL_RICIAN> s; run(s); return 0; }
  switch (cfg->n_te) {
    case 3: { Lbfgsb<T2FIT_MODEL_RICIAN, 3> s; run(s); return 0; }
    case 4: { Lbfgsb<T2FIT_MODEL_RICIAN, 4> s; run(s); return 0; }
    case 5: { Lbfgsb<T2FIT_MODEL_RICIAN, 5> s; run(s); return 0; }
    case 6: { Lbfgsb<T2FIT_MODEL_RICIAN, 6> s; run(s); return 0; }
    case 7: { Lbfgsb<T2FIT_MODEL_RICIAN, 7> s; run(s); return 0; }
    case 8: { Lbfgsb<T2FIT_MODEL_RICIAN, 8> s; run(s); return 0; }
  }
  return -1;
}


// the lane's log() for i0e values (fdlibm's algorithm, t2fit_lane.h) on n positive numbers
extern "C" int hostsim_log_lean(const double* x, int64_t n, double* out) {
  for (int64_t v = 0; v < n; ++v) out[v] = t2_log_lean(x[v]);
  return 0;
}

// The shared-seed square roots of the evaluations (t2fit_lane.h `_seq`): plain FMA sequences, run here from a seed as coarse as
// the hardware's reciprocal square root (float precision).  x[v] > 0 is the base argument, a[3v..3v+2] three arguments near it.
// out_base / out_near: the sequences; out_h: sqrt from the refined h alone (the Rician lane's way round).
extern "C" int hostsim_sqrt_near(const double* x, const double* a, int64_t n, double* out_base, double* out_near, double* out_h) {
  for (int64_t v = 0; v < n; ++v) {
    const double seed = (double)(1.0f / sqrtf((float)x[v]));
    double h;
    out_base[v] = t2_sqrt_from_seed_seq(x[v], seed, h);
    for (int j = 0; j < 3; ++j) {
      out_near[3 * v + j] = t2_sqrt_near_seq(a[3 * v + j], h);
      out_h[3 * v + j] = t2_sqrt_from_h_seq(a[3 * v + j], t2_rsqrt_half_near_seq(a[3 * v + j], h));
    }
  }
  return 0;
}

// t2_i0e4_by_lane with the reciprocals of the (8, inf) series taken from one shared reciprocal square root (`near`) against
// the same loop with independent divisions and roots: groups of four nearby arguments on one side of 8.
extern "C" int hostsim_i0e4_by_lane_near(const double* x, int64_t n, double* out, double* ref) {
  for (int64_t v = 0; v < n; ++v) {
    const bool lane_small = x[4 * v] <= 8.0;
    t2_i0e4_by_lane(x + 4 * v, lane_small, out + 4 * v, true);
    t2_i0e4_by_lane(x + 4 * v, lane_small, ref + 4 * v, false);
  }
  return 0;
}
