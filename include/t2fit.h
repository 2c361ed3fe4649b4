/*
 * t2fit.h -- C ABI of the MI355X per-voxel T2 relaxation fitter (libt2fit_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of Medical-Image-Analysis-Laboratory/
 * fetal_t2mapping: the voxel-wise fit that run_t2mapping.py drives.  The reference has no FFI of
 * its own (it is pure Python); each entry point below names the Python call site it replaces
 * (paths relative to the reference root).  Plain pointers and sizes only: no Python, torch or
 * C++ types cross this boundary.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - all functions return 0 on success, a negative T2FIT_E_* code otherwise; the message of the
 *     last failure on the calling thread is available from t2fit_last_error().
 *   - "dev" pointers are HIP device pointers (e.g. torch.Tensor.data_ptr() on ROCm); "host"
 *     pointers are ordinary memory.  `stream` is a hipStream_t passed as void* (NULL = default).
 *   - echo samples are float32; echo times, tables and tolerances are float64 in MILLISECONDS,
 *     as everywhere in the reference (run_t2mapping.py:369).
 *   - n_vox is the dense voxel count N = Z*Y*X (C order, run_t2mapping.py:411); masked-out voxels
 *     are written as zeros in every map (run_t2mapping.py:415-418).
 */
#ifndef T2FIT_H
#define T2FIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2FIT_ABI_VERSION 4
#define T2FIT_MAX_TE 32

/* error codes */
#define T2FIT_OK 0
#define T2FIT_E_INVALID (-1) /* bad argument (null pointer, size, unknown enum)            */
#define T2FIT_E_HIP (-2)     /* a HIP runtime call failed (no device, OOM, launch failure) */
#define T2FIT_E_BOUNDS (-3)  /* table bounds have lb > ub (scipy raises ValueError there)  */

/* objective, run_t2mapping.py:129-177 */
#define T2FIT_MODEL_GAUSSIAN 0        /* mean (y - k e^{-t/T2})^2                    :141-147 */
#define T2FIT_MODEL_GAUSSIAN_RICIAN 1 /* mean (y - sqrt(k^2 e^{-2t/T2} + s^2))^2     :149-155 */
#define T2FIT_MODEL_RICIAN 2          /* Rician negative log-likelihood with i0e     :157-177 */

/* solver */
#define T2FIT_SOLVER_LBFGSB 0 /* the reference's solver and stop rules: per-lane L-BFGS-B (m=10) with
                                 scipy's bound-aware forward-difference gradient, float64
                                 (run_t2mapping.py:260-286 -> scipy.optimize.minimize)             */
#define T2FIT_SOLVER_LM 1     /* log-linear seed + bounded Levenberg-Marquardt run to convergence of
                                 the same objective and bounds (least-squares models only)          */

#define T2FIT_SOLVER_LOGLIN 2 /* closed-form weighted log-linear regression ln y = ln k - t/T2 (weights
                                 y^2), clipped into the same bounds; 2-parameter model only; one pass,
                                 HBM-bound.  Extension named by BASELINE.json config 2: the reference
                                 has no log-linear routine (parity with it is unpinned)               */

/* arithmetic of the LM solver (the L-BFGS-B solver is always float64) */
#define T2FIT_PREC_F64 0
#define T2FIT_PREC_F32 1

/* memory layout of the echo stack */
#define T2FIT_LAYOUT_TE_MAJOR 0    /* (nTE, N): nTE contiguous volumes, what the reader produces
                                      before np.stack (run_t2mapping.py:377)                 */
#define T2FIT_LAYOUT_VOXEL_MAJOR 1 /* (N, nTE): the reference's reshaped_t2w (:411)          */

/* per-voxel status map values */
#define T2FIT_ST_MASKED 0     /* outside the mask, not fitted                                  */
#define T2FIT_ST_CONVERGED 1  /* solver's convergence test met (scipy result.success == True)  */
#define T2FIT_ST_NOT_CONV 2   /* iteration/evaluation cap or abnormal line-search termination  */
#define T2FIT_ST_NONFINITE 3  /* non-finite sample or objective at the start: x = clipped x0,
                                 nit = 0 (scipy: ABNORMAL, success False; run_t2mapping.py:298) */
#define T2FIT_ST_INFEASIBLE 4 /* data-dependent bounds with lb > ub (no-prior and S(TE0) > 10000):
                                 the reference raises ValueError and aborts the volume; maps get NaN */

/* Fit configuration: the reference's fit_params dict (run_t2mapping.py:29-111) plus the per-call
 * flags of fit_voxel (:120) flattened to a POD.  Fill with t2fit_config_default() first. */
typedef struct t2fit_config {
  int32_t abi_version;        /* T2FIT_ABI_VERSION */
  int32_t model;              /* T2FIT_MODEL_*                                     */
  int32_t solver;             /* T2FIT_SOLVER_*                                    */
  int32_t precision;          /* T2FIT_PREC_* (LM only)                            */
  int32_t n_te;               /* 2..T2FIT_MAX_TE                                   */
  int32_t no_prior;           /* 1: k >= S(TE0), k <= 10000, T2 in [10,2000]  (:243-245)      */
  int32_t norm;               /* 1: divide each voxel's samples by their maximum (:237-238)    */
  int32_t maxls;              /* L-BFGS-B line-search step cap ("maxls", 50 in every table)    */
  int32_t maxiter;            /* iteration cap: scipy default 15000 (L-BFGS-B); LM default 60  */
  int32_t maxfun;             /* objective-evaluation cap (scipy default 15000)                */
  int32_t numpy_legacy;       /* 0 (default): float promotion of numpy >= 2 (NEP 50), what the image's numpy does.
                                 1: value-based casting of numpy < 2 -- the reference freezes numpy 1.26.0
                                 (requirements_frozen.txt:103).  Two places on the path differ: the rician
                                 objective's  np.log(signal) - np.log(sigma**2)  (run_t2mapping.py:169) is then a
                                 FLOAT32 subtraction, and the residual map's prediction  k_map * np.exp(-te / t2_map)
                                 (utils/t2map_utils.py:74-80) is float32 throughout.  (ABI 4; was reserved0.)   */
  int32_t reserved1;
  double te_ms[T2FIT_MAX_TE]; /* echo times [ms], ascending                                     */
  double x0[3];               /* initial_guess (k, T2, sigma); clipped into the bounds as scipy does */
  double lb[3];               /* param_bounds lower (k, T2, sigma)                              */
  double ub[3];               /* param_bounds upper                                             */
  double ftol;                /* L-BFGS-B relative-reduction stop (1e-6 / 1e-2 in the tables)  */
  double gtol;                /* L-BFGS-B projected-gradient stop (scipy default 1e-5 / 1e-2)  */
  double fd_step;             /* absolute forward-difference step, scipy "eps" = 1e-8          */
  double lm_xtol;             /* LM relative step tolerance (0 = default for the precision)    */
  double noprior_k_ub;        /* 10000 (:244) */
  double noprior_t2_lb;       /* 10    (:245) */
  double noprior_t2_ub;       /* 2000  (:245) */
} t2fit_config;

/* Output maps.  t2/k/sigma/res are the reference's four maps (utils/t2map_utils.py:18-29); the
 * others are optional (NULL = not wanted) and replace the per-voxel tuples fit_voxel returns
 * (run_t2mapping.py:312).  All arrays have n_vox elements. */
typedef struct t2fit_maps {
  float *t2;       /* result.x[1]  (run_t2mapping.py:456-458)                                   */
  float *k;        /* result.x[0]                                                               */
  float *sigma;    /* result.x[2]; zeros for the 2-parameter model                              */
  float *res;      /* mean signed residual over TE (utils/t2map_utils.py:84)                    */
  float *r2;       /* optional: 1 - SS_res/SS_tot about the mean (no reference map; extension)  */
  float *fun;      /* optional: final objective value (result.fun, :293)                        */
  int32_t *nit;    /* optional: iterations (result.nit, :292)                                   */
  uint8_t *status; /* optional: T2FIT_ST_*                                                      */
  float *t2_se;    /* optional: standard error of T2 [ms] from the Gauss-Newton covariance
                      s^2 (J^T J)^-1, s^2 = SS_res / (nTE - n_par), J = d model / d (k, T2, sigma) at the
                      float32 map values; NaN when nTE <= n_par or J^T J is singular.  A 95 % confidence
                      interval is T2 +- 1.96 t2_se.  Extension: the reference has no CI map.            */
} t2fit_maps;

/* Fill *cfg with the reference table for (model, low_field): x0, bounds, ftol/gtol/maxls from
 * run_t2mapping.py:38-106, scipy defaults for the rest, solver = L-BFGS-B.  Replaces
 * set_fit_params() (run_t2mapping.py:29-111). */
int t2fit_config_default(t2fit_config *cfg, int model, int low_field);

/* Number of HIP devices visible (0 when there is none; never fails). */
int t2fit_device_count(void);

/* Volume seam, device buffers: replaces run_t2mapping.py:427-461 (Pool.map over fit_voxel, the
 * scatter into maps and compute_residuals) for one (sub,ses).
 *   echoes_dev : float32, layout per `layout`
 *   mask_dev   : uint8 [n_vox] (non-zero = fit), or NULL to fit every voxel
 *   maps       : device pointers
 *   n_vox      : below 2^32 per call (split larger stacks into slabs; voxels are independent)
 * Asynchronous on `stream`; the caller synchronises.  Two launches: the persistent fit kernel
 * (t2, k, sigma and the optional per-voxel extras) and a streaming epilogue (res, r2, t2_se). */
int t2fit_volume_dev(const t2fit_config *cfg, const float *echoes_dev, int layout,
                     const uint8_t *mask_dev, int64_t n_vox, const t2fit_maps *maps, void *stream);

/* Context of the host seam: the state a sequence of numpy-in / numpy-out fits shares (run_t2mapping.py:358 fits one
 * (sub, ses) after the other): three HIP streams, recycled events, a grow-only device arena, two page-locked
 * staging slots per direction and a few copy threads (T2FIT_COPY_THREADS, default 8).  One call at a time per
 * context; contexts of different devices (or several per device) are independent. */
typedef struct t2fit_context t2fit_context;
int t2fit_create(int device, t2fit_context **out);
int t2fit_destroy(t2fit_context *ctx); /* waits for outstanding work; NULL is a no-op */

/* Volume seam, host buffers (numpy arrays): replaces run_t2mapping.py:411-461 for one (sub,ses).  The volume goes
 * through in slabs of about 2 M voxels: while one slab is fitted the worker threads copy the next one's samples from
 * the caller's (pageable) arrays into page-locked staging and the previous one's maps out of it, so the device only
 * DMAs from and to page-locked memory.  The maps do not depend on the split (voxels are independent).  Synchronous:
 * the maps are complete on return. */
int t2fit_context_volume_host(t2fit_context *ctx, const t2fit_config *cfg, const float *echoes, int layout,
                              const uint8_t *mask, int64_t n_vox, const t2fit_maps *maps);

/* The same without a context of the caller's: uses a per-device context that the library creates on first use
 * and keeps for the life of the process.  `device` = HIP device ordinal. */
int t2fit_volume_host(const t2fit_config *cfg, const float *echoes, int layout, const uint8_t *mask,
                      int64_t n_vox, const t2fit_maps *maps, int device);

/* Voxel seam (run_t2mapping.py:120 fit_voxel, batched): fit rows idx[0..n_idx) of the
 * (N, nTE) or (nTE, N) stack and return the per-voxel tuple fields densely packed:
 *   x[n_idx*3] float64 (k, T2, sigma; sigma = 0 for the 2-parameter model), fun[n_idx] float64,
 *   nit[n_idx], status[n_idx].  Host pointers. */
int t2fit_voxels_host(const t2fit_config *cfg, const float *echoes, int layout, int64_t n_vox,
                      const int64_t *idx, int64_t n_idx, double *x, double *fun, int32_t *nit,
                      uint8_t *status, int device);

/* The same with per-iteration traces -- what the reference's callbacks collect into
 * iteration_info (run_t2mapping.py:180-234) and its convergence plots consume:
 *   trace[n_idx * trace_cap * 4] float64: (k, T2, sigma, objective) after each iteration,
 *   trace_len[n_idx]: iterations recorded (<= trace_cap; later ones are dropped). */
int t2fit_voxels_trace_host(const t2fit_config *cfg, const float *echoes, int layout, int64_t n_vox,
                            const int64_t *idx, int64_t n_idx, double *x, double *fun, int32_t *nit,
                            uint8_t *status, int trace_cap, double *trace, int32_t *trace_len,
                            int device);

/* Union mask + flat indices: replaces run_t2mapping.py:383-384,412,421.
 *   masks_dev : n_masks volumes of uint8 [n_vox] each, contiguous (n_masks, n_vox)
 *   mask_out  : uint8 [n_vox], 1 where any input mask is non-zero
 *   idx_out   : int64 [n_vox] capacity; ascending flat indices of the union (np.where order)
 *   count_out : device int64, number of indices written
 * Asynchronous on `stream`. */
int t2fit_union_mask_dev(const uint8_t *masks_dev, int n_masks, int64_t n_vox, uint8_t *mask_out,
                         int64_t *idx_out, int64_t *count_out, void *stream);

/* Residual map alone: replaces compute_residuals (utils/t2map_utils.py:62-89) for maps that are
 * already on the device (all float32 [n_vox]). */
int t2fit_residuals_dev(const t2fit_config *cfg, const float *echoes_dev, int layout,
                        const uint8_t *mask_dev, int64_t n_vox, const float *t2, const float *k,
                        const float *sigma, float *res, void *stream);

/* Phantom ROI statistics: replaces the per-vial nanmean / nanstd loop of save_phantom_csv
 * (utils/t2map_utils.py:43-53) for a map that is on the device.
 *   map_dev   : float32 [n_vox]
 *   label_dev : int32 [n_vox]; voxels labelled 1..n_labels are tallied (n_labels <= 32), others ignored
 *   mean_out, std_out : device float64 [n_labels]: mean and population standard deviation (ddof = 0) over
 *               the non-NaN values of each label, NaN for a label without any (numpy's result)
 *   count_out : device int64 [n_labels] or NULL: number of non-NaN values per label
 * Asynchronous on `stream`; the result does not depend on the launch (fixed summation order). */
int t2fit_label_stats_dev(const float *map_dev, const int32_t *label_dev, int64_t n_vox, int n_labels,
                          double *mean_out, double *std_out, int64_t *count_out, void *stream);

/* Kernel timing for benchmarks (no reference counterpart).  With timing enabled (t2fit_set_timing(1)) every
 * t2fit_volume_dev call of this thread records HIP events around its fit kernel on the launch stream.
 * t2fit_kernel_ms(k): duration in milliseconds of the fit kernel launched k timed calls ago (0 = the most recent;
 * the last 16 are kept), waiting for that launch to finish if it has not; negative when unavailable.
 * t2fit_last_kernel_ms() = t2fit_kernel_ms(0).
 * t2fit_epilogue_ms(k): duration of the streaming epilogue pass (residual map, R^2, T2 standard error) that followed
 * that fit kernel; 0 for the one-pass kernels (closed form, one-shot LM), which have none. */
int t2fit_set_timing(int enabled);
double t2fit_kernel_ms(int launches_ago);
double t2fit_last_kernel_ms(void);
double t2fit_epilogue_ms(int launches_ago);

/* Multi-GPU tuning: the reference-trajectory fit is a persistent kernel whose resident workgroups hold all of every
 * CU's LDS, so a collective's kernel on another stream (RCCL's all-gather of the previous maps) may not become resident
 * before it drains.  `cus` > 0 launches the fit that many CUs' worth of workgroups short (costs cus/256 of its speed):
 * the chip then keeps that many workgroup slots -- their LDS and wave slots -- free; with the one-wave workgroups of the
 * large-volume kernels the dispatcher spreads them over the CUs of its choice (free slots, not whole CUs).  0 = use
 * everything (default; the environment variable T2FIT_RESERVE_CUS sets the initial value).  Process-wide, may be called
 * from any thread at any time (atomic); never changes a result.  Returns the previous setting.  No reference
 * counterpart (the reference is single-process). */
int t2fit_set_reserve_cus(int cus);

const char *t2fit_last_error(void);
int t2fit_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* T2FIT_H */
