"""CPU oracle for the per-voxel T2 fit hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy restatement of the reference's voxel-wise fit.  It exists so the
HIP path can be checked on the GPU box, where /root/reference does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product
package ``fetal_t2mapping_amd`` never does (it fails loudly when the HIP library is missing).

What is restated, and where it lives in the reference (paths relative to /root/reference):

* fit tables ........................ run_t2mapping.py:29-111      -> ``fit_table``
* stack / union mask / flatten ...... run_t2mapping.py:383-386,411-421 -> ``stack_mask_flatten``
* models and objectives ............. run_t2mapping.py:129-177      -> ``objective``
* per-voxel bounded fit ............. run_t2mapping.py:237-312      -> ``fit_voxel``
* result scatter to maps ............ run_t2mapping.py:415-418,449-458 -> ``fit_volume``
* mean signed residual map .......... utils/t2map_utils.py:62-89   -> ``compute_residuals``
* (extension, no reference code) closed-form log-linear 2-parameter fit -> ``loglinear_fit``

Third-party arithmetic on the path that is NOT under /root/reference: ``scipy.optimize.minimize
(method="L-BFGS-B")`` (Byrd, Lu, Nocedal, Zhu 1995; Zhu et al. 1997 L-BFGS-B 3.0 with the
Morales-Nocedal 2011 subspace refinement) and ``scipy.special.i0e``.  The reference pins scipy
1.11.3 (requirements_frozen.txt:144); this image has scipy 1.15.3.  The oracle calls the same scipy
entry point with the same arguments, so on one interpreter it is the reference's computation.

Pinning: ``tests/test_oracle_golden.py`` checks this file bit-for-bit against fixtures produced by
importing the reference's own ``fit_voxel`` / ``compute_residuals`` in the build container
(``tests/golden/make_golden.py``).  A second, tight-tolerance solver (``tight_solve``) gives the
converged bounded minimiser of the same objective; it has no counterpart in the reference and is
used only to measure how far either solver is from the true minimum.
"""
from __future__ import annotations

import copy
import os
from dataclasses import dataclass

import numpy as np
from scipy.optimize import least_squares, minimize
from scipy.special import i0e

MODES = ("gaussian", "gaussian_rician", "rician")

# run_t2mapping.py:38-106 -- (x0, bounds, options) per (low_field, mode).
_LOOSE = {"gtol": 1e-2, "ftol": 1e-2, "maxls": 50, "disp": False}
_TIGHTER = {"ftol": 1e-6, "maxls": 50, "disp": False}
_TABLE = {
    (True, "gaussian"): ([650, 165], [(600, 10000), (10, 600)], _TIGHTER),
    (True, "gaussian_rician"): ([650, 110, 40], [(550, 10000), (10, 600), (2, 1000)], _LOOSE),
    (True, "rician"): ([650, 110, 40], [(550, 900), (10, 600), (2, 1000)], _LOOSE),
    (False, "gaussian"): ([890, 165], [(850, 30000), (10, 600)], _TIGHTER),
    (False, "gaussian_rician"): ([890, 110, 40], [(850, 30000), (30, 600), (2, 1000)], _LOOSE),
    (False, "rician"): ([17, 40, 0.15], [(850, 30000), (30, 600), (7, 200)], _LOOSE),
}


def fit_table(mode: str, low_field: bool) -> dict:
    """Fresh copy of the reference's ``fit_params`` dict (run_t2mapping.py:29-111)."""
    x0, bounds, opts = _TABLE[(bool(low_field), mode)]
    return {
        "initial_guess": list(x0),
        "param_bounds": [tuple(b) for b in bounds],
        "solver": "L-BFGS-B",
        "options": dict(opts),
    }


def stack_mask_flatten(echo_vols, mask_vols):
    """run_t2mapping.py:383-386,411-421.

    ``echo_vols`` / ``mask_vols``: sequences of nTE arrays shaped (Z,Y,X).  Returns
    ``(reshaped_t2w (N,nTE) f32, mask (Z,Y,X) bool, mask_indices (M,) int64)``.
    """
    mask = np.sum(np.stack(list(mask_vols), axis=-1), axis=3) > 0
    t2w = np.stack(list(echo_vols), axis=-1)
    reshaped = np.reshape(t2w, (-1, t2w.shape[-1])).astype(np.float32)
    idx, _ = np.where(np.reshape(mask, (-1, 1)))
    return reshaped, mask, idx


# --- objectives (run_t2mapping.py:129-177); float64 math on float32 samples -------------------
def _obj_gauss(p, te, y):
    k, t2 = p
    r = y - k * np.exp(-te / t2)
    return np.sum(r ** 2) / len(y)


def _obj_gauss_rician(p, te, y):
    k, t2, sigma = p
    r = y - (k ** 2 * np.exp(-2 * te / t2) + sigma ** 2) ** (1 / 2)
    return np.sum(r ** 2) / len(y)


def _obj_rician(p, te, y):
    k, t2, sigma = p
    m = k * np.exp(-te / t2)
    x = (m * y) / (sigma ** 2)
    ll = np.sum((np.log(y) - np.log(sigma ** 2)) - (y ** 2 + m ** 2) / (2 * sigma ** 2)
                + (np.abs(x) + np.log(i0e(x))))
    return -ll


def _obj_rician_legacy(p, te, y):
    """run_t2mapping.py:157-177 as numpy < 2 evaluates it (the reference freezes numpy 1.26.0,
    requirements_frozen.txt:103): ``np.log(signal) - np.log(sigma**2)`` is float32 array - float64 scalar, which
    value-based casting keeps in FLOAT32 (numpy >= 2, NEP 50: float64).  Written with explicit casts so that it
    means the same under either numpy.  Everything else in the objective is float64 under both."""
    k, t2, sigma = p
    m = k * np.exp(-te / t2)
    x = (m * y) / (sigma ** 2)
    a = np.log(y).astype(np.float32) - np.float32(np.log(sigma ** 2))
    ll = np.sum(a - (y ** 2 + m ** 2) / (2 * sigma ** 2) + (np.abs(x) + np.log(i0e(x))))
    return -ll


_OBJ = {"gaussian": _obj_gauss, "gaussian_rician": _obj_gauss_rician, "rician": _obj_rician}
_OBJ_LEGACY = {"gaussian": _obj_gauss, "gaussian_rician": _obj_gauss_rician, "rician": _obj_rician_legacy}


def objective(mode, p, te, y):
    return _OBJ[mode](np.asarray(p, dtype=np.float64), np.asarray(te, dtype=np.float64), np.asarray(y))


class VoxelBoundsError(ValueError):
    """The reference aborts the whole volume here (scipy raises lb>ub)."""


def fit_voxel(voxel, fit, fit_params, TEeffs, reshaped_t2w, prior, norm, want_trace=True, numpy_legacy=False):
    """One voxel, exactly as run_t2mapping.py:237-312 drives scipy.

    Returns ``(x f64[n_par], success, nit, fun, iteration_info)``; like the reference it mutates
    ``fit_params['param_bounds']`` when ``prior`` is False (:243-245).  ``numpy_legacy``: evaluate the objective
    with the promotion rules of numpy < 2 (see ``_obj_rician_legacy``; only the rician objective differs).
    """
    row = reshaped_t2w[voxel, :]
    y = row / np.max(row) if norm else row
    if not prior:
        fit_params["param_bounds"][0] = (reshaped_t2w[voxel, 0], 10000)
        fit_params["param_bounds"][1] = (10, 2000)
    y = np.array(y)
    fun = (_OBJ_LEGACY if numpy_legacy else _OBJ)[fit]
    trace, prev = [], [None]

    def _cb(xk):  # run_t2mapping.py:180-234: f at xk, ||xk - x_prev||
        step = np.nan if prev[0] is None else np.linalg.norm(xk - prev[0])
        prev[0] = xk
        trace.append({"f_val": fun(xk, TEeffs, y), "grad_norm": None, "step_size": step})

    res = minimize(fun, fit_params["initial_guess"], args=(TEeffs, y), method=fit_params["solver"],
                   bounds=fit_params["param_bounds"], options=fit_params["options"], jac=False,
                   callback=_cb if want_trace else None)
    return res.x, res.success, res.nit, res.fun, trace


def compute_residuals(reshaped_t2w, TEeffs, fit, norm, k_map, t2_map, sigma_map, res_map, mask_indices,
                      numpy_legacy=False):
    """utils/t2map_utils.py:62-89 on flat maps (the caller reshapes).  numpy-2 promotion: the
    np.float64 TE scalar makes the prediction float64, which is then stored as float32.  ``numpy_legacy``: the
    numpy < 2 rule instead -- the scalar takes the float32 of the maps and the whole prediction is float32."""
    pred = np.zeros_like(reshaped_t2w)
    with np.errstate(all="ignore"):
        for i, te in enumerate(TEeffs):
            if numpy_legacy:
                if fit == "gaussian":
                    pred[:, i] = k_map * np.exp(np.float32(-te) / t2_map)
                else:
                    pred[:, i] = np.sqrt(k_map ** 2 * np.exp(np.float32(-2 * te) / t2_map) + sigma_map ** 2)
                continue
            if fit == "gaussian":
                pred[:, i] = k_map * np.exp(-te / t2_map)
            else:
                pred[:, i] = (k_map ** 2 * np.exp(-2 * te / t2_map) + sigma_map ** 2) ** (1 / 2)
        data = reshaped_t2w / np.max(reshaped_t2w, axis=1, keepdims=True) if norm else reshaped_t2w
        resid = data - pred
    res_map[mask_indices] = np.sum(resid[mask_indices], axis=1) / len(TEeffs)
    return res_map


@dataclass
class VolumeFit:
    t2: np.ndarray
    k: np.ndarray
    sigma: np.ndarray
    res: np.ndarray
    success: np.ndarray
    nit: np.ndarray
    fun: np.ndarray


def _fit_chunk(args):
    idx, fit, fit_params, te, data, prior, norm = args
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    fp = copy.deepcopy(fit_params)
    out = []
    for v in idx:
        x, ok, nit, f, _ = fit_voxel(int(v), fit, fp, te, data, prior, norm, want_trace=False)
        out.append((x, ok, nit, f))
    return out


def fit_volume(reshaped_t2w, mask_indices, TEeffs, fit, fit_params, prior=True, norm=False, pool=None):
    """run_t2mapping.py:415-461 restated: fit every masked voxel, scatter, residual map.

    ``pool``: optional ``multiprocessing.Pool`` (the reference hard-codes 20 processes, :442).
    """
    n = reshaped_t2w.shape[0]
    te = np.asarray(TEeffs, dtype=np.float64)
    mask_indices = np.asarray(mask_indices)
    if pool is None:
        rows = _fit_chunk((mask_indices, fit, fit_params, te, reshaped_t2w, prior, norm))
    else:
        nchunk = max(1, min(len(mask_indices), 4 * pool._processes))
        chunks = [c for c in np.array_split(mask_indices, nchunk) if len(c)]
        rows = [r for part in pool.map(_fit_chunk, [(c, fit, fit_params, te, reshaped_t2w, prior, norm)
                                                    for c in chunks]) for r in part]
    t2 = np.zeros(n, np.float32)
    k = np.zeros(n, np.float32)
    sg = np.zeros(n, np.float32)
    res = np.zeros(n, np.float32)
    if len(rows):
        xs = np.array([r[0] for r in rows])
        t2[mask_indices] = xs[:, 1].astype(np.float32)
        k[mask_indices] = xs[:, 0].astype(np.float32)
        if fit != "gaussian":
            sg[mask_indices] = xs[:, 2].astype(np.float32)
    res = compute_residuals(reshaped_t2w, te, fit, norm, k, t2, sg, res, mask_indices)
    return VolumeFit(t2, k, sg, res,
                     np.array([r[1] for r in rows], bool), np.array([r[2] for r in rows], np.int32),
                     np.array([r[3] for r in rows], np.float64))


def voxel_bounds(fit_params, y0, prior):
    """Bounds a voxel is fitted under (run_t2mapping.py:243-245) as float64 (lb, ub) arrays."""
    b = [tuple(map(float, t)) for t in fit_params["param_bounds"]]
    if not prior:
        b[0] = (float(y0), 10000.0)
        b[1] = (10.0, 2000.0)
    lb = np.array([t[0] for t in b])
    ub = np.array([t[1] for t in b])
    return lb, ub


# --- closed-form log-linear fit (no reference counterpart) -----------------------------------------
def loglinear_fit(rows, TEeffs, fit_params, prior=True):
    """Weighted log-linear regression ``ln y = ln k - t/T2`` with weights ``y**2`` over the positive
    samples of each row, clipped into the voxel's bounds: the definition T2FIT_SOLVER_LOGLIN implements.

    BASELINE.json configuration 2 names a "2-param log-linear fit"; the reference has none
    (run_t2mapping.py:260-272 runs L-BFGS-B for the 2-parameter model too), so this is the oracle of
    an extension, not a restatement: parity with the reference is unpinned for it.  float64 throughout.
    Returns ``(x (M,2) [k, T2], ok (M,) bool)``; rows without two positive samples get the clipped
    table start point and ``ok`` False.
    """
    rows = np.asarray(rows, np.float64)
    te = np.asarray(TEeffs, np.float64)
    x = np.zeros((rows.shape[0], 2))
    ok = np.zeros(rows.shape[0], bool)
    x0 = np.asarray(fit_params["initial_guess"], np.float64)[:2]
    for v, y in enumerate(rows):
        lb, ub = voxel_bounds(fit_params, np.float32(y[0]), prior)
        pos = y > 0
        w = np.where(pos, y * y, 0.0)
        with np.errstate(all="ignore"):
            l = np.where(pos, np.log(np.where(pos, y, 1.0)), 0.0)
        sw = w.sum()
        if pos.sum() < 2 or not sw > 0 or not np.all(np.isfinite(y)):
            x[v] = np.clip(x0, lb[:2], ub[:2])
            continue
        tbar, lbar = (w * te).sum() / sw, (w * l).sum() / sw
        sxx = (w * (te - tbar) ** 2).sum()
        if not sxx > 0:
            x[v] = np.clip(x0, lb[:2], ub[:2])
            continue
        slope = (w * (te - tbar) * (l - lbar)).sum() / sxx   # centred form: no cancellation
        icpt = lbar - slope * tbar
        t2 = -1.0 / slope if slope < 0 else np.inf
        x[v] = np.clip([np.exp(min(icpt, 700.0)), t2], lb[:2], ub[:2])
        ok[v] = True
    return x, ok


# --- tight-tolerance bounded minimiser of the same objective (no reference counterpart) --------
def _resid_fun(mode, te, y):
    if mode == "gaussian":
        return lambda p: y - p[0] * np.exp(-te / p[1])
    return lambda p: y - np.sqrt(p[0] ** 2 * np.exp(-2 * te / p[1]) + p[2] ** 2)


def tight_solve(mode, te, y, lb, ub, starts):
    """Converged bounded minimiser of the mode's objective: best of several starts.

    Least-squares modes: scipy ``least_squares`` (TRF, 1e-15 tolerances).  ``rician``: L-BFGS-B
    with 1e-15/1e-12 tolerances and 3-point differences.  Returns ``(x, f)``.
    """
    te = np.asarray(te, np.float64)
    y_native = np.asarray(y)          # rician: the reference evaluates log(y), y**2 in the data's dtype
    y = np.asarray(y, np.float64)
    if np.any(lb > ub) or not np.all(np.isfinite(y)):
        return np.full(len(lb), np.nan), np.nan
    best_x, best_f = None, np.inf
    for s in starts:
        s = np.clip(np.asarray(s, np.float64), lb, ub)
        try:
            if mode == "rician":
                with np.errstate(all="ignore"):
                    r = minimize(_obj_rician, s, args=(te, y_native), method="L-BFGS-B", jac="3-point",
                                 bounds=list(zip(lb, ub)),
                                 options={"ftol": 1e-15, "gtol": 1e-12, "maxls": 50, "maxiter": 2000})
                x, f = r.x, r.fun
            else:
                # interior-only TRF needs lb<ub; nudge starts off the faces
                span = ub - lb
                s = np.clip(s, lb + 1e-9 * span, ub - 1e-9 * span)
                r = least_squares(_resid_fun(mode, te, y), s, bounds=(lb, ub), method="trf",
                                  x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=2000)
                x = r.x
                f = float(np.sum(r.fun ** 2) / len(y))
        except Exception:
            continue
        if np.isfinite(f) and f < best_f:
            best_x, best_f = x, f
    if best_x is None:
        return np.full(len(lb), np.nan), np.nan
    return best_x, best_f
