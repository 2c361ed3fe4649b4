"""Host-side mirror of the reference's fitting interface, backed by the HIP library.

Same names, argument meaning and error behaviour as the reference's hot path:

=============================  ================================================================
this module                    reference (paths relative to /root/reference)
=============================  ================================================================
``set_fit_params(args)``       run_t2mapping.py:29-111
``fit_voxel(...)``             run_t2mapping.py:120-312 (one voxel; same 5-tuple)
``fit_voxels(...)``            the ``Pool.map`` over ``fit_voxel`` (:430-443), batched
``stack_mask_flatten(...)``    run_t2mapping.py:383-386,411-421
``fit_volume(...)``            run_t2mapping.py:411-461 (flatten, fit, scatter, residual map)
``compute_residuals(...)``     utils/t2map_utils.py:62-89
=============================  ================================================================

Python here only marshals buffers; all arithmetic happens in libt2fit_hip.so through the C ABI of
include/t2fit.h.  torch is used for device buffers and streams, nothing else.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _abi
from ._lib import check, load, require_gpu

# scipy defaults that apply when a table omits an option (scipy.optimize._lbfgsb_py._minimize_lbfgsb)
_SCIPY_DEFAULTS = {"ftol": 2.2204460492503131e-09, "gtol": 1e-5, "eps": 1e-8, "maxfun": 15000,
                   "maxiter": 15000, "maxls": 20, "maxcor": 10}


# --------------------------------------------------------------------------------------------
# fit tables
# --------------------------------------------------------------------------------------------
def fit_table(fit: str, low_field: bool) -> dict:
    """The reference's ``fit_params`` dict for (fit, field), read from the library's tables."""
    cfg = _abi.T2FitConfig()
    check(load().t2fit_config_default(C.byref(cfg), _abi.MODELS[fit], int(bool(low_field))))
    n_par = 2 if fit == "gaussian" else 3

    def _num(v):  # the reference writes ints where it can; keep printing identical
        return int(v) if float(v).is_integer() else float(v)

    options = {"ftol": cfg.ftol, "maxls": cfg.maxls, "disp": False}
    if fit != "gaussian":
        options = {"gtol": cfg.gtol, "ftol": cfg.ftol, "maxls": cfg.maxls, "disp": False}
    return {
        "initial_guess": [_num(cfg.x0[j]) for j in range(n_par)],
        "param_bounds": [(_num(cfg.lb[j]), _num(cfg.ub[j])) for j in range(n_par)],
        "solver": "L-BFGS-B",
        "options": options,
    }


def set_fit_params(args):
    """run_t2mapping.py:29-111: ``args`` carries gaussian/gaussian_rician/rician, lf/hf, norm."""
    if getattr(args, "norm", False):
        print("Error: Normalization is set to true though no parameters where defined yet. "
              "Please modify set_fit_params to manage.")
        raise SystemExit(1)
    fit = "gaussian" if args.gaussian else "gaussian_rician" if args.gaussian_rician else "rician"
    if not (args.lf or args.hf):
        raise SystemExit(1)
    return fit, fit_table(fit, bool(args.lf))


def make_config(fit: str, fit_params: dict, TEeffs, prior: bool = True, norm: bool = False,
                solver: str = "lbfgsb", precision: str = "f64", numpy_legacy: bool = False) -> _abi.T2FitConfig:
    """Flatten (fit, fit_params, TEeffs, prior, norm) into the ABI struct.  ``numpy_legacy``: reproduce the reference
    as it runs under the numpy < 2 it freezes (requirements_frozen.txt:103) instead of under numpy >= 2: float32
    log term of the rician objective (run_t2mapping.py:169), float32 prediction of the residual map
    (utils/t2map_utils.py:74-80)."""
    if fit not in _abi.MODELS:
        raise ValueError(f"unknown fit {fit!r}")
    if fit_params.get("solver", "L-BFGS-B") != "L-BFGS-B":
        raise ValueError("only the reference's solver 'L-BFGS-B' is defined for fit_params['solver']")
    te = np.asarray(TEeffs, dtype=np.float64).ravel()
    if not 2 <= te.size <= _abi.MAX_TE:
        raise ValueError(f"need 2..{_abi.MAX_TE} echo times, got {te.size}")
    cfg = _abi.T2FitConfig()
    check(load().t2fit_config_default(C.byref(cfg), _abi.MODELS[fit], 1))
    n_par = 2 if fit == "gaussian" else 3
    x0 = list(fit_params["initial_guess"])
    bounds = list(fit_params["param_bounds"])
    if len(x0) != n_par:
        raise ValueError("length of initial_guess does not match the model")
    if len(bounds) != n_par:
        raise ValueError("length of x0 != length of bounds")  # scipy's message
    for j in range(3):
        cfg.x0[j] = float(x0[j]) if j < n_par else 0.0
        cfg.lb[j] = float(bounds[j][0]) if j < n_par else 0.0
        cfg.ub[j] = float(bounds[j][1]) if j < n_par else 0.0
    opts = dict(_SCIPY_DEFAULTS)
    opts.update({k: v for k, v in fit_params.get("options", {}).items() if k not in ("disp", "iprint")})
    if int(opts["maxcor"]) != 10:
        raise NotImplementedError("the lane solver keeps scipy's default maxcor=10 corrections")
    if not opts["maxls"] > 0:
        raise ValueError("maxls must be positive.")
    cfg.ftol, cfg.gtol, cfg.fd_step = float(opts["ftol"]), float(opts["gtol"]), float(opts["eps"])
    cfg.maxls, cfg.maxiter, cfg.maxfun = int(opts["maxls"]), int(opts["maxiter"]), int(opts["maxfun"])
    cfg.n_te = te.size
    for i in range(_abi.MAX_TE):
        cfg.te_ms[i] = float(te[i]) if i < te.size else 0.0
    cfg.no_prior = int(not prior)
    cfg.norm = int(bool(norm))
    cfg.numpy_legacy = int(bool(numpy_legacy))
    cfg.solver = _abi.SOLVERS[solver]
    cfg.precision = _abi.PRECISIONS[precision]
    if cfg.solver == _abi.SOLVER_LOGLIN and fit != "gaussian":
        raise ValueError("solver 'loglin' is the closed form of the 2-parameter 'gaussian' fit only")
    if cfg.solver == _abi.SOLVER_LM:
        cfg.maxiter = 0  # library default for LM
    return cfg


# --------------------------------------------------------------------------------------------
# stack / mask / flatten
# --------------------------------------------------------------------------------------------
def stack_mask_flatten(echo_vols: Sequence[np.ndarray], mask_vols: Sequence[np.ndarray], device: int = 0):
    """run_t2mapping.py:383-386,411-421 without the (Z,Y,X,nTE) transpose.

    Returns ``(echoes (nTE,N) float32 torch tensor on the GPU, mask (Z,Y,X) bool ndarray,
    mask_indices (M,) int64 ndarray)``; mask and indices are computed on the device and are
    bit-identical to ``np.sum(stack(masks),axis=3) > 0`` / ``np.where(...)[0]``.
    """
    import torch

    lib = require_gpu()
    shape = tuple(np.asarray(echo_vols[0]).shape)
    n = int(np.prod(shape))
    dev = torch.device("cuda", device)
    echoes = torch.empty((len(echo_vols), n), dtype=torch.float32, device=dev)
    for i, v in enumerate(echo_vols):
        echoes[i] = torch.from_numpy(np.ascontiguousarray(v).astype(np.float32, copy=False).reshape(-1)).to(dev)
    masks = torch.empty((len(mask_vols), n), dtype=torch.uint8, device=dev)
    for i, m in enumerate(mask_vols):
        masks[i] = torch.from_numpy((np.asarray(m) != 0).astype(np.uint8).reshape(-1)).to(dev)
    mask_d, idx_d, cnt_d = union_mask_dev(masks)
    count = int(cnt_d.item())
    return echoes, mask_d.cpu().numpy().astype(bool).reshape(shape), idx_d[:count].cpu().numpy()


def union_mask_dev(masks):
    """(n_masks, N) uint8 cuda tensor -> (mask uint8 [N], idx int64 [N] (first `count` valid), count)."""
    import torch

    lib = require_gpu()
    assert masks.is_cuda and masks.dtype == torch.uint8 and masks.is_contiguous() and masks.dim() == 2
    n = masks.shape[1]
    mask = torch.empty(n, dtype=torch.uint8, device=masks.device)
    idx = torch.empty(n, dtype=torch.int64, device=masks.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=masks.device)
    with torch.cuda.device(masks.device):
        st = torch.cuda.current_stream().cuda_stream
        check(lib.t2fit_union_mask_dev(masks.data_ptr(), masks.shape[0], n, mask.data_ptr(), idx.data_ptr(),
                                       cnt.data_ptr(), C.c_void_p(st)))
    return mask, idx, cnt


def label_stats(map_, label, n_labels: int, device: int = 0):
    """Per-label ``(nanmean, nanstd, count)`` of a map on the GPU: the loop of ``save_phantom_csv``
    (utils/t2map_utils.py:43-53).  ``map_``: float32 array or CUDA tensor of any shape; ``label``: integer
    array/tensor of the same shape, vials numbered 1..n_labels.  Returns float64 / int64 numpy arrays."""
    import torch

    lib = require_gpu()
    dev = map_.device if type(map_).__module__.startswith("torch") and map_.is_cuda else torch.device("cuda", device)
    m = (map_ if type(map_).__module__.startswith("torch") else torch.from_numpy(np.ascontiguousarray(map_, np.float32)))
    m = m.to(dev, torch.float32).contiguous().reshape(-1)
    lab = label if type(label).__module__.startswith("torch") else torch.from_numpy(np.ascontiguousarray(label).astype(np.int32))
    lab = lab.to(dev, torch.int32).contiguous().reshape(-1)
    if lab.numel() != m.numel():
        raise ValueError("label shape does not match the map")
    mean = torch.empty(n_labels, dtype=torch.float64, device=dev)
    std = torch.empty(n_labels, dtype=torch.float64, device=dev)
    cnt = torch.empty(n_labels, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(lib.t2fit_label_stats_dev(m.data_ptr(), lab.data_ptr(), m.numel(), int(n_labels), mean.data_ptr(),
                                        std.data_ptr(), cnt.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return mean.cpu().numpy(), std.cpu().numpy(), cnt.cpu().numpy()


# --------------------------------------------------------------------------------------------
# volume seam
# --------------------------------------------------------------------------------------------
@dataclass
class T2Maps:
    """The reference's four maps (utils/t2map_utils.py:18-29) plus optional per-voxel extras."""
    t2: object
    k: object
    sigma: object
    res: object
    r2: Optional[object] = None
    fun: Optional[object] = None
    nit: Optional[object] = None
    status: Optional[object] = None
    t2_se: Optional[object] = None  # standard error of T2 (extension; 95 % CI = T2 +- 1.96 t2_se)

    def success(self):
        """scipy ``result.success`` per voxel (False outside the mask)."""
        return None if self.status is None else (self.status == _abi.ST_CONVERGED)


def _layout_of(echoes_shape, n_te, layout):
    if layout in ("te_major", _abi.LAYOUT_TE_MAJOR):
        if echoes_shape[0] != n_te:
            raise ValueError(f"te_major echoes need shape (nTE, ...): got {tuple(echoes_shape)} for nTE={n_te}")
        return _abi.LAYOUT_TE_MAJOR, tuple(echoes_shape[1:])
    if layout in ("voxel_major", _abi.LAYOUT_VOXEL_MAJOR):
        if echoes_shape[-1] != n_te:
            raise ValueError(f"voxel_major echoes need shape (..., nTE): got {tuple(echoes_shape)} for nTE={n_te}")
        return _abi.LAYOUT_VOXEL_MAJOR, tuple(echoes_shape[:-1])
    raise ValueError(f"unknown layout {layout!r}")


def fit_volume(echoes, mask, TEeffs, fit, fit_params, prior=True, norm=False, *, layout="te_major",
               solver="lbfgsb", precision="f64", extras=False, strict=True, device=0, out: T2Maps = None,
               numpy_legacy=False):
    """Fit every masked voxel and return the maps (run_t2mapping.py:411-461).

    ``echoes``: float32 ``(nTE, Z, Y, X)`` (``layout='te_major'``, the per-TE volumes as read) or
    ``(Z, Y, X, nTE)`` (``'voxel_major'``, the reference's ``t2w``); numpy array or CUDA torch tensor.
    ``mask``: same spatial shape, non-zero = fit, or None.  Returns :class:`T2Maps` shaped ``(Z,Y,X)``
    -- numpy for numpy input (host entry point), torch for torch input (device entry point,
    asynchronous on the current stream).  ``strict``: raise ValueError, as the reference's scipy call
    does, if a voxel's data-dependent bounds are infeasible (numpy path; the torch path never syncs).
    """
    cfg = make_config(fit, fit_params, TEeffs, prior, norm, solver, precision, numpy_legacy)
    lib = require_gpu()
    lay, spatial = _layout_of(echoes.shape, cfg.n_te, layout)
    n = int(np.prod(spatial)) if len(spatial) else 1
    is_torch = type(echoes).__module__.startswith("torch")
    maps = _abi.T2FitMaps()
    if is_torch:
        import torch

        if not (echoes.is_cuda and echoes.dtype == torch.float32 and echoes.is_contiguous()):
            raise ValueError("torch echoes must be a contiguous float32 CUDA tensor")
        dev = echoes.device
        if mask is not None:
            if not (mask.is_cuda and mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == n):
                raise ValueError("torch mask must be a contiguous uint8 CUDA tensor of the spatial shape")
        if out is None:
            f32 = lambda: torch.empty(spatial, dtype=torch.float32, device=dev)  # noqa: E731
            out = T2Maps(f32(), f32(), f32(), f32())
            if extras:
                out.r2, out.fun, out.t2_se = f32(), f32(), f32()
                out.nit = torch.empty(spatial, dtype=torch.int32, device=dev)
                out.status = torch.empty(spatial, dtype=torch.uint8, device=dev)
        for name in ("t2", "k", "sigma", "res", "r2", "fun", "nit", "status", "t2_se"):
            t = getattr(out, name)
            if t is not None:  # the library writes raw bytes of this type through the pointer: check before it does
                want = {"nit": torch.int32, "status": torch.uint8}.get(name, torch.float32)
                if not (torch.is_tensor(t) and t.dtype == want and t.device == dev and t.is_contiguous() and t.numel() == n):
                    raise ValueError(f"out.{name} must be a contiguous {str(want).split('.')[-1]} tensor on {dev} with {n} elements")
            elif name in ("t2", "k", "sigma", "res"):
                raise ValueError(f"out.{name} is required")
            setattr(maps, name, None if t is None else t.data_ptr())
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            check(lib.t2fit_volume_dev(C.byref(cfg), echoes.data_ptr(), lay,
                                       None if mask is None else mask.data_ptr(), n, C.byref(maps),
                                       C.c_void_p(st)))
        return out
    e = np.ascontiguousarray(echoes, dtype=np.float32)
    m = None
    if mask is not None:
        m = np.asarray(mask)
        # the kernels test mask != 0 themselves: one-byte masks go in as they are
        m = np.ascontiguousarray(m).view(np.uint8) if m.dtype.itemsize == 1 else np.ascontiguousarray(m != 0, dtype=np.uint8)
        if m.size != n:
            raise ValueError("mask shape does not match the echoes")
    if out is None:  # (callers that fit one volume after the other may hand the previous T2Maps back in as `out`)
        f32 = lambda: _new_map(spatial, np.float32)  # noqa: E731
        out = T2Maps(f32(), f32(), f32(), f32())
        if extras:
            out.r2, out.fun, out.nit, out.t2_se = f32(), f32(), _new_map(spatial, np.int32), f32()
    want_status = out.status is not None
    if out.status is None:
        out.status = _new_map(spatial, np.uint8)
    for name in ("t2", "k", "sigma", "res", "r2", "fun", "nit", "status", "t2_se"):
        a = getattr(out, name)
        want = {"nit": np.int32, "status": np.uint8}.get(name, np.float32)  # what the library writes through the pointer
        if a is not None and not (isinstance(a, np.ndarray) and a.dtype == want and a.flags.c_contiguous
                                  and a.flags.writeable and a.size == n):
            raise ValueError(f"out.{name} must be a writable C-contiguous {np.dtype(want).name} numpy array with {n} elements")
        if a is None and name in ("t2", "k", "sigma", "res"):
            raise ValueError(f"out.{name} is required")
        setattr(maps, name, None if a is None else a.ctypes.data)
    check(lib.t2fit_volume_host(C.byref(cfg), e.ctypes.data, lay, None if m is None else m.ctypes.data, n,
                                C.byref(maps), int(device)))
    if strict and np.any(out.status == _abi.ST_INFEASIBLE):
        bad = int(np.flatnonzero(out.status.reshape(-1) == _abi.ST_INFEASIBLE)[0])
        raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound. "
                         f"(voxel {bad}: S(TE0) exceeds the no-prior upper bound)")
    if not extras and not want_status:
        out.status = None
    return out


_libc = None


def _new_map(shape, dtype):
    """A fresh output array.  Large ones are advised to use transparent huge pages: the library's copy threads touch
    every page of a new map for the first time, and 67 MB in 4 KiB pages are 16 384 page faults per map and call."""
    global _libc
    a = np.empty(shape, dtype)
    if a.nbytes >= (8 << 20):
        try:
            if _libc is None:
                _libc = C.CDLL(None, use_errno=True)
            huge = 2 << 20
            lo = (a.ctypes.data + huge - 1) & ~(huge - 1)
            hi = (a.ctypes.data + a.nbytes) & ~(huge - 1)
            if hi > lo:
                _libc.madvise(C.c_void_p(lo), C.c_size_t(hi - lo), 14)  # MADV_HUGEPAGE; failure is harmless
        except (OSError, AttributeError):
            pass
    return a


# --------------------------------------------------------------------------------------------
# voxel seam
# --------------------------------------------------------------------------------------------
def fit_voxels(indices, fit, fit_params, TEeffs, reshaped_t2w, prior, norm, *, solver="lbfgsb",
               precision="f64", device=0, numpy_legacy=False):
    """Batched ``fit_voxel``: rows ``indices`` of the (N, nTE) float32 stack.

    Returns ``(x (M,n_par) f64, success (M,) bool, nit (M,) int32, fun (M,) f64, status (M,) u8)``.
    """
    cfg = make_config(fit, fit_params, TEeffs, prior, norm, solver, precision, numpy_legacy)
    lib = require_gpu()
    data = np.ascontiguousarray(reshaped_t2w, dtype=np.float32)
    if data.ndim != 2 or data.shape[1] != cfg.n_te:
        raise ValueError("reshaped_t2w must be (N, nTE)")
    idx = np.ascontiguousarray(np.atleast_1d(indices), dtype=np.int64)
    m = idx.size
    x = np.zeros((m, 3))
    fun = np.zeros(m)
    nit = np.zeros(m, np.int32)
    st = np.zeros(m, np.uint8)
    check(lib.t2fit_voxels_host(C.byref(cfg), data.ctypes.data, _abi.LAYOUT_VOXEL_MAJOR, data.shape[0],
                                idx.ctypes.data, m, x.ctypes.data, fun.ctypes.data, nit.ctypes.data,
                                st.ctypes.data, int(device)))
    n_par = 2 if fit == "gaussian" else 3
    return x[:, :n_par], st == _abi.ST_CONVERGED, nit, fun, st


def fit_voxels_trace(indices, fit, fit_params, TEeffs, reshaped_t2w, prior, norm, *, trace_cap=64, solver="lbfgsb",
                     precision="f64", device=0, numpy_legacy=False):
    """``fit_voxels`` plus, per voxel, the reference's ``iteration_info`` (run_t2mapping.py:180-234): a
    list of ``{'f_val', 'grad_norm': None, 'step_size'}`` dicts, one per iteration (at most ``trace_cap``)."""
    cfg = make_config(fit, fit_params, TEeffs, prior, norm, solver, precision, numpy_legacy)
    lib = require_gpu()
    data = np.ascontiguousarray(reshaped_t2w, dtype=np.float32)
    idx = np.ascontiguousarray(np.atleast_1d(indices), dtype=np.int64)
    m = idx.size
    x, fun = np.zeros((m, 3)), np.zeros(m)
    nit, st = np.zeros(m, np.int32), np.zeros(m, np.uint8)
    tr, tl = np.zeros((m, trace_cap, 4)), np.zeros(m, np.int32)
    check(lib.t2fit_voxels_trace_host(C.byref(cfg), data.ctypes.data, _abi.LAYOUT_VOXEL_MAJOR, data.shape[0],
                                      idx.ctypes.data, m, x.ctypes.data, fun.ctypes.data, nit.ctypes.data,
                                      st.ctypes.data, int(trace_cap), tr.ctypes.data, tl.ctypes.data, int(device)))
    n_par = 2 if fit == "gaussian" else 3
    infos = []
    for r in range(m):
        pts = tr[r, : tl[r]]
        steps = np.r_[np.nan, np.linalg.norm(np.diff(pts[:, :n_par], axis=0), axis=1)] if len(pts) else []
        infos.append([{"f_val": float(p[3]), "grad_norm": None, "step_size": float(s)} for p, s in zip(pts, steps)])
    return x[:, :n_par], st == _abi.ST_CONVERGED, nit, fun, st, infos


def fit_voxel(voxel, fit, fit_params, TEeffs, reshaped_t2w, prior, norm, want_trace=True, **kw):
    """run_t2mapping.py:120-312 for one voxel: ``(params, success, nit, final_error, iteration_info)``.

    Like the reference, a voxel whose no-prior bounds are infeasible raises ValueError, and
    ``fit_params['param_bounds']`` is rewritten in place when ``prior`` is False (:243-245).
    ``iteration_info`` holds the objective value and step length of every iteration, as the
    reference's callbacks record them.
    """
    if not prior:
        fit_params["param_bounds"][0] = (reshaped_t2w[voxel, 0], 10000)
        fit_params["param_bounds"][1] = (10, 2000)
        if fit_params["param_bounds"][0][0] > 10000:
            raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
    if want_trace:
        x, ok, nit, fun, st, infos = fit_voxels_trace([voxel], fit, fit_params, TEeffs, reshaped_t2w, prior, norm, **kw)
    else:
        x, ok, nit, fun, st = fit_voxels([voxel], fit, fit_params, TEeffs, reshaped_t2w, prior, norm, **kw)
        infos = [[]]
    if not ok[0]:
        print(f"FAIL : Optimization failed for voxel {voxel}: status {int(st[0])}")
        print("Objective function value at optimum:", fun[0])
        print("params", x[0])
    return x[0], bool(ok[0]), int(nit[0]), float(fun[0]), infos[0]


# --------------------------------------------------------------------------------------------
# residual map
# --------------------------------------------------------------------------------------------
def compute_residuals(reshaped_t2w, TEeffs, fit, norm, k_map, t2_map, sigma_map, res_map, mask_indices, mask,
                      device=0, numpy_legacy=False):
    """utils/t2map_utils.py:62-89 with the reference's signature; evaluated on the GPU."""
    import torch

    lib = require_gpu()
    data = np.ascontiguousarray(reshaped_t2w, dtype=np.float32)
    n, n_te = data.shape
    cfg = make_config(fit, fit_table(fit, True), TEeffs, True, norm, numpy_legacy=numpy_legacy)
    dev = torch.device("cuda", device)
    e = torch.from_numpy(data).to(dev)
    sel = torch.zeros(n, dtype=torch.uint8, device=dev)
    sel[torch.from_numpy(np.asarray(mask_indices, dtype=np.int64)).to(dev)] = 1
    t2 = torch.from_numpy(np.ascontiguousarray(t2_map, np.float32).reshape(-1)).to(dev)
    k = torch.from_numpy(np.ascontiguousarray(k_map, np.float32).reshape(-1)).to(dev)
    sg = torch.from_numpy(np.ascontiguousarray(sigma_map, np.float32).reshape(-1)).to(dev)
    res = torch.empty(n, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        check(lib.t2fit_residuals_dev(C.byref(cfg), e.data_ptr(), _abi.LAYOUT_VOXEL_MAJOR, sel.data_ptr(), n,
                                      t2.data_ptr(), k.data_ptr(), sg.data_ptr(), res.data_ptr(), C.c_void_p(st)))
    out = np.asarray(res_map, dtype=np.float32).reshape(-1).copy()
    r = res.cpu().numpy()
    mi = np.asarray(mask_indices, dtype=np.int64)
    out[mi] = r[mi]
    return out.reshape(np.asarray(mask).shape[:3])
