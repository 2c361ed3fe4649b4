"""Loader for the host-side lane simulator (test infrastructure, see hostsim.cpp)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from fetal_t2mapping_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libt2fit_hostsim.so")
SRC = os.path.join(HERE, "hostsim.cpp")
CSRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "fetal_t2mapping_amd", "csrc")


def build(force: bool = False) -> str:
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "t2fit.h"))
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-o", SO, SRC])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.hostsim_config_default.argtypes = [C.POINTER(_abi.T2FitConfig), C.c_int, C.c_int]
        _lib.hostsim_fit_rows.argtypes = [C.POINTER(_abi.T2FitConfig), C.c_void_p, C.c_int64] + [C.c_void_p] * 6
    return _lib


def config(mode: str, low_field: bool, te, prior=True, norm=False, solver="lbfgsb", precision="f64", numpy_legacy=False):
    cfg = _abi.T2FitConfig()
    assert lib().hostsim_config_default(C.byref(cfg), _abi.MODELS[mode], int(low_field)) == 0
    te = np.asarray(te, np.float64)
    cfg.n_te = len(te)
    for i, t in enumerate(te):
        cfg.te_ms[i] = t
    cfg.no_prior = int(not prior)
    cfg.norm = int(norm)
    cfg.numpy_legacy = int(numpy_legacy)
    cfg.solver = _abi.SOLVERS[solver]
    cfg.precision = _abi.PRECISIONS[precision]
    if solver == "lm":
        cfg.maxiter = 0
    return cfg


def fit_rows(cfg, rows):
    rows = np.ascontiguousarray(rows, np.float32)
    n = rows.shape[0]
    x = np.zeros((n, 3))
    fun = np.zeros(n)
    nit = np.zeros(n, np.int32)
    st = np.zeros(n, np.uint8)
    res = np.zeros(n, np.float32)
    r2 = np.zeros(n, np.float32)
    rc = lib().hostsim_fit_rows(C.byref(cfg), rows.ctypes.data, n, x.ctypes.data, fun.ctypes.data,
                                nit.ctypes.data, st.ctypes.data, res.ctypes.data, r2.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"hostsim_fit_rows rc={rc}")
    return {"x": x, "fun": fun, "nit": nit, "status": st, "res": res, "r2": r2}


def trace_row(cfg, row, cap=200):
    row = np.ascontiguousarray(row, np.float32)
    tr = np.zeros((cap, 4))
    n = C.c_int(0)
    x = np.zeros(3)
    fun = C.c_double(0)
    nit = C.c_int32(0)
    st = C.c_uint8(0)
    L = lib()
    L.hostsim_trace_row.argtypes = [C.POINTER(_abi.T2FitConfig), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                    C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)]
    rc = L.hostsim_trace_row(C.byref(cfg), row.ctypes.data, tr.ctypes.data, cap, C.byref(n), x.ctypes.data,
                             C.byref(fun), C.byref(nit), C.byref(st))
    return rc, tr[: n.value], x, fun.value, nit.value, st.value


def residuals(cfg, rows, k, t2, sigma):
    rows = np.ascontiguousarray(rows, np.float32)
    k, t2, sigma = (np.ascontiguousarray(a, np.float32) for a in (k, t2, sigma))
    res = np.zeros(rows.shape[0], np.float32)
    L = lib()
    L.hostsim_residuals.argtypes = [C.POINTER(_abi.T2FitConfig), C.c_void_p, C.c_int64] + [C.c_void_p] * 4
    L.hostsim_residuals(C.byref(cfg), rows.ctypes.data, rows.shape[0], k.ctypes.data, t2.ctypes.data,
                        sigma.ctypes.data, res.ctypes.data)
    return res


def pair_roundtrip(s):
    """s: (n, 2 or 3) float64 -> what the correction-pair ring hands back for each row (a multiple of it)."""
    L = lib()
    s = np.ascontiguousarray(s, dtype=np.float64)
    out = np.empty_like(s)
    L.hostsim_pair_roundtrip.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    assert L.hostsim_pair_roundtrip(s.shape[1], s.ctypes.data, len(s), out.ctypes.data) == 0
    return out


def log_i0e4(x):
    """x: (n, 4) float64 -> (four-wide table-driven loop, one-value Cephes form), both (n, 4)."""
    L = lib()
    x = np.ascontiguousarray(x, np.float64)
    out, ref = np.empty_like(x), np.empty_like(x)
    L.hostsim_log_i0e4.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    assert L.hostsim_log_i0e4(x.ctypes.data, len(x), out.ctypes.data, ref.ctypes.data) == 0
    return out, ref


def i0e4_by_lane(x):
    """x: (n, 4) float64, each row on one side of 8 -> (the shared 30-step loop, the one-value Cephes form), (n, 4)."""
    L = lib()
    x = np.ascontiguousarray(x, np.float64)
    out, ref = np.empty_like(x), np.empty_like(x)
    L.hostsim_i0e4_by_lane.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    assert L.hostsim_i0e4_by_lane(x.ctypes.data, len(x), out.ctypes.data, ref.ctypes.data) == 0
    return out, ref


def rowsums4(terms, nte_special):
    """terms: (n, 4) float64 -> the four column sums as the Rician echo loop accumulates them."""
    L = lib()
    terms = np.ascontiguousarray(terms, np.float64)
    out = np.zeros(4)
    L.hostsim_rowsums4.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert L.hostsim_rowsums4(terms.ctypes.data, len(terms), int(nte_special), out.ctypes.data) == 0
    return out


def rician_eval(cfg, row, x, nte_special):
    """(f, g0, g1, g2) of the lane solver's evaluation at x, and the one-at-a-time reference form's f(x)."""
    L = lib()
    row = np.ascontiguousarray(row, np.float32)
    x = np.ascontiguousarray(x, np.float64)
    out, ref = np.zeros(4), np.zeros(4)
    L.hostsim_rician_eval.argtypes = [C.POINTER(_abi.T2FitConfig), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rc = L.hostsim_rician_eval(C.byref(cfg), row.ctypes.data, x.ctypes.data, int(nte_special), out.ctypes.data, ref.ctypes.data)
    assert rc == 0, rc
    return out, ref[0]


def log_lean(x):
    """The lane's log() for i0e values (fdlibm's algorithm) on positive float64 numbers."""
    L = lib()
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    L.hostsim_log_lean.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    assert L.hostsim_log_lean(x.ctypes.data, x.size, out.ctypes.data) == 0
    return out


def sqrt_near(x, a):
    """x: (n,) base arguments > 0, a: (n, 3) arguments near them -> (sqrt(x), sqrt(a), sqrt(a) from h alone) as the
    FMA sequences of the kernels compute them from a float-precision reciprocal-root seed."""
    L = lib()
    x = np.ascontiguousarray(x, np.float64)
    a = np.ascontiguousarray(a, np.float64)
    base, near, from_h = np.empty_like(x), np.empty_like(a), np.empty_like(a)
    L.hostsim_sqrt_near.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.hostsim_sqrt_near(x.ctypes.data, a.ctypes.data, len(x), base.ctypes.data, near.ctypes.data, from_h.ctypes.data) == 0
    return base, near, from_h


def i0e4_by_lane_near(x):
    """x: (n, 4) float64 > 0, each row's arguments close together on one side of 8 -> (shared reciprocal root, independent)."""
    L = lib()
    x = np.ascontiguousarray(x, np.float64)
    out, ref = np.empty_like(x), np.empty_like(x)
    L.hostsim_i0e4_by_lane_near.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    assert L.hostsim_i0e4_by_lane_near(x.ctypes.data, len(x), out.ctypes.data, ref.ctypes.data) == 0
    return out, ref
