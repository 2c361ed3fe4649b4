"""ctypes mirror of include/t2fit.h (structs and constants).  Keep in lock-step with the header."""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 4
MAX_TE = 32

OK, E_INVALID, E_HIP, E_BOUNDS = 0, -1, -2, -3

MODEL_GAUSSIAN, MODEL_GAUSSIAN_RICIAN, MODEL_RICIAN = 0, 1, 2
MODELS = {"gaussian": MODEL_GAUSSIAN, "gaussian_rician": MODEL_GAUSSIAN_RICIAN, "rician": MODEL_RICIAN}

SOLVER_LBFGSB, SOLVER_LM, SOLVER_LOGLIN = 0, 1, 2
SOLVERS = {"lbfgsb": SOLVER_LBFGSB, "L-BFGS-B": SOLVER_LBFGSB, "lm": SOLVER_LM, "loglin": SOLVER_LOGLIN}

PREC_F64, PREC_F32 = 0, 1
PRECISIONS = {"f64": PREC_F64, "f32": PREC_F32}

LAYOUT_TE_MAJOR, LAYOUT_VOXEL_MAJOR = 0, 1

ST_MASKED, ST_CONVERGED, ST_NOT_CONV, ST_NONFINITE, ST_INFEASIBLE = 0, 1, 2, 3, 4


class T2FitConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("model", C.c_int32), ("solver", C.c_int32), ("precision", C.c_int32),
        ("n_te", C.c_int32), ("no_prior", C.c_int32), ("norm", C.c_int32), ("maxls", C.c_int32),
        ("maxiter", C.c_int32), ("maxfun", C.c_int32), ("numpy_legacy", C.c_int32), ("reserved1", C.c_int32),
        ("te_ms", C.c_double * MAX_TE),
        ("x0", C.c_double * 3), ("lb", C.c_double * 3), ("ub", C.c_double * 3),
        ("ftol", C.c_double), ("gtol", C.c_double), ("fd_step", C.c_double), ("lm_xtol", C.c_double),
        ("noprior_k_ub", C.c_double), ("noprior_t2_lb", C.c_double), ("noprior_t2_ub", C.c_double),
    ]


class T2FitMaps(C.Structure):
    _fields_ = [
        ("t2", C.c_void_p), ("k", C.c_void_p), ("sigma", C.c_void_p), ("res", C.c_void_p),
        ("r2", C.c_void_p), ("fun", C.c_void_p), ("nit", C.c_void_p), ("status", C.c_void_p),
        ("t2_se", C.c_void_p),
    ]


# every symbol include/t2fit.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("t2fit_config_default", C.c_int, [C.POINTER(T2FitConfig), C.c_int, C.c_int]),
    ("t2fit_device_count", C.c_int, []),
    ("t2fit_volume_dev", C.c_int, [C.POINTER(T2FitConfig), _P, C.c_int, _P, C.c_int64, C.POINTER(T2FitMaps), _P]),
    ("t2fit_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("t2fit_destroy", C.c_int, [_P]),
    ("t2fit_context_volume_host", C.c_int, [_P, C.POINTER(T2FitConfig), _P, C.c_int, _P, C.c_int64, C.POINTER(T2FitMaps)]),
    ("t2fit_volume_host", C.c_int, [C.POINTER(T2FitConfig), _P, C.c_int, _P, C.c_int64, C.POINTER(T2FitMaps), C.c_int]),
    ("t2fit_voxels_host", C.c_int, [C.POINTER(T2FitConfig), _P, C.c_int, C.c_int64, _P, C.c_int64, _P, _P, _P, _P, C.c_int]),
    ("t2fit_voxels_trace_host", C.c_int, [C.POINTER(T2FitConfig), _P, C.c_int, C.c_int64, _P, C.c_int64, _P, _P, _P, _P,
                                          C.c_int, _P, _P, C.c_int]),
    ("t2fit_union_mask_dev", C.c_int, [_P, C.c_int, C.c_int64, _P, _P, _P, _P]),
    ("t2fit_residuals_dev", C.c_int, [C.POINTER(T2FitConfig), _P, C.c_int, _P, C.c_int64, _P, _P, _P, _P, _P]),
    ("t2fit_label_stats_dev", C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P, _P, _P]),
    ("t2fit_set_timing", C.c_int, [C.c_int]),
    ("t2fit_set_reserve_cus", C.c_int, [C.c_int]),
    ("t2fit_kernel_ms", C.c_double, [C.c_int]),
    ("t2fit_last_kernel_ms", C.c_double, []),
    ("t2fit_epilogue_ms", C.c_double, [C.c_int]),
    ("t2fit_last_error", C.c_char_p, []),
    ("t2fit_abi_version", C.c_int, []),
]


def bind(lib: C.CDLL) -> C.CDLL:
    """Attach prototypes; raises AttributeError if the library lacks a declared symbol."""
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
