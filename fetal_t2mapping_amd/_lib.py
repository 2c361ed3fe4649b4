"""Loader of the HIP library.  There is no CPU execution path: if libt2fit_hip.so is missing or
no MI355X is visible, every compute entry point raises."""
from __future__ import annotations

import ctypes as C
import os

from . import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
# T2FIT_LIB selects another build of the same library (diagnostic builds under tools/diag/)
LIB_PATH = os.environ.get("T2FIT_LIB") or os.path.join(_PKG, "lib", "libt2fit_hip.so")

_lib = None


class T2FitError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"t2fit error {code}: {message}")
        self.code = code


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every symbol of include/t2fit.h."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m fetal_t2mapping_amd.build` "
                "(hipcc, gfx950).  fetal_t2mapping_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 and this package
        # uses torch for device buffers, so torch is imported first and libt2fit_hip.so then binds
        # to the runtime torch already loaded (same SONAME).  Loading ours first makes torch's
        # later initialisation fail with "No HIP GPUs are available".
        import torch  # noqa: F401

        lib = _abi.bind(C.CDLL(LIB_PATH))
        if lib.t2fit_abi_version() != _abi.ABI_VERSION:
            raise RuntimeError("libt2fit_hip.so ABI version does not match fetal_t2mapping_amd/_abi.py")
        _lib = lib
    return _lib


def require_gpu() -> C.CDLL:
    lib = load()
    if lib.t2fit_device_count() < 1:
        raise RuntimeError("no HIP device visible: the T2 fit runs on an MI355X only (no CPU fallback)")
    return lib


def check(rc: int) -> None:
    if rc != _abi.OK:
        msg = load().t2fit_last_error().decode("utf-8", "replace")
        if rc == _abi.E_BOUNDS:
            # scipy's wording at the reference call site (run_t2mapping.py:261 -> _minimize_lbfgsb)
            raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound. (" + msg + ")")
        if rc == _abi.E_INVALID:
            raise ValueError(msg)
        raise T2FitError(rc, msg)
