"""Build libt2fit_hip.so (gfx950 only) in-tree with hipcc.  `python -m fetal_t2mapping_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libt2fit_hip.so")
SOURCES = [os.path.join(CSRC, "t2fit_kernels.hip")]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def deps():
    out = list(SOURCES)
    out += [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    out.append(os.path.join(os.path.dirname(PKG), "include", "t2fit.h"))
    return out


def up_to_date() -> bool:
    return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps())


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    if not force and up_to_date():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # -ffp-contract=off: the lane solvers say where a multiply-add is fused (fma()) and where numpy's / the
    # library's two roundings are kept; results then do not depend on how the compiler happens to group code
    cmd = [hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc",
           "-ffp-contract=off", "-Wall", "-Wno-unused-function", *extra, "-o", LIB, *SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra=["-Rpass-analysis=kernel-resource-usage"] if "--resources" in sys.argv else ()))
