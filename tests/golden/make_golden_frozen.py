"""Golden fixtures of the reference under the software stack it FREEZES (requirements_frozen.txt:103,144).

    cd /root/repo && OPENBLAS_NUM_THREADS=1 /opt/conda/bin/python3.9 -B tests/golden/make_golden_frozen.py

The reference pins numpy 1.26.0 / scipy 1.11.3 (Fortran L-BFGS-B 3.0).  The build container's default interpreter
(numpy 2.2.6 / scipy 1.15.3, C translation of L-BFGS-B) is what tests/golden/voxels_*.npz were made with.  The
container's second interpreter, /opt/conda/bin/python3.9, carries numpy 1.26.4 / scipy 1.7.1 -- the same numpy
promotion rules (value-based casting, before NEP 50) and the same Fortran optimiser family as the frozen stack --
and imports the reference unmodified in the same way (empty stand-ins for the five absent third-party modules).
This script runs the reference's own ``fit_voxel`` and ``compute_residuals`` under it on the SAME inputs as the 36
voxel fixtures (``y``, ``te`` are read from voxels_<name>.npz, nothing is regenerated) and writes

    frozen_voxels_<name>.npz : x, success, nit, fun, raised, res  -- the reference's outputs under numpy 1.26
                               stable  -- rows that K_SEEDS one-ulp-perturbed repeats of the fit reproduce
                                          (same nit, same success, T2 within 1e-3 ms), see below
                               y_crc   -- zlib.crc32 of the float32 input rows (ties the file to its input fixture)

Two things differ from the numpy-2 fixtures and both are numpy's promotion rules, not the optimiser:
  * run_t2mapping.py:169  ``np.log(signal) - np.log(sigma**2)`` is float32 array - float64 scalar: float32 under
    numpy 1.26, float64 under numpy >= 2.  The forward-difference sigma-gradient (h = 1e-8) of the rician objective
    then sees a term that almost never changes: the trajectory is a different one.
  * utils/t2map_utils.py:74-80  ``k_map * np.exp(-te / t2_map)`` with te an np.float64 scalar: float32 throughout
    under numpy 1.26, float64 rounded to float32 under numpy >= 2.
The library reproduces them with ``cfg.numpy_legacy = 1`` (include/t2fit.h); the default is the numpy-2 form.

Stable sets: the fit of every row is repeated K_SEEDS times with exp / log / i0e results moved by -1, 0 or +1 ulp
(float64 functions: fresh per call, as in oracle/noise_model.py; the float32 ``np.log(signal)``: one draw per row
and seed, the same in every evaluation of that fit, as a different libm would behave).
"""
import glob
import os
import sys
import types
import zlib

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

import numpy as np  # noqa: E402
import scipy  # noqa: E402
from scipy.optimize import minimize  # noqa: E402
from scipy.special import i0e  # noqa: E402

K_SEEDS = 8
EPS = np.finfo(float).eps


def _install_stubs():
    for name in ["SimpleITK", "pydicom", "skimage", "skimage.restoration", "skimage.measure", "nibabel",
                 "statsmodels"]:
        sys.modules[name] = types.ModuleType(name)
    s = sys.modules["SimpleITK"]
    s.sitkLinear = 1
    s.Image = object
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, "/root/reference")


_install_stubs()
import run_t2mapping as R  # noqa: E402  (the reference)


def _args(mode, low_field):
    return types.SimpleNamespace(gaussian=mode == "gaussian", gaussian_rician=mode == "gaussian_rician",
                                 rician=mode == "rician", lf=low_field, hf=not low_field, norm=False)


def _bounds(fit_params, y0, prior):
    b = [tuple(map(float, t)) for t in fit_params["param_bounds"]]
    if not prior:
        b[0] = (float(y0), 10000.0)
        b[1] = (10.0, 2000.0)
    return b


def perturbed_objectives(rng, y):
    """run_t2mapping.py:141-177 with the not-correctly-rounded library results moved by at most one ulp, keeping the
    dtypes numpy 1.26 gives every intermediate (the float32 ``np.log(signal)`` is jittered in float32, once)."""
    def jitter(v):
        return v * (1 + EPS * rng.integers(-1, 2, size=np.shape(v)))

    def pexp(z):
        return jitter(np.exp(z))

    with np.errstate(all="ignore"):
        ly = np.log(y)
    assert ly.dtype == np.float32
    step = rng.integers(-1, 2, size=ly.shape)
    ly = np.where(step > 0, np.nextafter(ly, np.float32(np.inf)), np.where(step < 0, np.nextafter(ly, np.float32(-np.inf)), ly))
    assert ly.dtype == np.float32

    def gauss(p, te, y):
        k, t2 = p
        r = y - k * pexp(-te / t2)
        return np.sum(r ** 2) / len(y)

    def gauss_rician(p, te, y):
        k, t2, s = p
        r = y - (k ** 2 * pexp(-2 * te / t2) + s ** 2) ** (1 / 2)
        return np.sum(r ** 2) / len(y)

    def rician(p, te, y):
        k, t2, s = p
        m = k * pexp(-te / t2)
        x = (m * y) / (s ** 2)
        a = ly - jitter(np.log(s ** 2))  # float32 array - float64 scalar: float32 (value-based casting)
        assert a.dtype == np.float32, a.dtype
        return -np.sum(a - (y ** 2 + m ** 2) / (2 * s ** 2) + (np.abs(x) + jitter(np.log(jitter(i0e(x))))))

    return {"gaussian": gauss, "gaussian_rician": gauss_rician, "rician": rician}


def one_run(args):
    """(path, seed): seed None = the reference's own fit_voxel; otherwise one perturbed repeat of every fittable row."""
    path, seed = args
    d = np.load(path)
    mode, lf, prior = str(d["mode"]), bool(d["low_field"]), bool(d["prior"])
    y, te = d["y"], d["te"]
    m = y.shape[0]
    n_par = d["x"].shape[1]
    x = np.full((m, n_par), np.nan)
    nit = np.zeros(m, np.int32)
    ok = np.zeros(m, bool)
    fun = np.full(m, np.nan)
    raised = np.zeros(m, bool)
    devnull = open(os.devnull, "w")
    if seed is None:
        fit, fit_params = R.set_fit_params(_args(mode, lf))
        for v in range(m):
            old = sys.stdout
            sys.stdout = devnull  # the reference prints on failures
            try:
                with np.errstate(all="ignore"):
                    xv, okv, it, f, _ = R.fit_voxel(v, fit, fit_params, te, y, prior, False)
            except ValueError:
                raised[v] = True
                continue
            finally:
                sys.stdout = old
            x[v], ok[v], nit[v], fun[v] = xv, okv, it, f
        okrows = np.where(~raised)[0]
        k_map = np.zeros(m, np.float32)
        t2_map = np.zeros(m, np.float32)
        sg_map = np.zeros(m, np.float32)
        k_map[okrows] = x[okrows, 0].astype(np.float32)
        t2_map[okrows] = x[okrows, 1].astype(np.float32)
        if n_par == 3:
            sg_map[okrows] = x[okrows, 2].astype(np.float32)
        with np.errstate(all="ignore"):
            res = R.compute_residuals(y, te, fit, False, k_map, t2_map, sg_map, np.zeros(m, np.float32), okrows,
                                      np.zeros((m, 1, 1), bool))
        return x, nit, ok, fun, raised, np.asarray(res).reshape(-1).astype(np.float32)
    rng = np.random.default_rng(seed)
    for v in range(m):
        if d["raised"][v] or not np.all(np.isfinite(y[v])):
            continue
        _, fp = R.set_fit_params(_args(mode, lf))
        yv = np.array(y[v])
        obj = perturbed_objectives(rng, yv)[mode]
        with np.errstate(all="ignore"):
            r = minimize(obj, fp["initial_guess"], args=(te, yv), method="L-BFGS-B",
                         bounds=_bounds(fp, y[v, 0], prior), options=fp["options"], jac=False)
        x[v], nit[v], ok[v] = r.x, r.nit, r.success
    return x, nit, ok


def main():
    import multiprocessing as mp

    assert np.__version__.startswith("1.26"), f"run this with /opt/conda/bin/python3.9 (numpy 1.26), not numpy {np.__version__}"
    # the promotion rule this whole file is about, checked on the running interpreter
    assert (np.ones(2, np.float32) - np.float64(0.1)).dtype == np.float32
    paths = sorted(glob.glob(os.path.join(HERE, "voxels_*.npz")))
    assert len(paths) == 36
    seeds = [None] + [777 + 1000 * j for j in range(K_SEEDS)]
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        runs = pool.map(one_run, [(p, sd) for p in paths for sd in seeds], chunksize=1)
    per = len(seeds)
    for pi, path in enumerate(paths):
        d = np.load(path)
        name = os.path.basename(path)[7:-4]
        x, nit, ok, fun, raised, res = runs[pi * per]
        assert np.array_equal(raised, d["raised"]), name
        good = np.isfinite(x[:, 1]) & np.isfinite(fun) & ~raised
        stable = good.copy()
        fr = []
        for xs, ns, oks in runs[pi * per + 1:(pi + 1) * per]:
            dt = np.abs(xs[:, 1] - x[:, 1])
            stable &= np.isfinite(xs[:, 1]) & (dt <= 1e-3) & (ns == nit) & (oks == ok)
            fr.append(np.mean(dt[good] <= 1.0))
        # how far is the numpy-2 / C-translation answer (the default fixtures) from this one?
        dt2 = np.abs(d["x"][:, 1] - x[:, 1])
        both = good & np.isfinite(d["x"][:, 1])
        np.savez_compressed(
            os.path.join(HERE, f"frozen_voxels_{name}.npz"), x=x, success=ok, nit=nit, fun=fun, raised=raised, res=res,
            stable=stable, good=good, frac_1ms_seeds=np.array(fr), k_seeds=np.int64(K_SEEDS),
            y_crc=np.int64(zlib.crc32(np.ascontiguousarray(d["y"]).tobytes())),
            numpy_version=np.array(np.__version__), scipy_version=np.array(scipy.__version__))
        print(f"{name:38s} stable {int(stable.sum()):3d}/{int(good.sum())}  self-agreement within 1 ms: min {np.min(fr):.3f}  "
              f"numpy-2 fixture within 1 ms of this: {np.mean(dt2[both] <= 1.0):.3f}  nit equal {np.mean(d['nit'][both] == nit[both]):.3f}",
              flush=True)


if __name__ == "__main__":
    main()
