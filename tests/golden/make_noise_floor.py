"""How well does the reference agree with ITSELF?  (fixture: tests/golden/noise_floor.npz)

The reference differentiates its objective by forward differences with an absolute step of 1e-8
(scipy `eps`), which amplifies rounding noise in the objective by 1e8, and it stops on very loose
tests (ftol = gtol = 1e-2 for the 3-parameter models).  Its result for a voxel therefore depends on
the last bit of every exp(): another libm, another SIMD width or another scipy build moves a
fraction of the voxels by many milliseconds (SURVEY.md F5 measured scipy 1.7.1 vs 1.15.3).

This script measures that floor in the build container: the reference's fit (through the oracle,
which tests/test_oracle_golden.py pins bit-for-bit to the reference) is repeated with every exp()
result multiplied by (1 + u*2^-52), u drawn from {-1, 0, +1} -- a one-ulp perturbation -- and
compared with the unperturbed golden answers.  No implementation that is not bit-identical to the
reference's own binary stack can be expected to agree with the golden vectors better than this,
so tests/test_gpu_parity.py requires the HIP lane solver to reach the same agreement.

    cd /root/repo && OPENBLAS_NUM_THREADS=1 python -B tests/golden/make_noise_floor.py
"""
import glob
import os
import sys

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402
import scipy  # noqa: E402
from scipy.optimize import minimize  # noqa: E402
from scipy.special import i0e  # noqa: E402

from oracle import t2fit_oracle as O  # noqa: E402

EPS = np.finfo(float).eps


def perturbed_objectives(rng):
    def pexp(z):
        e = np.exp(z)
        return e * (1 + EPS * rng.integers(-1, 2, size=np.shape(e)))

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
        return -np.sum((np.log(y) - np.log(s ** 2)) - (y ** 2 + m ** 2) / (2 * s ** 2) + (np.abs(x) + np.log(i0e(x))))

    return {"gaussian": gauss, "gaussian_rician": gauss_rician, "rician": rician}


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(HERE, "voxels_*.npz"))):
        d = np.load(path)
        name = os.path.basename(path)[7:-4]
        mode, lf, prior = str(d["mode"]), bool(d["low_field"]), bool(d["prior"])
        fun = perturbed_objectives(np.random.default_rng(12345))[mode]
        m = d["y"].shape[0]
        x = np.full((m, d["x"].shape[1]), np.nan)
        nit = np.zeros(m, np.int32)
        ok = np.zeros(m, bool)
        for v in range(m):
            if d["raised"][v] or not np.all(np.isfinite(d["y"][v])):
                continue
            fp = O.fit_table(mode, lf)
            lb, ub = O.voxel_bounds(fp, d["y"][v, 0], prior)
            with np.errstate(all="ignore"):
                r = minimize(fun, fp["initial_guess"], args=(d["te"], np.array(d["y"][v])), method="L-BFGS-B",
                             bounds=list(zip(lb, ub)), options=fp["options"], jac=False)
            x[v], nit[v], ok[v] = r.x, r.nit, r.success
        good = np.isfinite(x[:, 1]) & np.isfinite(d["x"][:, 1])
        dt = np.abs(x[good, 1] - d["x"][good, 1])
        out[name + "/x"] = x
        out[name + "/nit"] = nit
        out[name + "/frac_1ms"] = np.float64(np.mean(dt <= 1.0))
        out[name + "/median_dt2"] = np.float64(np.median(dt))
        out[name + "/nit_equal"] = np.float64(np.mean(nit[good] == d["nit"][good]))
        print(f"{name:38s} within 1 ms {out[name + '/frac_1ms']:.3f}  median |dT2| {out[name + '/median_dt2']:.2g} ms  "
              f"nit equal {out[name + '/nit_equal']:.3f}", flush=True)
    out["numpy_version"] = np.array(np.__version__)
    out["scipy_version"] = np.array(scipy.__version__)
    np.savez_compressed(os.path.join(HERE, "noise_floor.npz"), **out)


if __name__ == "__main__":
    main()
