"""The reference's objectives with every not-correctly-rounded library function moved by at most one ulp.
TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/: tests/, smoke() and tools that check parity).

Why: the reference differentiates its objective by forward differences with an absolute step of 1e-8 (scipy's
`eps`), which amplifies rounding noise in the objective by 1e8, and it stops on loose tests (ftol = gtol = 1e-2
for the 3-parameter models; run_t2mapping.py:38-106).  Its result for a voxel therefore depends on the last bit
of every exp() / log() / i0e(): another libm, another SIMD width or another scipy build moves a fraction of the
voxels by many milliseconds (SURVEY.md F5 measured scipy 1.7.1 against 1.15.3).  Repeating the reference's fit
with those results jittered by -1, 0 or +1 ulp at random measures how well the reference agrees with ITSELF --
the best any implementation that is not bit-identical to the reference's binary stack can do -- and tells the
voxels whose answer does not depend on that last bit (the stable set) from the ones where it does.
sqrt and the arithmetic operations are correctly rounded everywhere and are not touched.
"""
import numpy as np
from scipy.special import i0e

EPS = np.finfo(float).eps


def perturbed_objectives(rng, numpy_legacy=False, y_row=None):
    """{mode: objective(p, te, y)}: run_t2mapping.py:141-177 with exp, log and i0e jittered by one ulp (rng).
    ``numpy_legacy`` (with ``y_row``, the float32 samples of the one voxel this objective will be used for): the rician
    objective as numpy < 2 evaluates it -- ``np.log(signal) - np.log(sigma**2)`` in float32 (oracle._obj_rician_legacy) --
    with the float32 ``np.log(signal)`` moved by at most one float32 ulp ONCE for the voxel (another libm gives another,
    but always the same, value) and the float64 functions jittered per call as before."""
    def jitter(v):
        return v * (1 + EPS * rng.integers(-1, 2, size=np.shape(v)))

    def pexp(z):
        return jitter(np.exp(z))

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
        return -np.sum((jitter(np.log(y)) - jitter(np.log(s ** 2))) - (y ** 2 + m ** 2) / (2 * s ** 2)
                       + (np.abs(x) + jitter(np.log(jitter(i0e(x))))))

    if numpy_legacy:
        with np.errstate(all="ignore"):
            ly = np.log(np.asarray(y_row, np.float32)).astype(np.float32)
        step = rng.integers(-1, 2, size=ly.shape)
        ly = np.where(step > 0, np.nextafter(ly, np.float32(np.inf)), np.where(step < 0, np.nextafter(ly, np.float32(-np.inf)), ly)).astype(np.float32)

        def rician_legacy(p, te, y):
            k, t2, s = p
            m = k * pexp(-te / t2)
            x = (m * y) / (s ** 2)
            a = ly - np.float32(jitter(np.log(s ** 2)))  # float32 array - float32 scalar: float32 under either numpy
            return -np.sum(a - (y ** 2 + m ** 2) / (2 * s ** 2) + (np.abs(x) + jitter(np.log(jitter(i0e(x))))))

        return {"gaussian": gauss, "gaussian_rician": gauss_rician, "rician": rician_legacy}
    return {"gaussian": gauss, "gaussian_rician": gauss_rician, "rician": rician}


def perturbed_fit_rows(args):
    """Pool worker: the reference's fit (same table, bounds, options as oracle.fit_voxel) of rows ``idx`` of ``rows``
    with the jittered objective.  ``args = (idx, fit, low_field, prior, te, rows, seed)`` -> list of (x, nit, success)."""
    from scipy.optimize import minimize

    from . import t2fit_oracle as O

    idx, fit, low_field, prior, te, rows, seed = args[:7]
    legacy = len(args) > 7 and bool(args[7])
    rng = np.random.default_rng(seed)
    fun = None if legacy else perturbed_objectives(rng)[fit]
    out = []
    for v in idx:
        fp = O.fit_table(fit, low_field)
        lb, ub = O.voxel_bounds(fp, rows[v, 0], prior)
        if legacy:
            fun = perturbed_objectives(rng, True, rows[v])[fit]
        with np.errstate(all="ignore"):
            r = minimize(fun, fp["initial_guess"], args=(te, np.array(rows[v])), method="L-BFGS-B",
                         bounds=list(zip(lb, ub)), options=fp["options"], jac=False)
        out.append((r.x, r.nit, r.success))
    return out


def reference_fit_rows(args):
    """Pool worker: the unperturbed reference-equivalent fit (oracle.fit_voxel) of rows ``idx``.
    ``args = (idx, fit, low_field, prior, te, rows[, numpy_legacy])`` -> list of (x, nit, success)."""
    from . import t2fit_oracle as O

    idx, fit, low_field, prior, te, rows = args[:6]
    legacy = len(args) > 6 and bool(args[6])
    fp = O.fit_table(fit, low_field)
    out = []
    for v in idx:
        with np.errstate(all="ignore"):
            x, ok, nit, f, _ = O.fit_voxel(int(v), fit, fp, te, rows, prior, False, want_trace=False, numpy_legacy=legacy)
        out.append((x, nit, ok))
    return out
