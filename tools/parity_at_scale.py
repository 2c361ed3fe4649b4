"""Parity of the HIP path with the reference-equivalent oracle on a larger sample than the fixtures:
N masked voxels of the bench distribution, per configuration: HIP (through the C ABI) vs the oracle
(scipy loop, this host's cores) and, as the yardstick, the oracle vs itself with exp() perturbed by one
ulp (tests/golden/make_noise_floor.py explains why that is the best any implementation can do).

    python tools/parity_at_scale.py [N] [--all] > profiles/rNN_parity_at_scale.json      (on the GPU box)

--all adds the 2-parameter no-prior and the Rician-likelihood configurations and the closed-form solver against its oracle.
"""
import json
import multiprocessing as mp
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))

import numpy as np  # noqa: E402
from scipy.optimize import minimize  # noqa: E402

from fetal_t2mapping_amd import synth  # noqa: E402
from oracle import t2fit_oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
ALL = "--all" in sys.argv
EPS = np.finfo(float).eps


def _perturbed(args):
    idx, fit, prior, te, rows, seed = args
    rng = np.random.default_rng(seed)

    def pexp(z):
        e = np.exp(z)
        return e * (1 + EPS * rng.integers(-1, 2, size=np.shape(e)))

    def gauss(p, t, y):
        r = y - p[0] * pexp(-t / p[1])
        return np.sum(r ** 2) / len(y)

    def gr(p, t, y):
        r = y - (p[0] ** 2 * pexp(-2 * t / p[1]) + p[2] ** 2) ** (1 / 2)
        return np.sum(r ** 2) / len(y)

    def ric(p, t, y):  # run_t2mapping.py:157-177 with the perturbed exp()
        from scipy.special import i0e

        m = p[0] * pexp(-t / p[1])
        xx = (m * y) / (p[2] ** 2)
        return -np.sum((np.log(y) - np.log(p[2] ** 2)) - (y ** 2 + m ** 2) / (2 * p[2] ** 2) + (np.abs(xx) + np.log(i0e(xx))))

    fun = {"gaussian": gauss, "gaussian_rician": gr, "rician": ric}[fit]
    out = []
    for v in idx:
        fp = O.fit_table(fit, True)
        lb, ub = O.voxel_bounds(fp, rows[v, 0], prior)
        r = minimize(fun, fp["initial_guess"], args=(te, np.array(rows[v])), method="L-BFGS-B", bounds=list(zip(lb, ub)),
                     options=fp["options"], jac=False)
        out.append(r.x)
    return out


def main():
    ev, mv, te = synth.brain_volume((8, 128, 128), 8, synth.SEED_BASE + 3)
    rows = np.ascontiguousarray(ev.reshape(8, -1)[:, mv.reshape(-1) != 0].T)[:N]
    cores = min(16, len(os.sched_getaffinity(0)))
    report = {"n_voxels": int(len(rows)), "te_ms": te.tolist(), "cores": cores, "configs": {}}
    with mp.get_context("fork").Pool(cores) as pool:  # before the GPU is touched
        ref = {}
        configs = [("gaussian", True), ("gaussian_rician", True), ("gaussian_rician", False)]
        if ALL:
            configs += [("gaussian", False), ("rician", True)]
        for fit, prior in configs:
            r = O.fit_volume(rows, np.arange(len(rows)), te, fit, O.fit_table(fit, True), prior=prior, pool=pool)
            chunks = np.array_split(np.arange(len(rows)), cores * 4)
            pert = np.array([x for part in pool.map(_perturbed, [(c, fit, prior, te, rows, 7 + i) for i, c in enumerate(chunks)])
                             for x in part])
            ref[(fit, prior)] = (r, pert)
    import fetal_t2mapping_amd as t2

    for (fit, prior), (r, pert) in ref.items():
        x, ok, nit, fun, st = t2.fit_voxels(np.arange(len(rows)), fit, t2.fit_table(fit, True), te, rows, prior, False)
        dt = np.abs(x[:, 1] - r.t2.astype(np.float64))  # oracle maps are float32 casts
        dtp = np.abs(pert[:, 1] - r.t2.astype(np.float64))
        if fit != "rician":
            xl, okl, _, funl, _ = t2.fit_voxels(np.arange(len(rows)), fit, t2.fit_table(fit, True), te, rows, prior, False,
                                                solver="lm", precision="f32")
        else:  # the LM lane serves the least-squares models only
            xl, funl = np.full_like(x, np.nan), np.full(len(rows), np.nan)
        dl = np.abs(xl[:, 1] - r.t2.astype(np.float64))
        report["configs"][f"{fit}/{'prior' if prior else 'noprior'}"] = {
            "hip_lbfgsb_vs_reference": {"within_1ms": float(np.mean(dt <= 1.0)), "median_ms": float(np.median(dt)),
                                        "p90_ms": float(np.percentile(dt, 90)), "p99_ms": float(np.percentile(dt, 99)),
                                        "success_equal": float(np.mean(ok == r.success)),
                                        "nit_equal": float(np.mean(nit == r.nit))},
            "reference_vs_itself_one_ulp_exp": {"within_1ms": float(np.mean(dtp <= 1.0)), "median_ms": float(np.median(dtp)),
                                                "p90_ms": float(np.percentile(dtp, 90)),
                                                "p99_ms": float(np.percentile(dtp, 99))},
            "hip_lm_f32_vs_reference": {"within_1ms": float(np.mean(dl <= 1.0)), "median_ms": float(np.median(dl)),
                                        "objective_not_worse": float(np.mean(funl <= r.fun * (1 + 2e-3) + 1e-9))},
        }
    if ALL:  # closed-form solver against its own oracle (extension: no reference counterpart)
        for prior in (True, False):
            want, okw = O.loglinear_fit(rows, te, O.fit_table("gaussian", True), prior=prior)
            x, ok, _, _, _ = t2.fit_voxels(np.arange(len(rows)), "gaussian", t2.fit_table("gaussian", True), te, rows, prior,
                                           False, solver="loglin")
            rel = np.abs(x[:, 1] - want[:, 1]) / want[:, 1]
            report["configs"][f"loglin/{'prior' if prior else 'noprior'}"] = {
                "hip_vs_closed_form_oracle": {"max_rel_t2": float(rel.max()), "median_rel_t2": float(np.median(rel)),
                                              "max_rel_k": float((np.abs(x[:, 0] - want[:, 0]) / want[:, 0]).max()),
                                              "status_equal": float(np.mean(ok == okw))}}
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
