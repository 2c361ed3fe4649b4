"""Parity of the HIP path with the reference-equivalent oracle on a larger sample than the fixtures:
N masked voxels of the bench distribution, per configuration: HIP (through the C ABI) vs the oracle
(scipy loop, this host's cores) and, as the yardstick, the oracle vs itself with exp / log / i0e perturbed by one
ulp (oracle/noise_model.py; tests/golden/make_noise_floor.py explains why that is the best any implementation can do).

    python tools/parity_at_scale.py [N] [--all] > profiles/rNN_parity_at_scale.json      (on the GPU box)

--all adds the 2-parameter no-prior and the Rician-likelihood configurations (the latter also as numpy 1.26 evaluates it,
cfg.numpy_legacy) and the closed-form solver against its oracle.  --hf takes the high-field tables (run_t2mapping.py:68-106)
instead of the low-field ones, --n-te K a train of K echoes instead of eight.
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

from fetal_t2mapping_amd import synth  # noqa: E402
from oracle import t2fit_oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
ALL = "--all" in sys.argv
LOW_FIELD = "--hf" not in sys.argv
N_TE = int(sys.argv[sys.argv.index("--n-te") + 1]) if "--n-te" in sys.argv else 8
EPS = np.finfo(float).eps


def main():
    ev, mv, te = synth.brain_volume((8, 128, 128), N_TE, synth.SEED_BASE + 3)
    rows = np.ascontiguousarray(ev.reshape(N_TE, -1)[:, mv.reshape(-1) != 0].T)[:N]
    cores = min(16, len(os.sched_getaffinity(0)))
    report = {"n_voxels": int(len(rows)), "te_ms": te.tolist(), "cores": cores, "tables": "low field" if LOW_FIELD else "high field",
              "configs": {}}
    from oracle.noise_model import perturbed_fit_rows, reference_fit_rows

    with mp.get_context("fork").Pool(cores) as pool:  # before the GPU is touched
        ref = {}
        # (fit, prior, numpy_legacy): the last configuration is the rician objective as the numpy 1.26 the reference freezes
        # evaluates it (float32 log term), HIP with cfg.numpy_legacy = 1 against the oracle's explicit-cast restatement
        configs = [("gaussian", True, False), ("gaussian_rician", True, False), ("gaussian_rician", False, False)]
        if ALL:
            configs += [("gaussian", False, False), ("rician", True, False), ("rician", True, True)]
        chunks = [c for c in np.array_split(np.arange(len(rows)), cores * 4) if len(c)]
        for fit, prior, legacy in configs:
            plain = [r for part in pool.map(reference_fit_rows, [(c, fit, LOW_FIELD, prior, te, rows, legacy) for c in chunks]) for r in part]
            pert = [r for part in pool.map(perturbed_fit_rows, [(c, fit, LOW_FIELD, prior, te, rows, 7 + i, legacy)
                                                                for i, c in enumerate(chunks)]) for r in part]
            ref[(fit, prior, legacy)] = (np.array([r[0] for r in plain]), np.array([r[1] for r in plain]),
                                         np.array([r[2] for r in plain]), np.array([r[0] for r in pert]))
    import fetal_t2mapping_amd as t2

    for (fit, prior, legacy), (x_ref, nit_ref, ok_ref, pert) in ref.items():
        x, ok, nit, fun, st = t2.fit_voxels(np.arange(len(rows)), fit, t2.fit_table(fit, LOW_FIELD), te, rows, prior, False,
                                            numpy_legacy=legacy)
        dt = np.abs(x[:, 1] - x_ref[:, 1])
        dtp = np.abs(pert[:, 1] - x_ref[:, 1])
        entry = {
            "hip_lbfgsb_vs_reference": {"within_1ms": float(np.mean(dt <= 1.0)), "median_ms": float(np.median(dt)),
                                        "p90_ms": float(np.percentile(dt, 90)), "p99_ms": float(np.percentile(dt, 99)),
                                        "success_equal": float(np.mean(ok == ok_ref)),
                                        "nit_equal": float(np.mean(nit == nit_ref))},
            "reference_vs_itself_one_ulp": {"within_1ms": float(np.mean(dtp <= 1.0)), "median_ms": float(np.median(dtp)),
                                            "p90_ms": float(np.percentile(dtp, 90)),
                                            "p99_ms": float(np.percentile(dtp, 99))}}
        if fit != "rician":  # the LM lane serves the least-squares models only
            xl, okl, _, funl, _ = t2.fit_voxels(np.arange(len(rows)), fit, t2.fit_table(fit, LOW_FIELD), te, rows, prior, False,
                                                solver="lm", precision="f32")
            dl = np.abs(xl[:, 1] - x_ref[:, 1])
            entry["hip_lm_f32_vs_reference"] = {"within_1ms": float(np.mean(dl <= 1.0)), "median_ms": float(np.median(dl))}
        report["configs"][f"{fit}/{'prior' if prior else 'noprior'}{'/numpy_legacy' if legacy else ''}"] = entry
    if ALL:  # closed-form solver against its own oracle (extension: no reference counterpart)
        for prior in (True, False):
            want, okw = O.loglinear_fit(rows, te, O.fit_table("gaussian", LOW_FIELD), prior=prior)
            x, ok, _, _, _ = t2.fit_voxels(np.arange(len(rows)), "gaussian", t2.fit_table("gaussian", LOW_FIELD), te, rows, prior,
                                           False, solver="loglin")
            rel = np.abs(x[:, 1] - want[:, 1]) / want[:, 1]
            report["configs"][f"loglin/{'prior' if prior else 'noprior'}"] = {
                "hip_vs_closed_form_oracle": {"max_rel_t2": float(rel.max()), "median_rel_t2": float(np.median(rel)),
                                              "max_rel_k": float((np.abs(x[:, 0] - want[:, 0]) / want[:, 0]).max()),
                                              "status_equal": float(np.mean(ok == okw))}}
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
