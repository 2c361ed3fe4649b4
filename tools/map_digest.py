#!/usr/bin/env python3
"""SHA-256 of the maps the library writes for the synthetic bench volume, per configuration.

    python tools/map_digest.py [Z Y X] > digest_new.txt
    T2FIT_LIB=tools/diag/libt2fit_r01.so python tools/map_digest.py > digest_old.txt ; diff digest_old.txt digest_new.txt

Kernel restructurings that only change scheduling (which lane fits which voxel when, instruction order of
independent work) must leave every map bit for bit as it was: run this with the library before and after.
"""
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import fetal_t2mapping_amd as t2  # noqa: E402
from fetal_t2mapping_amd import synth  # noqa: E402


def main():
    shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 256, 256)
    dev = torch.device("cuda", 0)
    for n_te in (8, 6, 3, 9):
        echoes, mask, te = synth.brain_volume_torch(shape, n_te, synth.SEED_BASE + 3, dev)
        for fit, prior, solver, prec in (("gaussian_rician", True, "lbfgsb", "f64"), ("gaussian_rician", False, "lbfgsb", "f64"),
                                         ("gaussian", True, "lbfgsb", "f64"), ("gaussian", False, "lbfgsb", "f64"),
                                         ("rician", True, "lbfgsb", "f64"), ("gaussian_rician", True, "lm", "f32"),
                                         ("gaussian_rician", True, "lm", "f64"), ("gaussian", True, "loglin", "f64")):
            if n_te != 8 and (solver != "lbfgsb" or fit == "rician" or not prior):
                continue
            m = t2.fit_volume(echoes.reshape((n_te,) + shape), mask, te, fit, t2.fit_table(fit, True), prior=prior,
                              solver=solver, precision=prec, extras=(n_te == 8 and fit == "gaussian_rician" and prior))
            torch.cuda.synchronize()
            h = hashlib.sha256()
            for name in ("t2", "k", "sigma", "res", "nit", "status"):
                a = getattr(m, name)
                if a is not None:
                    h.update(a.cpu().numpy().tobytes())
            print(f"{shape[0]}x{shape[1]}x{shape[2]}x{n_te} {fit} {'prior' if prior else 'noprior'} {solver} {prec} {h.hexdigest()[:32]}",
                  flush=True)


if __name__ == "__main__":
    main()
