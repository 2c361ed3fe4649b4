"""Randomised cross-check of the large-volume kernels (one-wave workgroups, samples in registers) against the generic
small-volume kernel: random sizes just above 2^20 voxels, ragged ends, mask fills from 0.5 % to 100 %, 3 to 8 echoes,
both Gaussian objectives, prior / no prior, both layouts.  Every map must be bit-identical.  python tools/soak_kernel_variants.py [n_cases]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fetal_t2mapping_amd import synth  # noqa: E402
from fetal_t2mapping_amd import t2map as t2  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2026)
dev = torch.device("cuda", 0)
bad = 0
for case in range(n_cases):
    n_te = int(rng.choice([3, 4, 5, 6, 7, 8]))
    fit = str(rng.choice(["gaussian", "gaussian_rician"]))
    prior = bool(rng.integers(0, 2))
    layout = str(rng.choice(["te_major", "voxel_major"]))
    n = int((1 << 20) + rng.integers(1, 3_000_000))
    fill = float(rng.choice([0.005, 0.05, 0.3, 0.7, 1.0]))
    z = (n + 65535) // 65536
    e, _, te = synth.brain_volume_torch((z, 256, 256), n_te, synth.SEED_BASE + 100 + case, dev)
    e = e[:, :n].contiguous()
    g = torch.Generator(device="cpu").manual_seed(case)
    m = (torch.rand(n, generator=g) < fill).to(torch.uint8).to(dev) if fill < 1.0 else None
    table = t2.fit_table(fit, True)

    def run(ev, mv, cnt):
        vol = ev.reshape(n_te, 1, 1, cnt) if layout == "te_major" else ev.t().contiguous().reshape(1, 1, cnt, n_te)
        return t2.fit_volume(vol, mv, te, fit, table, prior=prior, layout=layout, extras=True, strict=False)

    whole = run(e, m, n)
    ok = True
    piece = 1 << 19
    for lo in range(0, n, piece):
        hi = min(n, lo + piece)
        part = run(e[:, lo:hi].contiguous(), None if m is None else m[lo:hi].contiguous(), hi - lo)
        for name in ("t2", "k", "sigma", "res", "nit", "status"):
            a, b = getattr(whole, name).reshape(-1)[lo:hi], getattr(part, name).reshape(-1)
            if not bool(((a == b) | (a.isnan() & b.isnan())).all()):
                ok = False
    bad += not ok
    print(f"case {case:2d}: n={n} n_te={n_te} {fit} prior={prior} {layout} fill={fill}: {'ok' if ok else 'MISMATCH'}", flush=True)
    del whole, e, m
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
