"""BASELINE.json config 5 on one GPU: S subjects of 256^3 x 8 TE streamed host->HBM->host with
fetal_t2mapping_amd.stream.fit_subjects (double-buffered), PCIe included.  Prints one JSON line.

    python tools/stream_bench.py [S] [solver] [precision] [fit] [depth]
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import fetal_t2mapping_amd as t2  # noqa: E402
from fetal_t2mapping_amd import stream, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
solver = sys.argv[2] if len(sys.argv) > 2 else "lbfgsb"
precision = sys.argv[3] if len(sys.argv) > 3 else "f64"
fit = sys.argv[4] if len(sys.argv) > 4 else "gaussian_rician"
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 2
shape = (256, 256, 256)
dev = torch.device("cuda", 0)
e, m, te = synth.brain_volume_torch(shape, 8, synth.SEED_BASE + 5, dev)
e_h = e.reshape((8,) + shape).cpu().numpy()   # one host copy, streamed S times (contents do not matter)
m_h = m.reshape(shape).cpu().numpy()
del e, m
table = t2.fit_table(fit, True)
# raw PCIe rates of this box (pinned memory, one direction at a time)
_p = torch.empty(1 << 27, dtype=torch.float32).pin_memory()
_d = torch.empty(1 << 27, dtype=torch.float32, device=dev)
pcie = {}
for name, dst, src in (("h2d", _d, _p), ("d2h", _p, _d)):
    dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    pcie[name + "_GBs"] = round(3 * _p.numel() * 4 / (time.perf_counter() - t0) / 1e9, 1)
del _p, _d
list(stream.fit_subjects([(e_h, m_h)], te, fit, table, solver=solver, precision=precision, depth=depth))  # warm-up
t0 = time.perf_counter()
n = 0
for maps in stream.fit_subjects(((e_h, m_h) for _ in range(S)), te, fit, table, solver=solver,
                                precision=precision, depth=depth):
    n += 1
dt = time.perf_counter() - t0
# the same with the echoes already in pinned memory and the maps left in the pinned output buffer
e_p = torch.from_numpy(e_h).pin_memory()
t0 = time.perf_counter()
for maps in stream.fit_subjects(((e_p, m_h) for _ in range(S)), te, fit, table, solver=solver,
                                precision=precision, copy_out=False, depth=depth):
    n += 1
dt_pinned = time.perf_counter() - t0
vox = S * shape[0] * shape[1] * shape[2]
print(json.dumps({"workload": f"{S} subjects x 256^3 x 8 TE streamed through one GPU (host buffers in and out)",
                  "solver": solver, "precision": precision, "fit": fit, "depth": depth, "seconds": round(dt, 3),
                  "ms_per_subject": round(dt / S * 1e3, 2), "Mvoxel_s_pcie_inclusive": round(vox / dt / 1e6, 1),
                  "pinned_io": {"ms_per_subject": round(dt_pinned / S * 1e3, 2),
                                "Mvoxel_s_pcie_inclusive": round(vox / dt_pinned / 1e6, 1)},
                  "pcie_pinned": pcie, "bytes_per_subject": {"h2d": int(e_h.nbytes + m_h.nbytes), "d2h": int(4 * 4 * m_h.size)}}))
