"""Wall time of the host-buffer entry point (numpy in, numpy out: t2fit_volume_host through t2map.fit_volume) on a
256^3 x 8 TE volume -- what a caller of the reference-style API sees, PCIe and staging included.  One JSON line.

    python tools/host_entry_bench.py [solver] [precision] [fit]
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
from fetal_t2mapping_amd import synth  # noqa: E402

solver = sys.argv[1] if len(sys.argv) > 1 else "lbfgsb"
precision = sys.argv[2] if len(sys.argv) > 2 else "f64"
fit = sys.argv[3] if len(sys.argv) > 3 else "gaussian_rician"
shape = (256, 256, 256)
e, m, te = synth.brain_volume_torch(shape, 8, synth.SEED_BASE + 5, torch.device("cuda", 0))
e_h = e.reshape((8,) + shape).cpu().numpy()
m_h = m.reshape(shape).cpu().numpy()
del e, m
table = t2.fit_table(fit, True)
ts, ts_reuse = [], []
for _ in range(5):  # fresh output arrays on every call (their first-touch page faults are inside the time)
    t0 = time.perf_counter()
    maps = t2.fit_volume(e_h, m_h, te, fit, table, solver=solver, precision=precision)
    ts.append(time.perf_counter() - t0)
for _ in range(6):  # steady state of a caller that fits subject after subject into the same arrays
    t0 = time.perf_counter()
    maps = t2.fit_volume(e_h, m_h, te, fit, table, solver=solver, precision=precision, out=maps)
    ts_reuse.append(time.perf_counter() - t0)
n = int(np.prod(shape))
print(json.dumps({"workload": "256^3 x 8 TE, numpy in -> fit_volume -> numpy out (t2fit_volume_host, context API)", "solver": solver,
                  "precision": precision, "fit": fit, "seconds_fresh_outputs": [round(t, 4) for t in ts],
                  "seconds_reused_outputs": [round(t, 4) for t in ts_reuse],
                  "steady_state_ms": round(1e3 * float(np.median(ts_reuse[1:])), 2),
                  "Mvoxel_s_host_inclusive": round(n / float(np.median(ts_reuse[1:])) / 1e6, 1)}))
