"""Agreement of the Rician-likelihood fit with the live oracle on 20 000 voxels of the bench distribution (8 TE, prior bounds,
numpy >= 2 form and numpy_legacy form) and on the stable sets of the twelve rician fixtures, for whatever library T2FIT_LIB
selects.  Used by tools/experiments/r03_exp12.sh to price a leaner log().

    python tools/rician_log_parity.py <label>
"""
import glob
import multiprocessing as mp
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

from fetal_t2mapping_amd import synth  # noqa: E402
from oracle.noise_model import reference_fit_rows  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "lib"
ev, mv, te = synth.brain_volume((8, 128, 128), 8, synth.SEED_BASE + 3)
rows = np.ascontiguousarray(ev.reshape(8, -1)[:, mv.reshape(-1) != 0].T)[:20000]
cores = min(16, len(os.sched_getaffinity(0)))
chunks = [c for c in np.array_split(np.arange(len(rows)), cores * 4) if len(c)]
ref = {}
with mp.get_context("fork").Pool(cores) as pool:  # before the GPU is touched
    for legacy in (False, True):
        plain = [r for part in pool.map(reference_fit_rows, [(c, "rician", True, True, te, rows, legacy) for c in chunks]) for r in part]
        ref[legacy] = (np.array([r[0] for r in plain]), np.array([r[1] for r in plain]))
import fetal_t2mapping_amd as t2  # noqa: E402

for legacy in (False, True):
    x, ok, nit, fun, st = t2.fit_voxels(np.arange(len(rows)), "rician", t2.fit_table("rician", True), te, rows, True, False,
                                        numpy_legacy=legacy)
    dt = np.abs(x[:, 1] - ref[legacy][0][:, 1])
    print(f"{label}: 20000 voxels{' numpy_legacy' if legacy else ''}: within 1 ms {np.mean(dt <= 1.0):.4f}  nit equal {np.mean(nit == ref[legacy][1]):.4f}  "
          f"p50 {np.median(dt):.2e}  p99 {np.percentile(dt, 99):.3f} ms", flush=True)
G = os.path.join(REPO, "tests", "golden")
floor = np.load(os.path.join(G, "noise_floor.npz"))
tot = off = nd = 0
for path in sorted(glob.glob(os.path.join(G, "voxels_*_rician_*.npz"))):
    name = os.path.basename(path)[7:-4]
    if "gaussian" in name:
        continue
    d = np.load(path)
    r = np.flatnonzero(floor[name + "/stable"])
    x, ok, nit, fun, st = t2.fit_voxels(r, "rician", t2.fit_table("rician", bool(d["low_field"])), d["te"], d["y"], bool(d["prior"]), False)
    tot += len(r)
    off += int(np.sum(np.abs(x[:, 1] - d["x"][r, 1]) > 1.0))
    nd += int(np.sum(nit != d["nit"][r]))
    good = ~d["raised"] & np.isfinite(d["fun"])
    xa = t2.fit_voxels(np.flatnonzero(good), "rician", t2.fit_table("rician", bool(d["low_field"])), d["te"], d["y"], bool(d["prior"]), False)[0]
print(f"{label}: stable sets of the 12 rician fixtures: {tot} rows, {off} beyond 1 ms, {nd} with another iteration count", flush=True)
