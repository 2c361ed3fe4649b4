"""Generate the golden fixtures by running the REFERENCE's own code in the build container.

    cd /root/repo && OPENBLAS_NUM_THREADS=1 python -B tests/golden/make_golden.py

The reference (/root/reference, read-only) is imported unmodified; the five third-party modules
it imports but this image lacks (SimpleITK, pydicom, skimage, nibabel, statsmodels) are replaced
by empty stand-in modules so that ``import run_t2mapping`` succeeds -- none of them is on the
per-voxel path.  For the volume fixture a small npy-backed stand-in for the four SimpleITK calls
the driver makes lets the reference's whole ``process_t2maps`` run (Pool, plots, file naming).

Outputs (committed, data only -- inputs and the reference's outputs):
  voxels_<field>_<mode>_<prior>_te<n>.npz   per-voxel fits: x, success, nit, fun, traces, residuals
  volume_lf_gaussian_noprior.npz             whole process_t2maps run on a 6x12x14x3 volume
  phantom_lf_gaussian_rician_fast.npz        whole process_t2maps run with phantom=True, fast=True (--in_vitro_fast) on an
                                             8x40x40x3 phantom: maps, file names and the ROI CSV text (--phantom-only
                                             regenerates just this one)
The reference never travels to the GPU box; these files and oracle/ do.
"""
import os
import sys
import tempfile
import types

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import scipy  # noqa: E402


# ---- stand-ins for absent third-party modules (not on the fit path) ---------------------------
class _FakeImage:
    def __init__(self, arr, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
                 direction=(1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0)):
        self.arr, self.spacing, self.origin, self.direction = arr, spacing, origin, direction

    def GetSpacing(self): return self.spacing
    def GetOrigin(self): return self.origin
    def GetDirection(self): return self.direction
    def SetSpacing(self, s): self.spacing = tuple(s)
    def SetOrigin(self, o): self.origin = tuple(o)
    def SetDirection(self, d): self.direction = tuple(d)


def _install_stubs():
    for name in ["SimpleITK", "pydicom", "skimage", "skimage.restoration", "skimage.measure", "nibabel",
                 "statsmodels"]:
        sys.modules[name] = types.ModuleType(name)
    s = sys.modules["SimpleITK"]
    s.sitkLinear = 1
    s.Image = _FakeImage
    s.written = {}
    s.ReadImage = lambda path: _FakeImage(np.load(path + ".npy"), spacing=(1.0, 1.0, 1.5),
                                          origin=(-3.0, 4.0, 5.0))
    s.GetArrayFromImage = lambda img: img.arr
    s.GetImageFromArray = lambda arr: _FakeImage(np.asarray(arr))

    def _write(img, path):
        s.written[path] = img
    s.WriteImage = _write
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, "/root/reference")


_install_stubs()
import run_t2mapping as R  # noqa: E402  (the reference)

from fetal_t2mapping_amd import synth  # noqa: E402
from oracle import t2fit_oracle as O  # noqa: E402

N_RANDOM = 240
N_TRACE = 8
TRACE_LEN = 64


def _args(mode, low_field):
    return types.SimpleNamespace(gaussian=mode == "gaussian", gaussian_rician=mode == "gaussian_rician",
                                 rician=mode == "rician", lf=low_field, hf=not low_field, norm=False)


def voxel_fixture(mode, low_field, prior, n_te, cfg_index):
    te = synth.te_vector(n_te, low_field, integer=True)
    rng = np.random.default_rng(synth.SEED_BASE + 1000 + cfg_index)
    y_rand, truth = synth.voxels(rng, te, N_RANDOM)
    y_edge, edge_names = synth.edge_rows(te, low_field)
    y = np.concatenate([y_edge, y_rand]).astype(np.float32)
    m = y.shape[0]
    fit, fit_params = R.set_fit_params(_args(mode, low_field))
    n_par = len(fit_params["initial_guess"])
    x = np.full((m, n_par), np.nan)
    success = np.zeros(m, bool)
    nit = np.zeros(m, np.int32)
    fun = np.full(m, np.nan)
    raised = np.zeros(m, bool)
    x_tight = np.full((m, n_par), np.nan)
    f_tight = np.full(m, np.nan)
    tr_f = np.full((N_TRACE, TRACE_LEN), np.nan)
    tr_s = np.full((N_TRACE, TRACE_LEN), np.nan)
    first_random = len(edge_names)
    devnull = open(os.devnull, "w")
    for v in range(m):
        old = sys.stdout
        sys.stdout = devnull  # the reference prints on failures
        try:
            with np.errstate(all="ignore"):
                xv, ok, it, f, info = R.fit_voxel(v, fit, fit_params, te, y, prior, False)
        except ValueError:
            raised[v] = True
            continue
        finally:
            sys.stdout = old
        x[v], success[v], nit[v], fun[v] = xv, ok, it, f
        t = v - first_random
        if 0 <= t < N_TRACE:
            for j, e in enumerate(info[:TRACE_LEN]):
                tr_f[t, j], tr_s[t, j] = e["f_val"], e["step_size"]
        lb, ub = O.voxel_bounds(R.set_fit_params(_args(mode, low_field))[1], y[v, 0], prior)
        starts = [xv, fit_params["initial_guess"], [truth[max(t, 0), 0], truth[max(t, 0), 1], 20.0][:n_par],
                  [y[v, 0] * 2.0, 60.0, 5.0][:n_par], [y[v, 0] * 1.2, 400.0, 60.0][:n_par]]
        with np.errstate(all="ignore"):
            x_tight[v], f_tight[v] = O.tight_solve(mode, te, y[v], lb, ub, starts)
    # reference residual map on the fitted rows (utils/t2map_utils.py:62-89)
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
    res = np.asarray(res).reshape(-1)
    name = f"voxels_{'lf' if low_field else 'hf'}_{mode}_{'prior' if prior else 'noprior'}_te{n_te}.npz"
    np.savez_compressed(
        os.path.join(HERE, name), y=y, te=te, x=x, success=success, nit=nit, fun=fun, raised=raised,
        x_tight=x_tight, f_tight=f_tight, res=res.astype(np.float32), trace_f=tr_f, trace_step=tr_s,
        trace_first_row=np.int64(first_random), truth=truth, edge_names=np.array(edge_names),
        x0=np.array(fit_params["initial_guess"], np.float64),
        table_bounds=np.array(R.set_fit_params(_args(mode, low_field))[1]["param_bounds"], np.float64),
        mode=np.array(mode), low_field=np.array(low_field), prior=np.array(prior),
        numpy_version=np.array(np.__version__), scipy_version=np.array(scipy.__version__))
    return name, int(raised.sum()), float(np.nanmean(nit))


def volume_fixture():
    """Run the reference's entire process_t2maps on a small synthetic subject."""
    import pandas as pd

    shape, n_te = (6, 12, 14), 3
    echoes, mask, te = synth.brain_volume(shape, n_te, synth.SEED_BASE + 1, low_field=True, fill=0.35)
    # per-TE masks differ slightly so the union (run_t2mapping.py:383-384) is exercised
    masks = [mask.copy() for _ in range(n_te)]
    masks[1][0, 0, 0] = 1
    masks[2][-1, -1, -1] = 1
    masks[0][3, 6, 7] = 0
    tmp = tempfile.mkdtemp(prefix="t2golden_")
    bids = os.path.join(tmp, "projects") + "/"
    prj, sub, ses = "prj-900", "sub-001", "ses-01"
    os.makedirs(os.path.join(bids, prj, "ada"))
    rows = []
    for i, t in enumerate(te):
        acq = {"prj": prj, "sub": sub, "ses": ses, "run": f"run-{i + 1:02d}", "EchoTime": t / 1000.0,
               "CoilString": "HeadNeck"}
        rows.append(acq)
        for dirname, arr in ((R.recon_dirname, echoes[i]), (R.mask_dirname, masks[i])):
            p = R.get_img_path(bids, acq, dirname).replace(" ", "")
            np.save(p + ".npy", arr)
    metadata = pd.DataFrame(rows)
    fit, fit_params = R.set_fit_params(_args("gaussian", True))
    import random
    random.seed(0)
    old = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        R.process_t2maps(metadata, bids, [int(t) for t in te], fit, fit_params, False, True, False, False,
                         False, "g1")
    finally:
        sys.stdout = old
    written = sys.modules["SimpleITK"].written
    out = {}
    names = []
    for path, img in written.items():
        rel = os.path.relpath(path, bids)
        names.append(rel)
        key = rel.split("_sim-g1_")[1].split("map_")[0]
        out[key] = np.asarray(img.arr)
        geom = (img.GetSpacing(), img.GetOrigin(), img.GetDirection())
    np.savez_compressed(
        os.path.join(HERE, "volume_lf_gaussian_noprior.npz"), echoes=echoes, masks=np.stack(masks),
        te=te, t2=out["t2"], k=out["k"], sigma=out["sigma"], res=out["res"],
        written=np.array(sorted(names)), spacing=np.array(geom[0]), origin=np.array(geom[1]),
        direction=np.array(geom[2]),
        numpy_version=np.array(np.__version__), scipy_version=np.array(scipy.__version__))
    return sorted(names)


def phantom_fixture():
    """The reference's whole ``process_t2maps`` with phantom=True, fast=True (--in_vitro_fast) on a small NIST-phantom-like
    volume: what it writes, including the ROI CSV of ``save_phantom_csv`` (utils/t2map_utils.py:30-59) as text."""
    import pandas as pd

    echoes, mask, label, te, gt = synth.phantom_volume((8, 40, 40), 3, synth.SEED_BASE + 2, low_field=True)
    tmp = tempfile.mkdtemp(prefix="t2golden_ph_")
    bids = os.path.join(tmp, "projects") + "/"
    prj, sub, ses = "prj-901", "sub-001", "ses-01"
    os.makedirs(os.path.join(bids, prj, "ada"))
    rows = []
    for i, t in enumerate(te):
        acq = {"prj": prj, "sub": sub, "ses": ses, "run": f"run-{i + 1:02d}", "EchoTime": t / 1000.0, "CoilString": "HeadNeck"}
        rows.append(acq)
        for dirname, arr in ((R.recon_dirname, echoes[i]), (R.mask_dirname, mask), (R.phantom_labels_dirname, label)):
            np.save(R.get_img_path(bids, acq, dirname).replace(" ", "") + ".npy", arr)
    fit, fit_params = R.set_fit_params(_args("gaussian_rician", True))
    import random
    random.seed(0)
    sys.modules["SimpleITK"].written.clear()
    old = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        R.process_t2maps(pd.DataFrame(rows), bids, [int(t) for t in te], fit, fit_params, True, True, True, True, False, "p1")
    finally:
        sys.stdout = old
    written = sys.modules["SimpleITK"].written
    out, names = {}, []
    for path, img in written.items():
        rel = os.path.relpath(path, bids)
        names.append(rel)
        out[rel.split("_sim-p1_")[1].split("map_")[0]] = np.asarray(img.arr)
    csvs = [os.path.join(d, f) for d, _, fs in os.walk(bids) for f in fs if f.endswith(".csv")]
    assert len(csvs) == 1, csvs
    np.savez_compressed(
        os.path.join(HERE, "phantom_lf_gaussian_rician_fast.npz"), echoes=echoes, mask=mask, label=label, te=te,
        t2=out["t2"], k=out["k"], sigma=out["sigma"], res=out["res"], written=np.array(sorted(names)),
        csv_name=np.array(os.path.relpath(csvs[0], bids)), csv_text=np.array(open(csvs[0]).read()),
        numpy_version=np.array(np.__version__), scipy_version=np.array(scipy.__version__))
    return os.path.relpath(csvs[0], bids), open(csvs[0]).read()


def main():
    if "--phantom-only" in sys.argv:
        print(*phantom_fixture(), sep="\n")
        return
    cfg = 0
    for low_field in (True, False):
        for mode in O.MODES:
            for prior in (True, False):
                for n_te in (3, 6, 8):
                    name, n_raised, mean_nit = voxel_fixture(mode, low_field, prior, n_te, cfg)
                    print(f"{name}: raised={n_raised} mean_nit={mean_nit:.1f}", flush=True)
                    cfg += 1
    for n in volume_fixture():
        print("volume wrote", n)
    print(*phantom_fixture(), sep="\n")


if __name__ == "__main__":
    main()
