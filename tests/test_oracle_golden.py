"""Pin the CPU oracle (oracle/t2fit_oracle.py) to the reference's own outputs.

The fixtures under tests/golden/ were produced by importing the reference's ``fit_voxel``,
``compute_residuals`` and ``process_t2maps`` (tests/golden/make_golden.py).  Because the oracle
drives the same scipy entry point with the same arguments, agreement is required BIT-FOR-BIT when
the interpreter's numpy/scipy match the versions recorded in the fixture.
"""
import copy
import os

import numpy as np
import pytest
import scipy

from conftest import golden_voxel_files
from oracle import t2fit_oracle as O

FILES = golden_voxel_files()


def _same_versions(d):
    return str(d["numpy_version"]) == np.__version__ and str(d["scipy_version"]) == scipy.__version__


def test_fixture_inventory():
    # 2 fields x 3 modes x 2 prior settings x 3 echo counts
    assert len(FILES) == 36
    assert os.path.exists(os.path.join(os.path.dirname(FILES[0]), "volume_lf_gaussian_noprior.npz"))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[7:-4] for f in FILES])
def test_fit_voxel_matches_reference(path):
    d = np.load(path)
    mode, low_field, prior = str(d["mode"]), bool(d["low_field"]), bool(d["prior"])
    table = O.fit_table(mode, low_field)
    assert np.array_equal(np.array(table["initial_guess"], float), d["x0"])
    assert np.array_equal(np.array(table["param_bounds"], float), d["table_bounds"])
    y, te = d["y"], d["te"]
    # every edge row + a spread of random rows keeps the CPU suite short
    rows = list(range(int(d["trace_first_row"]) + 8)) + list(range(40, y.shape[0], 9))
    fp = copy.deepcopy(table)
    exact = _same_versions(d)
    for v in rows:
        with np.errstate(all="ignore"):
            try:
                x, ok, nit, f, info = O.fit_voxel(v, mode, fp, te, y, prior, False)
            except ValueError:
                assert d["raised"][v], f"row {v}: oracle raised, reference did not"
                continue
        assert not d["raised"][v]
        if exact:
            assert np.array_equal(x, d["x"][v], equal_nan=True), (v, x, d["x"][v])
            assert nit == d["nit"][v] and ok == d["success"][v]
            assert (f == d["fun"][v]) or (np.isnan(f) and np.isnan(d["fun"][v]))
        else:  # other scipy build: same algorithm, different binary -> statistical agreement only
            assert ok == d["success"][v]
        t = v - int(d["trace_first_row"])
        if exact and 0 <= t < d["trace_f"].shape[0]:
            n = min(len(info), d["trace_f"].shape[1])
            assert np.array_equal([e["f_val"] for e in info[:n]], d["trace_f"][t, :n])
            assert np.array_equal([e["step_size"] for e in info[:n]], d["trace_step"][t, :n], equal_nan=True)


@pytest.mark.parametrize("path", FILES[::5], ids=[os.path.basename(f)[7:-4] for f in FILES[::5]])
def test_compute_residuals_matches_reference(path):
    d = np.load(path)
    mode = str(d["mode"])
    y, te, x = d["y"], d["te"], d["x"]
    m = y.shape[0]
    rows = np.where(~d["raised"])[0]
    k = np.zeros(m, np.float32)
    t2 = np.zeros(m, np.float32)
    sg = np.zeros(m, np.float32)
    k[rows] = x[rows, 0].astype(np.float32)
    t2[rows] = x[rows, 1].astype(np.float32)
    if x.shape[1] == 3:
        sg[rows] = x[rows, 2].astype(np.float32)
    res = O.compute_residuals(y, te, mode, False, k, t2, sg, np.zeros(m, np.float32), rows)
    if str(d["numpy_version"]) == np.__version__:
        assert np.array_equal(res, d["res"], equal_nan=True)
    else:
        assert np.allclose(res, d["res"], rtol=0, atol=1e-3, equal_nan=True)


def test_volume_path_matches_reference(golden_dir):
    """Whole process_t2maps run of the reference vs the oracle's stack/flatten + fit + scatter."""
    d = np.load(os.path.join(golden_dir, "volume_lf_gaussian_noprior.npz"))
    echoes, masks, te = d["echoes"], d["masks"], d["te"]
    data, mask, idx = O.stack_mask_flatten(list(echoes), list(masks))
    assert data.dtype == np.float32 and data.shape == (echoes[0].size, len(te))
    assert idx.dtype == np.int64 and np.all(np.diff(idx) > 0)
    assert mask.sum() == len(idx) == int((masks.sum(axis=0) > 0).sum())
    fit = O.fit_volume(data, idx, te, "gaussian", O.fit_table("gaussian", True), prior=False)
    shape = echoes.shape[1:]
    same = str(d["numpy_version"]) == np.__version__ and str(d["scipy_version"]) == scipy.__version__
    for name, got in (("t2", fit.t2), ("k", fit.k), ("sigma", fit.sigma), ("res", fit.res)):
        want = d[name]
        got = got.reshape(shape)
        if same:
            assert np.array_equal(got, want), name
        else:
            assert np.allclose(got, want, rtol=1e-4, atol=1e-2), name
    # zeros outside the mask (run_t2mapping.py:415-418)
    assert np.all(fit.t2.reshape(shape)[~mask] == 0)
    names = [str(s) for s in d["written"]]
    assert names == sorted(
        f"prj-900/derivatives/recon_1mm_t2map/sub-001/ses-01/anat/sub-001_ses-01_recon_1mm_sim-g1_{p}map_ada-gaussian.nii.gz"
        for p in ("t2", "k", "sigma", "res"))


def test_tight_solution_is_no_worse_than_reference():
    """The second golden set (converged bounded minimiser) must never lose to the reference."""
    for path in FILES:
        d = np.load(path)
        ok = np.isfinite(d["fun"]) & np.isfinite(d["f_tight"])
        assert ok.sum() > 200
        assert np.all(d["f_tight"][ok] <= d["fun"][ok] * (1 + 1e-9) + 1e-12), path


# The only numeric known-answer the reference itself commits: the printed output of
# notebooks/20240910_ada_jmri.ipynb (cell 26): white-matter ROI MEAN signal at nine echo times and the scipy result
# of the 2-parameter fit, `x = [369.3, 117.6], fun = 0.1308, nit = 13, nfev = 78`.  The notebook fitted the ROI
# MEDIANS, which it does not print, so the anchor is approximate in x; the length of the
# trajectory is reproduced exactly from the printed means.
NOTEBOOK_TE = np.array([114, 132, 150, 176, 202, 229, 255, 273, 299], np.float64)
NOTEBOOK_MEAN = np.array([141.99, 121.87, 104.77, 83.86, 67.57, 54.28, 44.00, 38.35, 31.66], np.float32)
NOTEBOOK_PARAMS = {"initial_guess": [630, 165], "param_bounds": [(float(NOTEBOOK_MEAN[0]), 1e4), (10, 600)],
                   "solver": "L-BFGS-B", "options": {"ftol": 1e-6, "maxls": 50, "disp": False}}


def test_notebook_known_answer():
    import copy

    x, ok, nit, fun, _ = O.fit_voxel(0, "gaussian", copy.deepcopy(NOTEBOOK_PARAMS), NOTEBOOK_TE, NOTEBOOK_MEAN[None, :],
                                     True, False)
    assert ok and nit == 13                      # the notebook prints nit: 13 (nfev: 78)
    assert abs(x[1] - 117.6) < 0.03 * 117.6 and abs(x[0] - 369.3) < 0.03 * 369.3   # means vs the unprinted medians
    assert abs(x[0] - 363.9) < 0.1 and abs(x[1] - 120.6) < 0.1                      # SURVEY.md's measurement here


# ---- the reference under the stack it freezes (tests/golden/make_golden_frozen.py) -----------------------------
def test_frozen_fixture_inventory():
    """36 frozen_voxels_*.npz, generated under numpy 1.26.x / scipy 1.7.1 (Fortran L-BFGS-B) from the very rows of the
    default fixtures (checksum), with a non-trivial stable set each."""
    import zlib

    for path in FILES:
        name = os.path.basename(path)[7:-4]
        fz = np.load(os.path.join(os.path.dirname(path), f"frozen_voxels_{name}.npz"))
        d = np.load(path)
        assert str(fz["numpy_version"]).startswith("1.26") and str(fz["scipy_version"]) == "1.7.1"
        assert int(fz["y_crc"]) == zlib.crc32(np.ascontiguousarray(d["y"]).tobytes())
        assert np.array_equal(fz["raised"], d["raised"]) and fz["x"].shape == d["x"].shape
        assert int(fz["stable"].sum()) >= 30, name


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[7:-4] for f in FILES])
def test_oracle_against_the_frozen_stack(path):
    """The oracle (this interpreter: numpy 2 / scipy 1.15, with `numpy_legacy=True` restating the old promotion rule
    by explicit casts) against the reference run under numpy 1.26 / Fortran L-BFGS-B, on the rows stable under both
    stacks' one-ulp perturbations (rician: the frozen stack's): T2 within 1 ms on all but one row, `success` equal.
    The residual map in its float32 form is the frozen stack's BIT FOR BIT (same float32 exp in both numpys)."""
    d = np.load(path)
    name = os.path.basename(path)[7:-4]
    fz = np.load(os.path.join(os.path.dirname(path), f"frozen_voxels_{name}.npz"))
    floor = np.load(os.path.join(os.path.dirname(path), "noise_floor.npz"))
    mode, low_field, prior = str(d["mode"]), bool(d["low_field"]), bool(d["prior"])
    rows = np.flatnonzero(fz["stable"] & (floor[name + "/stable"] if mode != "rician" else True))[::3]
    fp = O.fit_table(mode, low_field)
    off = 0
    for v in rows:
        with np.errstate(all="ignore"):
            x, ok, nit, f, _ = O.fit_voxel(int(v), mode, fp, d["te"], d["y"], prior, False, want_trace=False,
                                           numpy_legacy=True)
        assert ok == fz["success"][v]
        off += abs(x[1] - fz["x"][v, 1]) > 1.0
    assert off <= 1, (name, off, len(rows))
    m = d["y"].shape[0]
    okrows = np.where(~fz["raised"])[0]
    k, t2, sg = (np.zeros(m, np.float32) for _ in range(3))
    k[okrows], t2[okrows] = fz["x"][okrows, 0], fz["x"][okrows, 1]
    if fz["x"].shape[1] == 3:
        sg[okrows] = fz["x"][okrows, 2]
    res = O.compute_residuals(d["y"], d["te"], mode, False, k, t2, sg, np.zeros(m, np.float32), okrows, numpy_legacy=True)
    fin = np.isfinite(fz["res"])
    assert np.array_equal(res[fin], fz["res"][fin])
