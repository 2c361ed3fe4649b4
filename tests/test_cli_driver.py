"""Host logic of the CLI / volume driver (run_t2mapping.py mirror) on CPU: flags, metadata filter,
input and output file names, geometry copy, map scatter.  The GPU fit is replaced by the oracle here
(checker standing in for the kernel); tests/test_gpu_parity.py runs the same driver on the device."""
import os

import numpy as np
import pandas as pd
import pytest

import fake_sitk
from conftest import GOLDEN
from oracle import t2fit_oracle as O


def _subject(tmp_path, d):
    from fetal_t2mapping_amd import cli as R

    bids = str(tmp_path / "projects") + "/"
    os.makedirs(os.path.join(bids, "prj-900"))
    rows = []
    for i, t in enumerate(d["te"]):
        acq = {"prj": "prj-900", "sub": "sub-001", "ses": "ses-01", "run": f"run-{i + 1:02d}", "EchoTime": t / 1000.0,
               "CoilString": "HeadNeck"}
        rows.append(acq)
        np.save(R.get_img_path(bids, acq, R.recon_dirname).replace(" ", "") + ".npy", d["echoes"][i])
        np.save(R.get_img_path(bids, acq, R.mask_dirname).replace(" ", "") + ".npy", d["masks"][i])
    return bids, pd.DataFrame(rows)


def _oracle_fit_subject(vols, masks, keep, te_eff, fit, fit_params, prior, norm, solver, precision, device):
    data, mask, idx = O.stack_mask_flatten(vols, masks)
    if keep is not None:
        mask = mask & keep
        idx = np.flatnonzero(mask.reshape(-1))
    r = O.fit_volume(data, idx, te_eff, fit, O.fit_table(fit, True), prior=prior, norm=norm)
    st = np.zeros(data.shape[0], np.uint8)
    st[idx] = np.where(r.success, 1, 2)
    sh = mask.shape
    return mask, (r.t2.reshape(sh), r.k.reshape(sh), r.sigma.reshape(sh), r.res.reshape(sh)), st


def test_process_t2maps_plumbing_matches_reference_run(tmp_path, monkeypatch):
    from fetal_t2mapping_amd import cli as R

    sitk = fake_sitk.install()
    d = np.load(os.path.join(GOLDEN, "volume_lf_gaussian_noprior.npz"))
    bids, md = _subject(tmp_path, d)
    monkeypatch.setattr(R, "_fit_subject", _oracle_fit_subject)
    args = R.parse_arguments(["--path", str(tmp_path), "--csv", "x.csv", "--in_vivo", "--gaussian", "--lf", "--sim", "g1",
                              "--no_prior"])
    fit, fit_params = R.t2map.set_fit_params(args)
    R.process_t2maps(md, bids, [int(t) for t in d["te"]], fit, fit_params, False, True, False, False, False, "g1")
    names = sorted(os.path.relpath(p, bids) for p in sitk.written)
    assert names == [str(s) for s in d["written"]]
    for path, img in sitk.written.items():
        key = path.split("_sim-g1_")[1].split("map_")[0]
        assert np.array_equal(img.arr, d[key]), key  # oracle == reference, so the plumbing must be exact
        assert img.GetSpacing() == tuple(d["spacing"]) and img.GetOrigin() == tuple(d["origin"])


def test_cli_flags_and_tables():
    from fetal_t2mapping_amd import cli as R

    with pytest.raises(SystemExit):
        R.parse_arguments(["--path", "p", "--csv", "c.csv", "--in_vivo", "--gaussian", "--rician", "--lf", "--sim", "1"])
    with pytest.raises(SystemExit):
        R.parse_arguments(["--path", "p", "--csv", "c.csv", "--gaussian", "--lf", "--sim", "1"])
    a = R.parse_arguments(["--path", "p", "--csv", "a.csv", "b.csv", "--in_vitro_fast", "--rician", "--hf", "--sim", "s"])
    fit, fp = R.t2map.set_fit_params(a)
    assert fit == "rician" and fp == O.fit_table("rician", False)
    for mode in O.MODES:
        for lf in (True, False):
            assert R.t2map.fit_table(mode, lf) == O.fit_table(mode, lf)
    a.norm = True
    with pytest.raises(SystemExit):  # run_t2mapping.py:107-109
        R.t2map.set_fit_params(a)
    assert R.set_phantom_gt(True)[0][0] == 594 and len(R.set_phantom_gt(False)[0]) == 14


def test_missing_te_is_skipped(tmp_path, monkeypatch, capsys):
    from fetal_t2mapping_amd import cli as R

    sitk = fake_sitk.install()
    d = np.load(os.path.join(GOLDEN, "volume_lf_gaussian_noprior.npz"))
    bids, md = _subject(tmp_path, d)
    monkeypatch.setattr(R, "_fit_subject", _oracle_fit_subject)
    R.process_t2maps(md, bids, [114, 202, 250], "gaussian", O.fit_table("gaussian", True), False, True, True, False,
                     False, "s")
    assert "T2 fit is skipped" in capsys.readouterr().out and not sitk.written


def test_convergence_figures_from_traces(tmp_path, monkeypatch):
    """convergence.py (run_t2mapping.py:465-468): sampling, row gathering and the three file names, with
    the trace entry point replaced by the oracle's own traces (the GPU test draws them from the device)."""
    pytest.importorskip("matplotlib")
    from fetal_t2mapping_amd import convergence, t2map

    d = np.load(os.path.join(GOLDEN, "volume_lf_gaussian_noprior.npz"))
    te, vols = d["te"], list(d["echoes"])
    mask = d["masks"].sum(axis=0) > 0
    idx = np.flatnonzero(mask.reshape(-1))
    seen = {}

    def fake_trace(indices, fit, fit_params, TEeffs, rows, prior, norm, **kw):
        seen["rows"] = rows
        infos = []
        for r in indices:
            infos.append(O.fit_voxel(int(r), fit, O.fit_table(fit, True), TEeffs, rows, prior, norm)[4])
        return None, None, None, None, None, infos

    monkeypatch.setattr(t2map, "fit_voxels_trace", fake_trace)
    ada = convergence.set_ada_path(str(tmp_path), "prj-900")
    n = mask.size
    nit, fun = np.zeros(n, np.int32), np.zeros(n, np.float32)
    nit[idx], fun[idx] = 5, 1.0
    out = convergence.convergence_study(ada, vols, idx, d["t2"], nit, fun, te, "gaussian", O.fit_table("gaussian", True),
                                        False, False, "sub-001", "ses-01", "g1", seed=7)
    assert [os.path.basename(p) for p in out] == [
        "convergence_20_random_voxels_colored_by_t2_sub-001_ses-01_sim-g1_gaussian.png",
        "step_size_convergence_20_random_voxels_colored_by_t2_sub-001_ses-01_sim-g1.png",
        "scatter_iterations_vs_loss_colored_by_t2_sub-001_ses-01_sim-g1.png"]
    assert all(os.path.getsize(p) > 1000 for p in out)
    m = min(len(idx), 50) + min(len(idx), 20)
    assert seen["rows"].shape == (m, len(te)) and seen["rows"].dtype == np.float32
    # the gathered rows are rows of the (N, nTE) stack at sampled mask indices
    stack = np.stack([v.reshape(-1) for v in vols], axis=1).astype(np.float32)
    assert all(any(np.array_equal(r, stack[i]) for i in idx) for r in seen["rows"][:5])


def test_phantom_csv_text_equals_the_references_file():
    """tests/golden/phantom_lf_gaussian_rician_fast.npz holds a whole `process_t2maps(phantom=True, fast=True)` run of
    the reference (make_golden.py::phantom_fixture): its maps and the text of the ROI CSV `save_phantom_csv` wrote
    (utils/t2map_utils.py:30-59).  From the reference's own maps -- per-vial statistics as numpy computes them on float32
    maps -- cli.phantom_frame must write that text character for character: column names and order, the swapped
    `id` / `trueT2` columns (run_t2mapping.py:27 vs :478), float32-valued numbers in pandas' digits."""
    import io
    import os

    from conftest import GOLDEN
    from fetal_t2mapping_amd import cli as R

    d = np.load(os.path.join(GOLDEN, "phantom_lf_gaussian_rician_fast.npz"))
    label = d["label"]
    id_, gt_ = R.set_phantom_gt(True)  # the reference's swapped unpacking: `id` holds the T2 values, `gt` the vial names
    stats = {}
    for arr, m, s in ((d["t2"], "meanT2", "stdT2"), (d["k"], "meanK", "stdK"), (d["sigma"], "meanC", "stdC")):
        stats[m] = [np.nanmean(arr[label == i + 1]) for i in range(len(gt_))]
        stats[s] = [np.nanstd(arr[label == i + 1]) for i in range(len(gt_))]
    buf = io.StringIO()
    R.phantom_frame(stats, id_, gt_).to_csv(buf, index=False)
    assert buf.getvalue() == str(d["csv_text"])
    assert str(d["csv_name"]).endswith("sub-001_ses-01_recon_1mm_sim-p1_ROI_data_ada-gaussian_rician.csv")
    # float64-accumulated statistics (what the GPU reduction delivers) land on the same text or one float32 step beside it
    stats64 = {k: [float(np.float64(v)) for v in vals] for k, vals in stats.items()}
    for arr, m, s in ((d["t2"], "meanT2", "stdT2"),):
        stats64[m] = [np.nanmean(arr[label == i + 1].astype(np.float64)) for i in range(len(gt_))]
        stats64[s] = [np.nanstd(arr[label == i + 1].astype(np.float64)) for i in range(len(gt_))]
    got = R.phantom_frame(stats64, id_, gt_)
    want = R.phantom_frame(stats, id_, gt_)
    assert np.allclose(got["meanT2"], want["meanT2"], rtol=3e-7, atol=0) and np.allclose(got["stdT2"], want["stdT2"], rtol=2e-6, atol=0)


def test_project_shortcuts_of_the_csv_flag(tmp_path, capsys):
    """`--csv prj-004` (prj-003, prj-002) stands for that project's session logs (utils/metadata_utils.py:19-85,96-113):
    13 low-field / 10 high-field logs for prj-004, one each for the phantom projects, prj-003 at 1.5 T exits like the
    reference; the logs are concatenated in list order, and anything that is neither a project nor a .csv name exits."""
    import pandas as pd

    from fetal_t2mapping_amd import cli as R

    assert len(R.project_csvs("prj-004", True)) == 13 and len(R.project_csvs("prj-004", False)) == 10
    assert R.project_csvs("prj-004", True)[0] == "2024083017_17510000.csv" and R.project_csvs("prj-004", False)[-1] == "2024102122_28450000.csv"
    assert R.project_csvs("prj-003", True) == ["20240806_30540000_1.csv"]
    assert R.project_csvs("prj-002", True) == ["20240527_095111_2.csv"] and R.project_csvs("prj-002", False) == ["20240609_50140000_2.csv"]
    with pytest.raises(SystemExit):
        R.project_csvs("prj-003", False)
    for i, name in enumerate(R.project_csvs("prj-004", False)):
        pd.DataFrame([{"prj": "prj-004", "sub": f"sub-{i:03d}", "EchoTime": 0.114}]).to_csv(tmp_path / name, index=False)
    md = R.set_metadata(str(tmp_path), ["prj-004"], False)
    assert list(md["sub"]) == [f"sub-{i:03d}" for i in range(10)]
    assert "PRJ-004 - In vivo adult brain data acquired using the head coil" in capsys.readouterr().out
    with pytest.raises(SystemExit):
        R.set_metadata(str(tmp_path), ["prj-005"], True)
