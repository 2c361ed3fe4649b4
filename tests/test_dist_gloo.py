"""N > 1 path on CPU: slab partition + all-gather of the packed maps with gloo, world_size 2 and 3.

The fit itself needs the GPU, so each rank fills its packed slab with a deterministic per-voxel
function (standing in for the kernel, whose per-voxel result does not depend on the partition);
what is tested is that partition + padding + one all_gather_into_tensor reproduce the whole volume
bit-for-bit on every rank, including ragged sizes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fetal_t2mapping_amd import dist as t2dist


def _fake_fit(echoes_slab, mask_slab):
    """Per-voxel stand-in for the kernel: depends only on that voxel's samples and mask."""
    e = echoes_slab.astype(np.float64)
    t2 = np.where(mask_slab != 0, e.sum(axis=0) * 0.5, 0.0)
    k = np.where(mask_slab != 0, e[0] * 2.0 + 1.0, 0.0)
    sg = np.where(mask_slab != 0, e[-1] - e[0], 0.0)
    res = np.where(mask_slab != 0, e.mean(axis=0), 0.0)
    return np.stack([t2, k, sg, res]).astype(np.float32)


def _worker(rank, world, port, n_vox, n_te, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    echoes = rng.normal(size=(n_te, n_vox)).astype(np.float32)
    mask = (rng.random(n_vox) < 0.6).astype(np.uint8)
    e, m = t2dist.take_slab(echoes, mask, rank, world)
    assert e.shape == (n_te, t2dist.slab_len(n_vox, world))
    packed = torch.from_numpy(_fake_fit(e, m))
    full = t2dist.gather_maps(packed, n_vox).numpy()
    whole = _fake_fit(echoes, mask)
    ok = np.array_equal(full, whole)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _worker_cyclic(rank, world, port, n_vox, n_te, chunk, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(6)
    echoes = rng.normal(size=(n_te, n_vox)).astype(np.float32)
    mask = (rng.random(n_vox) < 0.6).astype(np.uint8)
    e, m = t2dist.take_cyclic(echoes, mask, rank, world, chunk)
    per = t2dist.cyclic_len(n_vox, world, chunk)
    assert e.shape == (n_te, per) and m.shape == (per,)
    idx = t2dist.cyclic_index(n_vox, rank, world, chunk)  # the documented slot -> voxel map
    real = idx >= 0
    assert np.array_equal(e[:, real], echoes[:, idx[real]]) and np.array_equal(m[real], mask[idx[real]]) and not m[~real].any()
    packed = torch.from_numpy(_fake_fit(e, m))
    full = t2dist.gather_maps_cyclic(packed, n_vox, chunk).numpy()
    out[rank] = bool(np.array_equal(full, _fake_fit(echoes, mask)))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_vox", [(2, 1000), (2, 1001), (3, 10), (2, 1)])
def test_slab_partition_and_allgather(world, n_vox):
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_vox, 4, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)) and len(out) == world


@pytest.mark.parametrize("world,n_vox,chunk", [(2, 1000, 64), (2, 1001, 64), (3, 10, 4), (2, 1, 16), (3, 5000, 256)])
def test_cyclic_partition_and_allgather(world, n_vox, chunk):
    """Chunks dealt round-robin (the balanced partition of one volume): share of each rank, padding, one
    all_gather_into_tensor and the strided view back into voxel order reproduce the whole volume bit for bit."""
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker_cyclic, args=(world, _free_port(), n_vox, 4, chunk, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)) and len(out) == world


def test_cyclic_partition_covers_every_voxel_once_and_balances_an_ellipsoid():
    for n, g, chunk in ((0, 2, 8), (1, 8, 8), (1000, 3, 16), (256 * 256 * 32, 8, t2dist.CHUNK)):
        idx = np.concatenate([t2dist.cyclic_index(n, r, g, chunk) for r in range(g)])
        assert np.array_equal(np.sort(idx[idx >= 0]), np.arange(n))
        assert len(idx) == g * t2dist.cyclic_len(n, g, chunk)
    # the reason the partition exists: masked voxels per rank on an ellipsoidal mask, 8 ranks
    from fetal_t2mapping_amd import synth

    _, mask, _ = synth.brain_volume((64, 64, 64), 2, seed=1)
    flat = mask.reshape(-1)
    per_slab = [int(flat[slice(*t2dist.slab_range(flat.size, r, 8))].sum()) for r in range(8)]
    per_cyc = []
    for r in range(8):
        idx = t2dist.cyclic_index(flat.size, r, 8, 1024)
        per_cyc.append(int(flat[idx[idx >= 0]].sum()))
    assert max(per_slab) / np.mean(per_slab) > 1.4      # contiguous slabs: the middle ranks do 1.5x the average work
    assert max(per_cyc) / np.mean(per_cyc) < 1.08       # cyclic chunks: within a few percent


def test_slab_ranges_cover_the_volume():
    for n in (0, 1, 7, 8, 9, 360 * 512 * 512):
        for g in (1, 2, 4, 8):
            per = t2dist.slab_len(n, g)
            spans = [t2dist.slab_range(n, r, g) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= per for lo, hi in spans)
    # cfg4: 360 slices over 8 GPUs are whole Z-slabs of 45 slices
    assert t2dist.slab_range(360 * 512 * 512, 3, 8) == (3 * 45 * 512 * 512, 4 * 45 * 512 * 512)


def test_subject_round_robin_covers_every_subject_once(monkeypatch):
    """Config 5 (32 subjects over 8 GPUs): subject s goes to rank s % world; every subject is fitted exactly
    once, in order within a rank, and the loader is called lazily.  The streaming fit is replaced by a stand-in."""
    from fetal_t2mapping_amd import stream

    for n, world in ((32, 8), (5, 4), (3, 8), (0, 2)):
        seen = sorted(s for r in range(world) for s in t2dist.subjects_of_rank(n, r, world))
        assert seen == list(range(n))
    assert list(t2dist.subjects_of_rank(32, 3, 8)) == [3, 11, 19, 27]
    with pytest.raises(ValueError):
        t2dist.subjects_of_rank(4, 2, 2)
    loaded = []

    def fake_stream(subjects, TEeffs, fit, fit_params, prior, norm, **kw):
        for e, m in subjects:
            yield float(e.sum())

    monkeypatch.setattr(stream, "fit_subjects", fake_stream)

    def load(s):
        loaded.append(s)
        return np.full((2, 1, 1, 1), s, np.float32), None

    got = list(t2dist.fit_subjects_round_robin(load, 10, [1.0, 2.0], "gaussian", {}, rank=1, world=4))
    assert got == [(1, 2.0), (5, 10.0), (9, 18.0)] and loaded == [1, 5, 9]


# ---- the CLI's --gpus control flow (fetal_t2mapping_amd/cli.py) with gloo, the GPU fit replaced by a stand-in -------
def _fake_status(e, m):
    """Per-voxel stand-in for the status byte: 2 (not converged) where the first sample is below 95, else 1; 0 outside."""
    return np.where(m != 0, np.where(e[0] < 95.0, 2, 1), 0).astype(np.uint8)


def _cli_fake_share(e_share, m_share, te_eff, fit, fit_params, prior, norm, solver, precision, device):
    """Stand-in for cli._fit_share: same contract -- (packed [6, per]: t2, k, sigma, res, fun, nit bits; status [per])."""
    e = e_share.numpy() if torch.is_tensor(e_share) else e_share
    m = m_share.numpy() if torch.is_tensor(m_share) else m_share
    four = _fake_fit(e, m)
    fun = np.where(m != 0, e[1] * 0.25, 0.0).astype(np.float32)
    nit = np.where(m != 0, (np.abs(e[0]) % 17).astype(np.int32) + 1, 0).astype(np.int32)
    packed = np.concatenate([four, fun[None], nit.view(np.float32)[None]])
    return torch.from_numpy(packed), torch.from_numpy(_fake_status(e, m))


def _cli_fake_subject(vols, masks, keep, te_eff, fit, fit_params, prior, norm, solver, precision, device):
    e = np.stack([np.asarray(v, np.float32).reshape(-1) for v in vols])
    mask = np.zeros(np.asarray(masks[0]).shape, bool)
    for m in masks:
        mask |= np.asarray(m) != 0
    mflat = mask.reshape(-1).astype(np.uint8)
    out = _fake_fit(e, mflat)
    return mask, tuple(o.reshape(mask.shape) for o in out), _fake_status(e, mflat).reshape(mask.shape)


def _cli_worker(rank, world, port, root, n_subjects, out, extra_args):
    import contextlib
    import io
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_sitk

    from fetal_t2mapping_amd import cli as R
    from fetal_t2mapping_amd import convergence

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), T2FIT_CLI_BACKEND="gloo")
    sitk = fake_sitk.install()
    R._fit_share = _cli_fake_share
    R._fit_subject = _cli_fake_subject
    study = {}

    def fake_study(ada_path, echo_vols, mask_indices, t2_map, nit_map, fun_map, TEeffs, fit, fit_params, prior, norm, sub, ses,
                   sim, **kw):  # what the figures would be drawn from
        study.update(rows=None if kw.get("rows") is None else np.array(kw["rows"]), picks=kw.get("picks"),
                     nit=np.array(nit_map), fun=np.array(fun_map), mask_indices=np.array(mask_indices), have_vols=echo_vols is not None)
        return []

    convergence.convergence_study = fake_study
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        R.main(["--path", root, "--csv", "log.csv", "--in_vivo", "--gaussian", "--lf", "--sim", "d1", "--TEs", "114", "202", "299",
                "--gpus", str(world)] + list(extra_args))
    out[rank] = {"files": {os.path.relpath(p, root): np.array(img.arr) for p, img in sitk.written.items()},
                 "decoded": dict(R.DECODED), "stdout": buf.getvalue(), "study": study}


def _cli_tree(tmp_path, n_subjects, shape):
    import pandas as pd

    from fetal_t2mapping_amd import cli as R

    root = str(tmp_path)
    bids = os.path.join(root, "projects") + "/"
    os.makedirs(os.path.join(bids, "prj-950"))
    os.makedirs(os.path.join(root, "dicom", "logs"))
    rng = np.random.default_rng(9)
    rows, data = [], {}
    for s in range(n_subjects):
        vols = rng.normal(100.0, 10.0, size=(3,) + shape).astype(np.float32)
        masks = [(rng.random(shape) < 0.5).astype(np.uint8) for _ in range(3)]  # per-echo masks differ: the union matters
        data[f"sub-{s + 1:03d}"] = (vols, masks)
        for i, t in enumerate((114, 202, 299)):
            acq = {"prj": "prj-950", "sub": f"sub-{s + 1:03d}", "ses": "ses-01", "run": f"run-{i + 1:02d}", "EchoTime": t / 1000.0,
                   "CoilString": "HeadNeck"}
            rows.append(acq)
            np.save(R.get_img_path(bids, acq, R.recon_dirname).replace(" ", "") + ".npy", vols[i])
            np.save(R.get_img_path(bids, acq, R.mask_dirname).replace(" ", "") + ".npy", masks[i])
    pd.DataFrame(rows).to_csv(os.path.join(root, "dicom", "logs", "log.csv"), index=False)
    return root, data


def _check_files(written, data):
    for sub, (vols, masks) in data.items():
        union = (np.sum(np.stack(masks), axis=0) > 0).astype(np.uint8)
        want = _fake_fit(vols.reshape(3, -1), union.reshape(-1))
        for j, tag in enumerate(("t2", "k", "sigma", "res")):
            key = [p for p in written if sub in p and f"_{tag}map_" in p]
            assert len(key) == 1, (sub, tag, list(written))
            assert np.array_equal(written[key[0]].reshape(-1), want[j]), (sub, tag)


@pytest.mark.parametrize("n_subjects", [1, 3])
def test_cli_gpus_control_flow_two_ranks(tmp_path, n_subjects):
    """`--gpus 2` as torch.distributed.run would start it (two ranks, gloo instead of RCCL, the fit a per-voxel
    stand-in).  One subject: the volume is shared -- each rank decodes its echo files only, the shares are swapped by
    one all-to-all, the maps are all-gathered and rank 0 alone writes the four files, equal to the stand-in applied to
    the whole volume.  Three subjects: they are dealt to the ranks (0 and 2 to rank 0, 1 to rank 1), nothing is
    exchanged, each rank writes its own files."""
    root, data = _cli_tree(tmp_path, n_subjects, (3, 40, 300))  # 36 000 voxels: three chunks, the last one ragged
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_cli_worker, args=(2, _free_port(), root, n_subjects, out, ()), nprocs=2, join=True)
    written = {**out[0]["files"], **out[1]["files"]}
    if n_subjects == 1:
        assert len(out[0]["files"]) == 4 and len(out[1]["files"]) == 0
        # decode work is split: echoes 0 and 2 (and their masks) on rank 0, echo 1 on rank 1
        assert out[0]["decoded"] == {"echo": 2, "mask": 2} and out[1]["decoded"] == {"echo": 1, "mask": 1}
    else:
        assert len(out[0]["files"]) == 8 and len(out[1]["files"]) == 4
        assert all("sub-002" in p for p in out[1]["files"]) and not any("sub-002" in p for p in out[0]["files"])
        assert out[0]["decoded"] == {"echo": 6, "mask": 6} and out[1]["decoded"] == {"echo": 3, "mask": 3}
    _check_files(written, data)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_cli_shared_volume_status_and_plots(tmp_path, world):
    """One subject over 2 / 3 / 4 ranks (4: more ranks than echo files -- rank 3 decodes nothing and still fits its
    share): maps bit-identical to the whole-volume stand-in, each rank decodes 1/G of the files, the per-voxel status is
    gathered -- so the `FAIL : Optimization failed for N voxels` line (run_t2mapping.py:298-303) reports the true count,
    on every rank -- and --plots gets nit / fun maps plus the sampled voxels' rows from the ranks that hold each echo."""
    root, data = _cli_tree(tmp_path, 1, (3, 40, 300))
    vols, masks = data["sub-001"]
    union = (np.sum(np.stack(masks), axis=0) > 0).astype(np.uint8).reshape(-1)
    e = vols.reshape(3, -1)
    n_fail = int(np.sum(_fake_status(e, union) == 2))
    assert n_fail > 100
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_cli_worker, args=(world, _free_port(), root, 1, out, ("--plots", "--plot_seed", "3")), nprocs=world, join=True)
    _check_files(out[0]["files"], data)
    for r in range(world):
        assert len(out[r]["files"]) == (4 if r == 0 else 0)
        n_mine = len(range(r, 3, world))
        assert out[r]["decoded"] == {"echo": n_mine, "mask": n_mine}, (r, out[r]["decoded"])
        assert f"FAIL : Optimization failed for {n_fail} voxels" in out[r]["stdout"]
    st = out[0]["study"]
    idx = np.flatnonzero(union)
    assert not st["have_vols"] and np.array_equal(st["mask_indices"], idx)
    sel = idx[list(st["picks"][0]) + list(st["picks"][1])]
    assert len(sel) == 70 and np.array_equal(st["rows"], e[:, sel].T)
    assert np.array_equal(st["fun"].reshape(-1), np.where(union != 0, e[1] * 0.25, 0.0).astype(np.float32))
    assert np.array_equal(st["nit"].reshape(-1), np.where(union != 0, (np.abs(e[0]) % 17).astype(np.int32) + 1, 0))


def _exchange_worker(rank, world, port, n_vox, n_te, chunk, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(8)
    echoes = rng.normal(size=(n_te, n_vox)).astype(np.float32)
    masks = (rng.random((n_te, n_vox)) < 0.3).astype(np.uint8)
    mine = t2dist.echoes_of_rank(n_te, rank, world)
    share = t2dist.exchange_echo_shares(torch.from_numpy(echoes[mine].reshape(len(mine), n_vox)), n_te, n_vox, chunk).numpy()
    want, _ = t2dist.take_cyclic(echoes, None, rank, world, chunk)
    part = np.zeros(n_vox, np.uint8)
    for i in mine:
        part |= masks[i]
    union = t2dist.union_mask_over_ranks(torch.from_numpy(part))
    m_share = t2dist.share_of(union, n_vox, rank, world, chunk).numpy()
    _, want_m = t2dist.take_cyclic(echoes, masks.max(axis=0), rank, world, chunk)
    out[rank] = bool(np.array_equal(share, want) and np.array_equal(m_share, want_m))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_vox,n_te,chunk", [(2, 1000, 3, 64), (3, 1001, 8, 64), (4, 10, 3, 4), (2, 1, 2, 16), (3, 5000, 6, 256)])
def test_echo_share_exchange_equals_the_cut_of_the_whole_stack(world, n_vox, n_te, chunk):
    """Each rank holds only the echoes i = rank, rank + G, ...; after ONE all-to-all it holds every echo of its share of
    the voxels -- bit for bit what take_cyclic cuts out of the complete stack -- and the all-reduced union mask's share
    equals the share of the union (ragged sizes, more ranks than echoes, uneven echo counts per rank)."""
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, _free_port(), n_vox, n_te, chunk, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)) and len(out) == world
