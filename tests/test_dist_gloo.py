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
