"""Voxel-slab partition across the GPUs of one node + the final all-gather of the maps.

The reference has no distributed code (its only parallelism is ``Pool(20).map`` over voxels,
run_t2mapping.py:442-443).  Voxels are independent -- ``fit_voxel`` reads one row (:240) -- so the path
shards by flat C-order voxel index with no exchange during the fit; north_star asks for one RCCL
all-gather of the output maps at the end.  One process per GPU (``torch.distributed``, backend
"nccl" = RCCL on ROCm; "gloo" on CPU for the tests of this module's host logic).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

N_MAPS = 4  # t2, k, sigma, res (utils/t2map_utils.py:18-29)


def slab_len(n_vox: int, world: int) -> int:
    """Voxels per rank: ceil(N / G); the last slab is padded (mask 0) so all-gather counts are equal."""
    return int(math.ceil(n_vox / world)) if n_vox else 0


def slab_range(n_vox: int, rank: int, world: int) -> Tuple[int, int]:
    """Half-open flat-index range ``[lo, hi)`` of rank ``rank``; equals Z-slabs when Z % world == 0."""
    per = slab_len(n_vox, world)
    lo = min(rank * per, n_vox)
    return lo, min(lo + per, n_vox)


def take_slab(echoes: np.ndarray, mask: Optional[np.ndarray], rank: int, world: int):
    """Cut rank's padded slab out of a TE-major ``(nTE, N)`` stack (+ ``(N,)`` mask) on the host.

    Returns ``(echoes_slab (nTE, per) f32, mask_slab (per,) u8)``; padding voxels have mask 0.
    """
    n_te, n = echoes.shape
    per = slab_len(n, world)
    lo, hi = slab_range(n, rank, world)
    e = np.zeros((n_te, per), np.float32)
    e[:, : hi - lo] = echoes[:, lo:hi]
    m = np.zeros(per, np.uint8)
    m[: hi - lo] = 1 if mask is None else (np.asarray(mask).reshape(-1)[lo:hi] != 0)
    return e, m


def gather_maps(packed_local, n_vox: int, group=None):
    """All-gather the per-rank packed maps ``[N_MAPS, per]`` into ``[N_MAPS, n_vox]`` on every rank.

    ``packed_local``: torch tensor (CUDA with RCCL, CPU with gloo).  One collective moves all maps.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = packed_local.shape[1]
    assert packed_local.shape[0] == N_MAPS and per == slab_len(n_vox, world)
    packed_local = packed_local.contiguous()
    gathered = torch.empty((world, N_MAPS, per), dtype=packed_local.dtype, device=packed_local.device)
    dist.all_gather_into_tensor(gathered.view(-1), packed_local.view(-1), group=group)
    # [world, N_MAPS, per] -> [N_MAPS, world*per] -> trim the padding of the last slab
    return gathered.permute(1, 0, 2).reshape(N_MAPS, world * per)[:, :n_vox]


# ---- cyclic partition: chunks of CHUNK consecutive voxels dealt to the ranks group by group -----------------------
# Contiguous slabs carry unequal work whenever the mask is not uniform along Z: on the ellipsoidal synthetic brain
# the two middle slabs of eight hold 1.56x the average number of masked voxels, which alone caps strong scaling of
# ONE volume at 5.1x on eight GPUs.  Here the volume is cut into chunks of 16 Ki voxels (64 image rows of 256); every
# group of `world` consecutive chunks is dealt one chunk per rank, the deal rotated by a hash of the group number so
# that no rank keeps meeting the same part of the image rows or slices (a plain round-robin hands rank r the same
# quarter of every slice of a 256 x 256 image: 22 % imbalance on a disc).  Each rank still reads long contiguous
# runs (64 KiB per echo per chunk); the gathered maps come back into voxel order through one index gather.
CHUNK = 16384


def _rotation(group, world: int):
    """Rotation of the deal of chunk group ``group`` (array or int): a multiplicative hash, identical everywhere."""
    return ((np.asarray(group, dtype=np.int64) * 2654435761) % (1 << 32) >> 7) % world


def cyclic_len(n_vox: int, world: int, chunk: int = CHUNK) -> int:
    """Voxels per rank under the cyclic partition (a whole number of chunks; the tail is padding with mask 0)."""
    n_chunks = -(-n_vox // chunk)
    return -(-n_chunks // world) * chunk


def cyclic_chunks(n_vox: int, rank: int, world: int, chunk: int = CHUNK) -> np.ndarray:
    """Chunk numbers of rank ``rank``, slot by slot (int64; may point past the last real chunk: padding)."""
    groups = np.arange(cyclic_len(n_vox, world, chunk) // chunk, dtype=np.int64)
    return groups * world + (rank - _rotation(groups, world)) % world


def cyclic_index(n_vox: int, rank: int, world: int, chunk: int = CHUNK) -> np.ndarray:
    """Flat voxel index of every slot of rank ``rank`` (int64, length ``cyclic_len``); -1 marks padding."""
    c = cyclic_chunks(n_vox, rank, world, chunk)[:, None]
    idx = (c * chunk + np.arange(chunk, dtype=np.int64)[None, :]).reshape(-1)
    idx[idx >= n_vox] = -1
    return idx


def take_cyclic(echoes: np.ndarray, mask: Optional[np.ndarray], rank: int, world: int, chunk: int = CHUNK):
    """Rank's share of a TE-major ``(nTE, N)`` stack under the cyclic partition: ``(echoes (nTE, per) f32,
    mask (per,) u8)``.  Copies chunk by chunk (contiguous runs), never the whole volume."""
    n_te, n = echoes.shape
    per = cyclic_len(n, world, chunk)
    e = np.zeros((n_te, per), np.float32)
    m = np.zeros(per, np.uint8)
    flat_mask = None if mask is None else np.asarray(mask).reshape(-1)
    for j, c in enumerate(cyclic_chunks(n, rank, world, chunk)):
        lo = int(c) * chunk
        if lo >= n:
            continue
        hi = min(lo + chunk, n)
        e[:, j * chunk: j * chunk + hi - lo] = echoes[:, lo:hi]
        m[j * chunk: j * chunk + hi - lo] = 1 if flat_mask is None else (flat_mask[lo:hi] != 0)
    return e, m


def gather_maps_cyclic(packed_local, n_vox: int, chunk: int = CHUNK, group=None):
    """All-gather of the per-rank packed maps ``[R, per]`` of the cyclic partition (R = N_MAPS, or more rows when
    per-voxel extras travel along; any dtype) into ``[R, n_vox]`` in voxel order on every rank: one collective, then
    one gather of whole chunks (chunk ``g*world + j`` sits in slot ``g`` of rank ``(j + rotation(g)) % world``)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    N_MAPS, per = packed_local.shape  # noqa: N806  (rows of this call)
    assert per == cyclic_len(n_vox, world, chunk)
    packed_local = packed_local.contiguous()
    gathered = torch.empty((world, N_MAPS, per), dtype=packed_local.dtype, device=packed_local.device)
    dist.all_gather_into_tensor(gathered.view(-1), packed_local.view(-1), group=group)
    slots = per // chunk
    c = np.arange(slots * world, dtype=np.int64)
    g = c // world
    src = ((c % world + _rotation(g, world)) % world) * slots + g          # row of [world * slots] holding chunk c
    src_t = torch.from_numpy(src).to(gathered.device)
    by_chunk = gathered.view(world, N_MAPS, slots, chunk).permute(1, 0, 2, 3).reshape(N_MAPS, world * slots, chunk)
    return by_chunk.index_select(1, src_t).reshape(N_MAPS, world * per)[:, :n_vox]


# ---- one volume, G ranks, each rank decodes only some of the echo FILES ------------------------------------------------
# The file edge dominates end to end (gzip decode of 8 x 256^3 float32 is seconds on a CPU, the fit
# tens of milliseconds), so when one volume is shared by G ranks each rank decodes the echo files i = rank, rank + G, ...
# only (config 4 on 8 GPUs: one 377 MB file instead of eight), and the ranks then swap voxel shares of their echoes with
# ONE all-to-all (RCCL over xGMI: every rank sends each peer that peer's chunks of its echoes): afterwards every rank
# holds all echoes of its own share of the voxels, which is what the fit needs.
def echoes_of_rank(n_te: int, rank: int, world: int) -> list:
    """Echo files rank ``rank`` decodes: i = rank, rank + world, ..."""
    return list(range(rank, n_te, world))


def exchange_echo_shares(mine, n_te: int, n_vox: int, chunk: int = CHUNK, group=None):
    """``mine``: torch float32 ``[len(echoes_of_rank), n_vox]`` -- the whole volumes of the echoes this rank decoded, on
    the device the backend moves (CUDA with RCCL, CPU with gloo).  Returns ``[n_te, per]``: every echo of this rank's
    share of the voxels under the cyclic partition (``cyclic_index``; padding voxels are 0).  One ``all_to_all_single``."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    per = cyclic_len(n_vox, world, chunk)
    slots = per // chunk
    n_max = -(-n_te // world)                      # echoes per rank, padded to the same number everywhere
    n_mine = len(echoes_of_rank(n_te, rank, world))
    assert tuple(mine.shape) == (n_mine, n_vox), (tuple(mine.shape), n_mine, n_vox)
    by_chunk = torch.zeros((n_mine, slots * world * chunk), dtype=torch.float32, device=mine.device)
    by_chunk[:, :n_vox] = mine
    by_chunk = by_chunk.view(n_mine, slots * world, chunk)
    send = torch.zeros((world, n_max, slots, chunk), dtype=torch.float32, device=mine.device)
    for d in range(world):                         # what peer d fits: its chunks of my echoes
        sel = torch.from_numpy(cyclic_chunks(n_vox, d, world, chunk)).to(mine.device)
        if n_mine:
            send[d, :n_mine] = by_chunk.index_select(1, sel)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    out = torch.empty((n_te, per), dtype=torch.float32, device=mine.device)
    for i in range(n_te):                          # echo i was decoded by rank i % world as its (i // world)-th
        out[i] = recv[i % world, i // world].reshape(-1)
    return out


def union_mask_over_ranks(partial, group=None):
    """run_t2mapping.py:383-384 when the mask files are spread over the ranks: ``partial`` = uint8 ``[n_vox]`` union of
    the masks this rank read (zeros if it read none) -> the union over all ranks, on every rank (one all-reduce)."""
    import torch.distributed as dist

    dist.all_reduce(partial, op=dist.ReduceOp.MAX, group=group)
    return partial


def share_of(flat, n_vox: int, rank: int, world: int, chunk: int = CHUNK):
    """This rank's share ``[per]`` of a full-volume vector ``[n_vox]`` (torch; padding = 0) under the cyclic partition."""
    import torch

    per = cyclic_len(n_vox, world, chunk)
    padded = torch.zeros((per * world,), dtype=flat.dtype, device=flat.device)
    padded[:n_vox] = flat
    sel = torch.from_numpy(cyclic_chunks(n_vox, rank, world, chunk)).to(flat.device)
    return padded.view(per * world // chunk, chunk).index_select(0, sel).reshape(-1)


def rows_from_owners(sel, my_vols: dict, n_te: int, device, group=None):
    """Samples of the voxels ``sel`` (flat indices) at every echo, assembled from the ranks that decoded each echo:
    ``my_vols`` maps echo index -> this rank's host volume.  Returns float32 numpy ``(len(sel), n_te)`` on every rank
    (one small all-reduce)."""
    import torch
    import torch.distributed as dist

    rows = np.zeros((len(sel), n_te), np.float32)
    for i, v in my_vols.items():
        rows[:, i] = np.asarray(v).reshape(-1)[sel]
    t = torch.from_numpy(rows).to(device)
    dist.all_reduce(t, group=group)
    return t.cpu().numpy()


def fit_volume_sharded(echoes, mask, TEeffs, fit, fit_params, prior=True, norm=False, *, solver="lbfgsb",
                       precision="f64", group=None, partition="cyclic", chunk: int = CHUNK):
    """One volume over the GPUs of the group: each rank fits its share on its own GPU, one all-gather (RCCL over
    xGMI) assembles the four maps on every rank, and all ranks return the complete ``T2Maps`` (torch CUDA tensors
    shaped ``(Z, Y, X)``).

    ``echoes``: host ``(nTE, Z, Y, X)`` float32 stack, complete on every rank (numpy, a memory map or a pinned block;
    this rank's share is copied out of it chunk by chunk).  A caller that starts from FILES should not decode them all
    on every rank: ``cli._fit_subject_shared`` decodes 1/G of the echo files per rank and swaps shares with
    ``exchange_echo_shares``.  ``partition``: ``"cyclic"`` (chunks dealt round-robin: balanced whatever the mask looks like) or
    ``"slab"`` (contiguous flat ranges, ``slab_range``: equals Z-slabs when Z divides by the group size)."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from . import _abi
    from ._lib import check, require_gpu
    from .t2map import T2Maps, make_config

    lib = require_gpu()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    spatial = echoes.shape[1:]
    n = int(np.prod(spatial))
    flat = echoes.reshape(echoes.shape[0], n)
    if partition == "cyclic":
        e, m = take_cyclic(flat, mask, rank, world, chunk)
    elif partition == "slab":
        e, m = take_slab(flat, mask, rank, world)
    else:
        raise ValueError(f"unknown partition {partition!r}")
    dev = torch.device("cuda", torch.cuda.current_device())
    e_d, m_d = torch.from_numpy(e).to(dev), torch.from_numpy(m).to(dev)
    per = e.shape[1]
    packed = torch.empty((N_MAPS, per), dtype=torch.float32, device=dev)
    cfg = make_config(fit, fit_params, TEeffs, prior, norm, solver, precision)
    maps = _abi.T2FitMaps()
    maps.t2, maps.k, maps.sigma, maps.res = (packed[j].data_ptr() for j in range(N_MAPS))
    st = torch.cuda.current_stream().cuda_stream
    check(lib.t2fit_volume_dev(C.byref(cfg), e_d.data_ptr(), _abi.LAYOUT_TE_MAJOR, m_d.data_ptr(), per,
                               C.byref(maps), C.c_void_p(st)))
    full = gather_maps_cyclic(packed, n, chunk, group) if partition == "cyclic" else gather_maps(packed, n, group)
    return T2Maps(*(full[j].reshape(spatial) for j in range(N_MAPS)))


def subjects_of_rank(n_subjects: int, rank: int, world: int) -> range:
    """BASELINE.json config 5 (32 subjects over 8 GPUs): subject s belongs to rank ``s % world``.  Whole
    subjects are independent, so nothing is exchanged: every rank streams its own subjects host->HBM->host
    (``stream.fit_subjects``) and writes its own output files."""
    if not 0 <= rank < world:
        raise ValueError("rank outside [0, world)")
    return range(rank, n_subjects, world)


def fit_subjects_round_robin(load_subject, n_subjects: int, TEeffs, fit, fit_params, prior=True, norm=False, *,
                             rank: Optional[int] = None, world: Optional[int] = None, **stream_kw):
    """Yield ``(s, T2Maps)`` for the subjects of this rank, streamed double-buffered through this rank's GPU.

    ``load_subject(s) -> (echoes (nTE,Z,Y,X), mask or None)`` is called lazily, one subject ahead of the
    fit (e.g. ``nifti.read_stack`` into pinned memory).  ``rank`` / ``world`` default to the initialised
    ``torch.distributed`` group, or to a single process."""
    from . import stream

    if rank is None or world is None:
        import torch.distributed as dist

        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    mine = subjects_of_rank(n_subjects, rank, world)
    fitted = stream.fit_subjects((load_subject(s) for s in mine), TEeffs, fit, fit_params, prior, norm, **stream_kw)
    for s, maps in zip(mine, fitted):
        yield s, maps
