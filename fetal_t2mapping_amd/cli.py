"""CLI and volume driver: the reference's ``run_t2mapping.py`` surface on the MI355X fit.

    python -m fetal_t2mapping_amd.cli --path /data/qMRI --csv 2024083017_17510000.csv \
        --in_vivo --gaussian --lf --sim 1 [--TEs 114 202 299] [--no_prior] [--solver lbfgsb|lm|loglin] [--gpus N]

--gpus N (N > 1) starts one process per GPU (torch.distributed.run, RCCL) before anything touches a GPU: with at
least N subjects in the CSVs each rank streams its own subjects (dist.subjects_of_rank: nothing is exchanged,
BASELINE.json config 5); with fewer, every volume is shared: each rank decodes 1/N of its echo files, one all-to-all
hands every rank all echoes of its balanced share of the voxels, the maps (and the per-voxel status) are all-gathered
(dist.exchange_echo_shares / gather_maps_cyclic, config 4); rank 0 writes the files of a shared volume.

Same flags, metadata CSVs, input/output file names and maps as the reference
(run_t2mapping.py:483-576, utils/metadata_utils.py, utils/qmri_utils.py:13-33,
utils/t2map_utils.py:18-59); the voxel loop (:411-461) is one call into the HIP library.  NIfTI I/O
stays SimpleITK on the host, as in the reference (nifti.py stands in where SimpleITK is not installed).
The convergence-study figures (:465-468) are written on request (--plots, convergence.py).  ``--csv prj-004``
(prj-003, prj-002) stands for the session logs of that project of the reference's paper, as in
utils/metadata_utils.py:19-85.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

from . import t2map

# derivative directory names (utils/metadata_utils.py:4-17)
recon_dirname = "recon_1mm"
mask_dirname = "recon_1mm_mask"
phantom_labels_dirname = "recon_1mm_label"
t2map_dirname = recon_dirname + "_t2map"


def _sitk():
    """NIfTI I/O: SimpleITK as in the reference when it is importable, else the package's own NIfTI-1
    reader/writer (same five calls, same (Z,Y,X) arrays and LPS geometry; fetal_t2mapping_amd/nifti.py)."""
    try:
        import SimpleITK as sitk
    except ImportError:
        from . import nifti as sitk
    return sitk


_pinned = {}


def _pinned_stack(recon_paths):
    """Page-locked float32 staging block for one subject's echoes (kept for the next subject of the same size):
    the decoder writes into it and the host->device copy is a DMA from where the samples lie.  None when torch
    or a GPU is not there (the fit would fail loudly later anyway)."""
    try:
        import torch

        from . import nifti

        if not torch.cuda.is_available():
            return None
        with open(recon_paths[0], "rb") as f:
            head = f.read(4096)
        if head[:2] == b"\x1f\x8b":
            import zlib

            head = zlib.decompressobj(wbits=31).decompress(head, 352)
        n = len(recon_paths) * int(np.prod(nifti._parse_header(head).shape))
        if _pinned.get("n") != n:
            _pinned.clear()
            _pinned.update(n=n, buf=torch.empty(n, dtype=torch.float32).pin_memory())
        return _pinned["buf"].numpy()
    except Exception:
        return None


DECODED = {"echo": 0, "mask": 0}  # volume files this process has decoded (the shared-volume tests assert 1/G of them)


def _read_geometry(sitk, path):
    """Spacing / origin / direction of the image at `path` (an object with the three Get* methods) without decoding the
    voxels where the reader can: the maps carry the geometry of the LAST echo's image (run_t2mapping.py:377 /
    utils/t2map_utils.py:22-24), which the rank that writes them may not have decoded."""
    from . import nifti

    if sitk is nifti:
        return nifti.ReadGeometry(path)
    if hasattr(sitk, "ImageFileReader"):  # SimpleITK: header only
        r = sitk.ImageFileReader()
        r.SetFileName(path)
        r.ReadImageInformation()
        return nifti.Image(np.zeros((0, 0, 0), np.float32), r.GetSpacing(), r.GetOrigin(), r.GetDirection())
    return sitk.ReadImage(path)


def _read_subject(sitk, recon_paths, mask_paths, label_path):
    """All volumes of one (sub, ses): echoes, masks, optional vial labels, and the last recon image (its
    geometry goes onto the maps, run_t2mapping.py:377 / utils/t2map_utils.py:22-24).  With the native
    reader the files are decoded concurrently, the echoes straight into one float32 (nTE,Z,Y,X) block."""
    from . import nifti

    DECODED["echo"] += len(recon_paths)
    DECODED["mask"] += len(mask_paths)
    if not recon_paths:  # a rank of a shared volume with more ranks than echoes
        label = sitk.GetArrayFromImage(sitk.ReadImage(label_path)) if label_path else None
        return [], [], label, None

    if sitk is nifti:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(8) as pool:
            masks_f = pool.map(nifti.ReadImage, mask_paths)
            label_f = pool.submit(nifti.ReadImage, label_path) if label_path else None
            stack, images = nifti.read_stack(recon_paths, out=_pinned_stack(recon_paths))
            masks = [m.arr for m in masks_f]
            label = label_f.result().arr if label_f else None
        return list(stack), masks, label, images[-1]
    vols, masks, recon_img = [], [], None
    for rp, mp in zip(recon_paths, mask_paths):
        recon_img = sitk.ReadImage(rp)
        masks.append(sitk.GetArrayFromImage(sitk.ReadImage(mp)))
        vols.append(sitk.GetArrayFromImage(recon_img))
    label = sitk.GetArrayFromImage(sitk.ReadImage(label_path)) if label_path else None
    return vols, masks, label, recon_img


# ---- metadata / paths --------------------------------------------------------------------------
def mk_bids_dir(bids_dir, *dirs):
    """utils/dcm_utils.py:189-195 (there: `if not exists: mkdir`; with --gpus N several ranks create the shared
    parents of their subjects' directories at the same moment, so an existing directory is not an error here)."""
    path = bids_dir
    for d in dirs:
        path = os.path.join(path, d)
        os.makedirs(path, exist_ok=True)


def get_img_path(bids_path, acq, type: str = "anat"):
    """File naming of utils/qmri_utils.py:13-33: raw images under <prj>/<sub>/<ses>/anat, everything
    else under <prj>/derivatives/<type>/<sub>/<ses>/anat; recon-type names carry the echo time (width 3)."""
    sub, ses = acq["sub"], acq["ses"]
    derived = [acq["prj"], "derivatives", type, sub, ses, "anat"]
    if type == "anat":
        dirs, stem = [acq["prj"], sub, ses, "anat"], [sub, ses, acq["run"] + "_T2w.nii.gz"]
    elif "t2map" in type:
        dirs, stem = derived, [sub, ses, type + ".nii.gz"]
    elif "recon" in type:
        simulated = acq["CoilString"] == "Simulation"
        te_field = f"te-{int(acq['EchoTime'] if simulated else acq['EchoTime'] * 1000):3}"
        dirs, stem = derived, [sub, ses] + ([f"t2-{int(acq['T2']):3}"] if simulated else []) + [te_field, type + ".nii.gz"]
    else:
        dirs, stem = derived, [sub, ses, acq["run"], "T2w", type + ".nii.gz"]
    mk_bids_dir(bids_path, *dirs)
    return os.path.join(bids_path, *dirs, "_".join(stem))


# `--csv prj-00X` instead of file names: the session logs of the three projects of the reference's paper
# (utils/metadata_utils.py:19-85), keyed by (project, low_field).  Data, not logic: the file names as the reference lists them
# (entries it has commented out are left out here too).  None: the reference has no such data and exits.
_PROJECT_LOGS = {
    ("prj-004", True): "2024083017_17510000 2024090320_55420000 2024090618_37050000 2024090811_14320000 2024091017_53530000_1 "
                       "2024091017_53530000_2 2024091020_45220000 2024091320_23400000 2024091321_22550000 2024091322_27490000 "
                       "2024092720_10110000 2024092719_10310000 2024102120_48480000",
    ("prj-004", False): "2024083019_26300000 2024090322_28560000 2024090619_26370000 2024090812_21470000 2024091021_57280000 "
                        "2024091319_13240000 2024091318_13560000 2024092721_25410000 2024102616_18560000 2024102122_28450000",
    ("prj-003", True): "20240806_30540000_1",
    ("prj-003", False): None,
    ("prj-002", True): "20240527_095111_2",
    ("prj-002", False): "20240609_50140000_2",
}
_PROJECT_BANNERS = {
    "prj-004": ["PRJ-004 - In vivo adult brain data acquired using the head coil"],
    "prj-003": ["PRJ-003 - In vitro NIST Phantom data acquired using the abdominal coil M",
                "Notes: only data selected for paper submission are processed."],
    "prj-002": ["PRJ-002 - In vitro NIST Phantom data acquired using the head coil.",
                "Notes: only data selected for paper submission are processed."],
}


def project_csvs(project: str, low_field: bool):
    """utils/metadata_utils.py:19-85 (`prj_004` / `prj_003` / `prj_002`): the CSV list a project name stands for."""
    logs = _PROJECT_LOGS[(project, bool(low_field))]
    if logs is None:
        print("Error: no data to process yet at 1.5 T.")
        raise SystemExit(1)
    return [name + ".csv" for name in logs.split()]


def set_metadata(csv_path, csvs, low_field):
    """utils/metadata_utils.py:92-125: concatenate the session log CSVs into one DataFrame; `--csv prj-004` (or
    prj-003 / prj-002) stands for that project's list of logs, with the reference's banner."""
    import pandas as pd

    if csvs[0] in _PROJECT_BANNERS:
        lines = _PROJECT_BANNERS[csvs[0]]
        rule = "*" * max(len(ln) for ln in lines)
        print("\n".join([rule] + lines + [rule]))
        csvs = project_csvs(csvs[0], low_field)
    elif ".csv" not in csvs[0].lower():
        print(f"Error: {csvs} is not a valid metadata log file nor a valid project to process (only prj-002, prj-003 and "
              "prj-004 metadata can be processed all at once.)")
        raise SystemExit(1)
    return pd.concat([pd.read_csv(os.path.join(csv_path, c)) for c in csvs])


def set_phantom_gt(low_field):
    """run_t2mapping.py:14-27 (NMR ground-truth T2 of the NIST phantom vials); returns (gt, id)."""
    if low_field:
        gt = [594, 416, 284, 221, 167, 122, 80, 53, 41]
        ids = ["T2-3", "T2-4", "T2-5", "T2-6", "T2-7", "T2-8", "T2-9", "T2-10", "T2-11"]
    else:
        gt = [1044, 624, 428, 258, 186, 137, 90, 63, 44, 27, 19, 15, 10, 8]
        ids = [f"T2-{i}" for i in range(1, 15)]
    return gt, ids


# ---- outputs -----------------------------------------------------------------------------------
def save_nifti_maps(t2_map, k_map, sigma_map, res_map, dirname, recon_img, bids_path, acq, sim, analysis):
    """utils/t2map_utils.py:18-29."""
    sitk = _sitk()
    items = []
    for arr, tag in zip([t2_map, k_map, sigma_map, res_map], ["t2", "k", "sigma", "res"]):
        img = sitk.GetImageFromArray(arr)
        img.SetSpacing(recon_img.GetSpacing())
        img.SetOrigin(recon_img.GetOrigin())
        img.SetDirection(recon_img.GetDirection())
        path = get_img_path(bids_path, acq.iloc[0], dirname)
        path = path.replace("t2map.nii.gz", "sim-" + str(sim) + f"_{tag}map_ada-{analysis}.nii.gz")
        items.append((img, path))
    if hasattr(sitk, "WriteImages"):  # native writer: the four maps are compressed concurrently
        sitk.WriteImages(items)
    else:
        for img, path in items:
            sitk.WriteImage(img, path)
    print(f"T2 map saved as nifti file in {dirname}")


def phantom_frame(stats, id, gt):
    """The table ``save_phantom_csv`` writes (utils/t2map_utils.py:55-66): ``stats`` maps the six column names to
    per-vial values.  ``np.nanmean`` / ``np.nanstd`` of a float32 map are float32 numbers (stored into float64 arrays,
    :37-48), so the values are rounded to float32 here; the text pandas then writes has the reference's digits."""
    import pandas as pd

    cols = {name: np.asarray(stats[name], np.float64).astype(np.float32).astype(np.float64)
            for name in ("meanT2", "stdT2", "meanK", "stdK", "meanC", "stdC")}
    return pd.DataFrame({"id": id, "trueT2": gt, **cols})


def save_phantom_csv(t2_map, k_map, sigma_map, label, id, gt, bids_path, acq, dirname, sim, analysis, device=0):
    """utils/t2map_utils.py:30-59: nanmean / nanstd of each map per vial label, reduced on the GPU
    (t2fit_label_stats_dev: mean, then mean squared deviation from it, as numpy computes them -- accumulated in float64
    where numpy sums the float32 values in float32, then rounded to the float32 numpy returns: the last float32 digit can
    differ from the reference's file, nothing more)."""
    n_roi = len(gt)
    stats = {}
    for arr, m, s in ((t2_map, "meanT2", "stdT2"), (k_map, "meanK", "stdK"), (sigma_map, "meanC", "stdC")):
        stats[m], stats[s], _ = t2map.label_stats(arr, label, n_roi, device=device)
    path = get_img_path(bids_path, acq.iloc[0], dirname).replace("t2map.nii.gz", f"sim-{sim}_ROI_data_ada-{analysis}.csv")
    phantom_frame(stats, id, gt).to_csv(path, index=False)


# ---- driver ------------------------------------------------------------------------------------
def _fit_subject(vols, masks, keep, te_eff, fit, fit_params, prior, norm, solver, precision, device, numpy_legacy=False):
    """One (sub, ses): union mask + flat indices on the device (bit-identical to
    run_t2mapping.py:383-384,412,421), fit, maps back to the host as (Z,Y,X) float32."""
    import torch

    echoes, mask, _ = t2map.stack_mask_flatten(vols, masks, device=device)
    if keep is not None:
        mask = mask & keep
    shape = mask.shape
    mask_d = torch.from_numpy(mask.astype(np.uint8).reshape(-1)).to(echoes.device)
    maps = t2map.fit_volume(echoes.reshape((len(vols),) + shape), mask_d, te_eff, fit, fit_params, prior=prior,
                            norm=norm, solver=solver, precision=precision, extras=True, numpy_legacy=numpy_legacy)
    torch.cuda.synchronize()
    out = tuple(getattr(maps, n).cpu().numpy() for n in ("t2", "k", "sigma", "res"))
    extras = {"nit": maps.nit.cpu().numpy(), "fun": maps.fun.cpu().numpy()}  # what the convergence figures need
    return mask, out, maps.status.cpu().numpy(), extras


def _dist_env():
    """(rank, world, local_rank) of this process: set by torch.distributed.run, else a single process."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _fit_share(e_share, m_share, te_eff, fit, fit_params, prior, norm, solver, precision, device, numpy_legacy=False):
    """This rank's share of a volume ((nTE, per) float32 and (per,) uint8: CUDA tensors, or host arrays that are sent
    first) -> ``(packed [6, per] float32 on the GPU: t2, k, sigma, res, fun, nit (int32 bit patterns), status (per,)
    uint8)``.  The only function of the shared-volume path that calls the fit."""
    import ctypes as C

    import torch

    from . import _abi
    from ._lib import check, require_gpu

    lib = require_gpu()
    dev = torch.device("cuda", device)
    with torch.cuda.device(dev):
        e_d = (e_share if torch.is_tensor(e_share) else torch.from_numpy(e_share)).to(dev, non_blocking=True).contiguous()
        m_d = (m_share if torch.is_tensor(m_share) else torch.from_numpy(m_share)).to(dev, non_blocking=True).contiguous()
        per = e_d.shape[1]
        packed = torch.empty((6, per), dtype=torch.float32, device=dev)
        status = torch.empty(per, dtype=torch.uint8, device=dev)
        cfg = t2map.make_config(fit, fit_params, te_eff, prior, norm, solver, precision, numpy_legacy)
        maps = _abi.T2FitMaps()
        maps.t2, maps.k, maps.sigma, maps.res, maps.fun, maps.nit = (packed[j].data_ptr() for j in range(6))
        maps.status = status.data_ptr()
        check(lib.t2fit_volume_dev(C.byref(cfg), e_d.data_ptr(), _abi.LAYOUT_TE_MAJOR, m_d.data_ptr(), per, C.byref(maps),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
    return packed, status


def _fit_subject_shared(sitk, recon_paths, mask_paths, label_path, fast, te_eff, fit, fit_params, prior, norm, solver,
                        precision, device, numpy_legacy=False):
    """One (sub, ses) over all ranks (BASELINE.json config 4, or any run with fewer subjects than ranks).

    Rank r decodes the echo files r, r + G, ... and their mask files only; the union mask (run_t2mapping.py:383-384) is
    an all-reduce of the partial unions; one all-to-all hands every rank all echoes of its balanced share of the voxels
    (dist.exchange_echo_shares); every rank fits its share; one all-gather assembles t2 / k / sigma / res / fun / nit on
    every rank and a second, a quarter its size, the status bytes -- so the `FAIL : Optimization failed` count
    (run_t2mapping.py:298-303) and the convergence figures work here as in the single-process run.
    Returns ``(mask, (t2, k, sigma, res), status, extras, recon_img, label, my_vols)``."""
    import torch
    import torch.distributed as dist

    from . import dist as t2dist

    rank, world = dist.get_rank(), dist.get_world_size()
    n_te = len(recon_paths)
    mine = t2dist.echoes_of_rank(n_te, rank, world)
    vols, masks, label, _ = _read_subject(sitk, [recon_paths[i] for i in mine], [mask_paths[i] for i in mine], label_path)
    recon_img = _read_geometry(sitk, recon_paths[-1])
    # shape of the volume: from an echo this rank decoded, else from the label, else from a peer (broadcast)
    shape = [tuple(np.asarray(vols[0]).shape) if vols else None]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, shape[0])
        shape[0] = next(sh for sh in gathered if sh is not None)
    shape = tuple(shape[0])
    n = int(np.prod(shape))
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device) if on_gpu else torch.device("cpu")
    part = np.zeros(n, np.uint8)
    for m in masks:
        part |= (np.asarray(m).reshape(-1) != 0).astype(np.uint8)
    if vols:
        # (the native reader decodes into one page-locked block: sent as it lies, no staging copy)
        host = _one_block(vols) if _is_one_block(vols) else np.stack([np.asarray(v, np.float32).reshape(-1) for v in vols])
        mine_t = torch.from_numpy(host).to(dev, non_blocking=True)
    else:
        mine_t = torch.zeros((0, n), dtype=torch.float32, device=dev)
    mask_t = t2dist.union_mask_over_ranks(torch.from_numpy(part).to(dev))
    if fast and label is not None:  # run_t2mapping.py:394-400: fit the labelled vials only
        mask_t = mask_t * torch.from_numpy((np.asarray(label).reshape(-1) != 0).astype(np.uint8)).to(dev)
    e_share = t2dist.exchange_echo_shares(mine_t, n_te, n)
    m_share = t2dist.share_of(mask_t, n, rank, world)
    fitted = _fit_share(e_share, m_share, te_eff, fit, fit_params, prior, norm, solver, precision, device,
                        **({"numpy_legacy": True} if numpy_legacy else {}))
    packed, st_share = fitted if isinstance(fitted, tuple) else (fitted, None)
    if not on_gpu:
        packed = packed.cpu()
    full = t2dist.gather_maps_cyclic(packed, n).cpu().numpy()
    mask = mask_t.cpu().numpy().astype(bool).reshape(shape)
    out = tuple(np.ascontiguousarray(full[j]).reshape(shape) for j in range(4))
    if st_share is not None:
        status = t2dist.gather_maps_cyclic(st_share.view(1, -1) if on_gpu else st_share.cpu().view(1, -1), n)[0].cpu().numpy().reshape(shape)
    else:  # a fit stand-in without a status row (tests): NaN maps = infeasible bounds, everything else converged
        status = np.where(mask, np.where(np.isnan(out[0]), 4, 1), 0).astype(np.uint8)
    extras = None
    if full.shape[0] >= 6:
        extras = {"fun": np.ascontiguousarray(full[4]).reshape(shape),
                  "nit": np.ascontiguousarray(full[5]).view(np.int32).reshape(shape)}
    my_vols = {i: v for i, v in zip(mine, vols)}
    return mask, out, status, extras, recon_img, label, my_vols


def _is_one_block(vols):
    """True when the echo volumes are consecutive views of one C-contiguous float32 block (nifti.read_stack)."""
    try:
        a0 = vols[0]
        if a0.dtype != np.float32 or not a0.flags.c_contiguous:
            return False
        step = a0.nbytes
        base = a0.ctypes.data
        return all(v.dtype == np.float32 and v.flags.c_contiguous and v.ctypes.data == base + j * step for j, v in enumerate(vols))
    except AttributeError:
        return False


def _one_block(vols):
    """The (n, N) float32 view over the block `_is_one_block` recognised: no copy."""
    n = vols[0].size
    return np.lib.stride_tricks.as_strided(vols[0].reshape(-1), shape=(len(vols), n), strides=(vols[0].nbytes, 4))


def process_t2maps(metadata, bids_path, TEs, fit, fit_params, phantom, low_field, prior, fast, norm, sim,
                   solver="lbfgsb", precision="f64", device=0, plots=False, plot_seed=None, numpy_legacy=False):
    """run_t2mapping.py:333-479 with the voxel loop on the GPU.  ``plots``: also write the reference's
    convergence-study figures (:465-468) under <prj>/ada/convergence_analysis."""
    sitk = _sitk()
    tes_s = [x / 1000 for x in TEs]
    metadata = metadata[metadata["EchoTime"].isin(tes_s)]
    # more than one process (--gpus N): whole subjects are dealt to the ranks when there are enough of them, otherwise
    # every volume is shared by all ranks and rank 0 writes its files
    rank, world, _ = _dist_env()
    subjects = [(prj, sub, ses) for prj, prj_md in metadata.groupby("prj") for (sub, ses), _ in prj_md.groupby(["sub", "ses"])]
    share_volumes = world > 1 and len(subjects) < world
    mine = set(subjects if (world == 1 or share_volumes) else
               [subjects[i] for i in dist_subjects_of_rank(len(subjects), rank, world)])
    writer = (rank == 0) or not share_volumes
    for prj, prj_md in metadata.groupby("prj"):
        for (sub, ses), sub_md in prj_md.groupby(["sub", "ses"]):
            if (prj, sub, ses) not in mine:
                continue
            recon_paths, mask_paths, te_eff, label_path = [], [], [], None
            for echotime, acq in sub_md.groupby("EchoTime"):
                te_eff.append(echotime * 1000)
                recon_paths.append(get_img_path(bids_path, acq.iloc[0], recon_dirname).replace(" ", ""))
                mask_paths.append(get_img_path(bids_path, acq.iloc[0], mask_dirname).replace(" ", ""))
                if phantom:
                    label_path = get_img_path(bids_path, acq.iloc[0], phantom_labels_dirname).replace(" ", "")
            te_eff = np.array(te_eff)
            if not np.array_equal(te_eff, TEs):
                print(f"Warning: one or more TEs selected to fit is missing for {sub}_{ses}. T2 fit is skipped.")
                continue
            print(f"T2 Mapping: {prj}_{sub}_{ses}")
            print(f"TEeffs: {te_eff}")
            print(f"Fitting using {fit} model ... ")
            my_vols = None
            if share_volumes:  # every rank decodes 1/G of the echo files; shares are swapped on the way to the fit
                t0 = time.time()
                mask, maps4, status, extras, recon_img, label, my_vols = _fit_subject_shared(
                    sitk, recon_paths, mask_paths, label_path, phantom and fast, te_eff, fit, fit_params, prior, norm, solver,
                    precision, device, numpy_legacy)
                vols = None
            else:
                vols, masks, label, recon_img = _read_subject(sitk, recon_paths, mask_paths, label_path)
                keep = (label != 0) if (phantom and fast) else None  # :394-400
                t0 = time.time()
                fitted = _fit_subject(vols, masks, keep, te_eff, fit, fit_params, prior, norm, solver, precision, device,
                                      **({"numpy_legacy": True} if numpy_legacy else {}))
                mask, maps4, status = fitted[:3]
                extras = fitted[3] if len(fitted) > 3 else None
            t2_map, k_map, sigma_map, res_map = maps4
            print(f"Dimensions of the t2w images: {mask.shape + (te_eff.size,)} (z,y,x,necho)")
            print(f"Mask Dimension: {mask.shape} -  Number of voxels inside mask: {int(np.sum(mask))}")
            if np.any(status == 4):  # scipy raises here and the reference's pool.map aborts the run
                raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
            n_fail = int(np.sum((status != 1) & (status != 0)))
            if n_fail:
                print(f"FAIL : Optimization failed for {n_fail} voxels")
            print(f"... done. Time to fit: {round(time.time() - t0, 4)} sec")
            want_plots = plots and extras is not None and solver != "loglin"  # the closed form has no iterations to plot
            picks = rows = None
            if want_plots and share_volumes:
                # the sampled voxels' rows come from the ranks that decoded each echo: rank 0 draws the sample (the
                # reference's is unseeded), everybody contributes its echoes' columns
                import torch
                import torch.distributed as dist

                from . import convergence
                from . import dist as t2dist

                mask_idx = np.flatnonzero(mask.reshape(-1))
                box = [convergence.pick_voxels(len(mask_idx), plot_seed) if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                picks = box[0]
                dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
                rows = t2dist.rows_from_owners(mask_idx[list(picks[0]) + list(picks[1])], my_vols, len(te_eff), dev)
            if not writer:
                continue
            if want_plots:
                from . import convergence

                convergence.convergence_study(convergence.set_ada_path(bids_path, prj), vols,
                                              np.flatnonzero(mask.reshape(-1)), t2_map, extras["nit"], extras["fun"],
                                              te_eff, fit, fit_params, prior, norm, sub, ses, sim, solver=solver,
                                              precision=precision, device=device, seed=plot_seed, picks=picks, rows=rows,
                                              numpy_legacy=numpy_legacy)
            save_nifti_maps(t2_map, k_map, sigma_map, res_map, t2map_dirname, recon_img, bids_path, acq, sim, fit)
            if phantom:
                # the reference unpacks (gt, id) as id, gt (run_t2mapping.py:27 vs :478), which swaps the
                # CSV's `id` and `trueT2` columns; kept so the output file is identical
                id_, gt_ = set_phantom_gt(low_field)
                save_phantom_csv(t2_map, k_map, sigma_map, label, id_, gt_, bids_path, acq, t2map_dirname, sim, fit,
                                 device=device)


def dist_subjects_of_rank(n_subjects, rank, world):
    from . import dist as t2dist

    return t2dist.subjects_of_rank(n_subjects, rank, world)


_EXCLUSIVE_GROUPS = (
    (("in_vivo", "in vivo subject data"),
     ("in_vitro", "NIST phantom, full maps"),
     ("in_vitro_fast", "NIST phantom, labelled vials only")),
    (("gaussian", "2-parameter least squares  k*exp(-TE/T2)"),
     ("gaussian_rician", "3-parameter least squares  sqrt(k^2 exp(-2TE/T2) + sigma^2)"),
     ("rician", "3-parameter Rician likelihood")),
    (("lf", "0.55 T tables and default echo times"),
     ("hf", "1.5 T tables and default echo times")),
)


def parse_arguments(argv=None):
    """Flag set of run_t2mapping.py:483-518 (three required one-of groups, --sim, --TEs, --no_prior,
    --norm) plus --solver / --precision / --device."""
    p = argparse.ArgumentParser(prog="fetal_t2mapping_amd.cli", description="voxel-wise T2 mapping on an MI355X")
    p.add_argument("--path", required=True, help="root of the qMRI tree (contains projects/ and dicom/logs/)")
    p.add_argument("--csv", nargs="+", required=True, help="metadata log CSV file name(s) under dicom/logs/")
    for group in _EXCLUSIVE_GROUPS:
        g = p.add_mutually_exclusive_group(required=True)
        for flag, text in group:
            g.add_argument("--" + flag, action="store_true", help=text)
    p.add_argument("--sim", required=True, help="identifier written into the output file names")
    p.add_argument("--TEs", nargs="+", type=int, help="echo times [ms] to fit (default 114/115, 202, 299)")
    p.add_argument("--no_prior", action="store_true", help="k >= S(TE0) instead of the table's lower bound")
    p.add_argument("--norm", action="store_true", help="divide each voxel's samples by their maximum")
    p.add_argument("--solver", choices=["lbfgsb", "lm", "loglin"], default="lbfgsb",
                   help="lbfgsb: the reference's solver and stop rules (default); lm: converged bounded LM; "
                        "loglin: closed-form weighted log-linear fit (--gaussian only)")
    p.add_argument("--precision", choices=["f64", "f32"], default="f64", help="arithmetic of the lm solver")
    p.add_argument("--device", type=int, default=0, help="HIP device ordinal (one process; with --gpus each rank uses its own)")
    p.add_argument("--gpus", type=int, default=1,
                   help="GPUs of this node to use: N > 1 starts one process per GPU (torch.distributed.run, RCCL); subjects "
                        "are dealt to the ranks, or, with fewer subjects than ranks, every volume is cut over the ranks")
    p.add_argument("--numpy_legacy", action="store_true",
                   help="reproduce the reference as it runs under the numpy < 2 it freezes (requirements_frozen.txt:103): "
                        "float32 log term of the rician objective, float32 prediction of the residual map; default: numpy >= 2")
    p.add_argument("--plots", action="store_true",
                   help="write the reference's convergence-study PNGs (run_t2mapping.py:465-468) under "
                        "<prj>/ada/convergence_analysis; off by default, the reference always draws them")
    p.add_argument("--plot_seed", type=int, default=None, help="seed of the voxel sample in the figures")
    return p.parse_args(argv)


def _relaunch_per_gpu(args, argv):
    """--gpus N from a plain invocation: start N ranks of this module under torch.distributed.run as a child process
    and leave with its exit code.  Runs before anything has touched a GPU (a process that has initialised HIP must not
    spawn the ranks' parent on this platform, and needs no device itself)."""
    import socket
    import subprocess

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "fetal_t2mapping_amd.cli",
           *(list(argv) if argv is not None else sys.argv[1:])]
    raise SystemExit(subprocess.call(cmd))


def main(argv=None):
    """run_t2mapping.py:522-576."""
    args = parse_arguments(argv)
    rank, world, local_rank = _dist_env()
    if args.gpus > 1 and world == 1:
        _relaunch_per_gpu(args, argv)
    if world > 1:  # a rank started by torch.distributed.run: its own GPU, one process group for the node
        import torch
        import torch.distributed as dist

        backend = os.environ.get("T2FIT_CLI_BACKEND", "nccl")  # "gloo": rehearsal of the control flow without RCCL
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC between the ranks' processes (before HIP starts)
        if backend == "nccl":
            args.device = local_rank
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not os.path.exists(args.path):
        print(f"Error: The specified path does not exist: {args.path}")
        raise SystemExit(1)
    bids_path = os.path.join(args.path, "projects/")
    csv_path = os.path.join(args.path, "dicom/logs/")
    low_field = bool(args.lf)
    TEs = args.TEs if args.TEs is not None else ([114, 202, 299] if args.lf else [115, 202, 299])
    phantom = bool(args.in_vitro or args.in_vitro_fast)
    fast = bool(args.in_vitro_fast)
    if args.norm:
        print("Warning: Fitting using normalization is not optimal !")
    fit, fit_params = t2map.set_fit_params(args)
    metadata = set_metadata(csv_path, args.csv, low_field)
    try:
        process_t2maps(metadata, bids_path, TEs, fit, fit_params, phantom, low_field, not args.no_prior, fast,
                       bool(args.norm), args.sim, solver=args.solver, precision=args.precision, device=args.device,
                       plots=args.plots, plot_seed=args.plot_seed, numpy_legacy=args.numpy_legacy)
    finally:
        if world > 1:
            import torch.distributed as dist

            if dist.is_initialized():
                dist.barrier()
                dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
