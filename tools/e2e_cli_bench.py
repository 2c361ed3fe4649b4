"""End-to-end wall time of the command line on one subject of BASELINE.json config 3 size (256x256x180 x 6 TE):
.nii.gz files on disk -> cli.main -> .nii.gz maps, with the package's own NIfTI reader/writer.  Prints one JSON line.

    python tools/e2e_cli_bench.py [solver] [fit] [Z Y X]
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.modules.setdefault("SimpleITK", None)  # absent in this image: use nifti.py

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

from fetal_t2mapping_amd import cli, nifti, synth  # noqa: E402

solver = sys.argv[1] if len(sys.argv) > 1 else "lbfgsb"
fit = sys.argv[2] if len(sys.argv) > 2 else "gaussian_rician"
shape = tuple(int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (180, 256, 256)
echoes, mask, te = synth.brain_volume(shape, 6, synth.SEED_BASE + 3)
te = np.round(te)
with tempfile.TemporaryDirectory() as root:
    bids = os.path.join(root, "projects") + "/"
    os.makedirs(os.path.join(bids, "prj-903"))
    os.makedirs(os.path.join(root, "dicom", "logs"))
    rows, items = [], []
    for i, t in enumerate(te):
        acq = {"prj": "prj-903", "sub": "sub-001", "ses": "ses-01", "run": f"run-{i + 1:02d}", "EchoTime": t / 1000.0,
               "CoilString": "HeadNeck"}
        rows.append(acq)
        items.append((nifti.GetImageFromArray(echoes[i]), cli.get_img_path(bids, acq, cli.recon_dirname).replace(" ", "")))
        items.append((nifti.GetImageFromArray(mask), cli.get_img_path(bids, acq, cli.mask_dirname).replace(" ", "")))
    nifti.WriteImages(items, threads=4)
    pd.DataFrame(rows).to_csv(os.path.join(root, "dicom", "logs", "log.csv"), index=False)
    in_bytes = sum(os.path.getsize(p) for _, p in items)
    argv = ["--path", root, "--csv", "log.csv", "--in_vivo", "--" + fit, "--lf", "--sim", "e2e", "--solver", solver,
            "--TEs"] + [str(int(t)) for t in te]
    stamps = {}
    real_read, real_fit, real_save = cli._read_subject, cli._fit_subject, cli.save_nifti_maps

    def timed(name, fn):
        def wrap(*a, **k):
            t0 = time.perf_counter()
            out = fn(*a, **k)
            stamps[name] = stamps.get(name, 0.0) + time.perf_counter() - t0
            return out
        return wrap

    cli._read_subject, cli._fit_subject, cli.save_nifti_maps = (timed("read_s", real_read), timed("fit_s", real_fit),
                                                                timed("write_s", real_save))
    for rep in range(2):  # the first pass pays torch / HIP start-up
        stamps.clear()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(argv)
        total = time.perf_counter() - t0
    out_dir = os.path.join(bids, "prj-903", "derivatives", cli.t2map_dirname, "sub-001", "ses-01", "anat")
    out_bytes = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir))
n = int(np.prod(shape))
print(json.dumps({"workload": f"{shape[2]}x{shape[1]}x{shape[0]} x 6 TE, one subject, .nii.gz in -> cli.main -> .nii.gz out",
                  "solver": solver, "fit": fit, "total_s": round(total, 3), **{k: round(v, 3) for k, v in stamps.items()},
                  "Mvoxel_s_end_to_end": round(n / total / 1e6, 1), "input_MB_on_disk": round(in_bytes / 1e6, 1),
                  "output_MB_on_disk": round(out_bytes / 1e6, 1), "host_cores": len(os.sched_getaffinity(0))}))
