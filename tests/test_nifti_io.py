"""fetal_t2mapping_amd/nifti.py (the file edge of the driver): header parsing against
hand-packed NIfTI-1 headers (written here with struct, independently of the module's writer), write/read
round trips, concurrent stack decode into a caller's buffer, and the CLI driver end to end on .nii.gz
files with the fit replaced by the oracle (tests/test_gpu_parity.py runs the same on the device)."""
import gzip
import os
import struct
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN
from fetal_t2mapping_amd import nifti


def _raw_header(shape_xyz, code, bitpix, endian="<", pixdim=(1, 1, 1, 1), qform=None, sform=None, slope=0.0, inter=0.0,
                vox_offset=352.0):
    h = bytearray(352)
    struct.pack_into(endian + "i", h, 0, 348)
    struct.pack_into(endian + "8h", h, 40, 3, *shape_xyz, 1, 1, 1, 1)
    struct.pack_into(endian + "hh", h, 70, code, bitpix)
    struct.pack_into(endian + "8f", h, 76, *pixdim, 0, 0, 0, 0)
    struct.pack_into(endian + "3f", h, 108, vox_offset, slope, inter)
    if qform is not None:
        struct.pack_into(endian + "h", h, 252, 1)
        struct.pack_into(endian + "6f", h, 256, *qform)
    if sform is not None:
        struct.pack_into(endian + "h", h, 254, 2)
        struct.pack_into(endian + "12f", h, 280, *np.asarray(sform, np.float32).reshape(-1))
    h[344:348] = b"n+1\0"
    return bytes(h)


def test_parse_hand_packed_headers(tmp_path):
    rng = np.random.default_rng(0)
    # int16, big-endian, qform only: 90 degrees about z, qfac -1, offsets, scl_slope 0.5 / inter 10
    data = rng.integers(-500, 500, size=(3, 4, 5)).astype(">i2")  # (Z, Y, X)
    s = np.sin(np.pi / 4)
    hdr = _raw_header((5, 4, 3), 4, 16, ">", pixdim=(-1.0, 0.8, 0.9, 2.5), qform=(0.0, 0.0, s, 11.0, -12.0, 13.0),
                      slope=0.5, inter=10.0)
    p = str(tmp_path / "be.nii")
    open(p, "wb").write(hdr + data.tobytes())
    img = nifti.ReadImage(p)
    assert img.arr.shape == (3, 4, 5) and img.arr.dtype == np.float32
    assert np.array_equal(img.arr, (data.astype(np.float64) * 0.5 + 10.0).astype(np.float32))
    assert np.allclose(img.GetSpacing(), (0.8, 0.9, 2.5))
    assert np.allclose(img.GetOrigin(), (-11.0, 12.0, 13.0))  # RAS -> LPS
    # RAS rotation [[0,-1,0],[1,0,0],[0,0,-1 (qfac)]] -> LPS: first two rows negated
    assert np.allclose(np.array(img.GetDirection()).reshape(3, 3), [[0, 1, 0], [-1, 0, 0], [0, 0, -1]], atol=1e-6)
    # float32, gzip, sform wins over qform; oblique affine
    vol = rng.normal(size=(2, 3, 4)).astype("<f4")
    A = np.array([[0.0, -1.2, 0.1, 5.0], [1.1, 0.0, 0.0, 6.0], [0.0, 0.2, 2.0, 7.0]])
    hdr = _raw_header((4, 3, 2), 16, 32, "<", qform=(0, 0, 0, 0, 0, 0), sform=A)
    p = str(tmp_path / "s.nii.gz")
    with gzip.open(p, "wb") as f:
        f.write(hdr + vol.tobytes())
    img = nifti.ReadImage(p)
    assert np.array_equal(img.arr, vol) and img.arr.dtype == np.float32
    L = np.diag([-1.0, -1.0, 1.0]) @ A[:, :3]
    sp = np.linalg.norm(L, axis=0)
    assert np.allclose(img.GetSpacing(), sp, rtol=1e-6)
    assert np.allclose(np.array(img.GetDirection()).reshape(3, 3), L / sp, atol=1e-6)
    assert np.allclose(img.GetOrigin(), (-5.0, -6.0, 7.0))
    # errors
    open(str(tmp_path / "bad.nii"), "wb").write(b"\0" * 400)
    with pytest.raises(ValueError):
        nifti.ReadImage(str(tmp_path / "bad.nii"))
    open(str(tmp_path / "short.nii"), "wb").write(_raw_header((4, 3, 2), 16, 32) + b"\0" * 10)
    with pytest.raises(ValueError):
        nifti.ReadImage(str(tmp_path / "short.nii"))


@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_write_read_round_trip(tmp_path, ext):
    rng = np.random.default_rng(1)
    # oblique, with a reflected axis (left-handed direction matrix exercises qfac)
    th = 0.3
    D = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, -1.0]])
    for dtype in (np.float32, np.uint8, np.int16, np.float64, np.uint16):
        arr = (rng.normal(size=(5, 6, 7)) * 100).astype(dtype)
        img = nifti.GetImageFromArray(arr)
        img.SetSpacing((0.5, 0.75, 3.0)); img.SetOrigin((-10.5, 20.25, 3.0)); img.SetDirection(D.reshape(-1))
        p = str(tmp_path / f"rt_{np.dtype(dtype).name}{ext}")
        nifti.WriteImage(img, p)
        back = nifti.ReadImage(p)
        assert back.arr.dtype == dtype and np.array_equal(back.arr, arr)
        assert np.allclose(back.GetSpacing(), img.GetSpacing(), rtol=1e-6)
        assert np.allclose(back.GetOrigin(), img.GetOrigin(), rtol=1e-6)
        assert np.allclose(back.GetDirection(), img.GetDirection(), atol=1e-6)
        assert back.GetSize() == (7, 6, 5)
    # the qform written agrees with the sform: read it back with the sform code cleared
    raw = bytearray(gzip.open(p).read() if ext.endswith(".gz") else open(p, "rb").read())
    struct.pack_into("<h", raw, 254, 0)
    q = str(tmp_path / "qonly.nii")
    open(q, "wb").write(raw)
    back = nifti.ReadImage(q)
    assert np.allclose(back.GetDirection(), D.reshape(-1), atol=1e-5) and np.allclose(back.GetOrigin(), (-10.5, 20.25, 3.0))
    # bool maps are written as uint8
    nifti.WriteImage(nifti.GetImageFromArray(arr > 0), str(tmp_path / ("b" + ext)))
    assert nifti.ReadImage(str(tmp_path / ("b" + ext))).arr.dtype == np.uint8


def test_read_stack_concurrent_into_buffer(tmp_path):
    rng = np.random.default_rng(2)
    vols = [(rng.random((9, 40, 50)) * 3000).astype(np.float32) for _ in range(6)]
    paths = []
    for i, v in enumerate(vols):
        src = v if i != 2 else v.astype(np.int16)  # mixed file dtypes are converted on the fly
        vols[i] = src.astype(np.float32)
        img = nifti.GetImageFromArray(src)
        img.SetSpacing((1, 1, 1.5 + i))
        paths.append(str(tmp_path / f"te{i}.nii.gz"))
        nifti.WriteImage(img, paths[-1])
    buf = np.full(6 * 9 * 40 * 50 + 0, -1, np.float32)
    stack, images = nifti.read_stack(paths, out=buf, threads=4)
    assert stack.shape == (6, 9, 40, 50) and np.shares_memory(stack, buf)
    for i in range(6):
        assert np.array_equal(stack[i], vols[i]) and np.shares_memory(images[i].arr, buf)
        assert images[i].GetSpacing()[2] == 1.5 + i
    stack2, _ = nifti.read_stack(paths, threads=1)
    assert np.array_equal(stack, stack2)
    with pytest.raises(ValueError):
        nifti.read_stack(paths, out=np.empty(10, np.float32))
    nifti.WriteImage(nifti.GetImageFromArray(np.zeros((2, 2, 2), np.float32)), str(tmp_path / "small.nii"))
    with pytest.raises(ValueError):
        nifti.read_stack(paths[:1] + [str(tmp_path / "small.nii")])


def test_cli_end_to_end_on_nifti_files(tmp_path, monkeypatch):
    """process_t2maps on real .nii.gz inputs through the native reader/writer (no SimpleITK in the image):
    same output names, arrays and geometry as the reference run recorded in the golden volume fixture."""
    from fetal_t2mapping_amd import cli as R
    from test_cli_driver import _oracle_fit_subject

    monkeypatch.setitem(sys.modules, "SimpleITK", None)  # make `import SimpleITK` fail, whatever other tests installed
    d = np.load(os.path.join(GOLDEN, "volume_lf_gaussian_noprior.npz"))
    bids = str(tmp_path / "projects") + "/"
    os.makedirs(os.path.join(bids, "prj-900"))
    rows = []
    for i, t in enumerate(d["te"]):
        acq = {"prj": "prj-900", "sub": "sub-001", "ses": "ses-01", "run": f"run-{i + 1:02d}", "EchoTime": t / 1000.0,
               "CoilString": "HeadNeck"}
        rows.append(acq)
        for arr, dirname in ((d["echoes"][i], R.recon_dirname), (d["masks"][i], R.mask_dirname)):
            img = nifti.GetImageFromArray(arr)
            img.SetSpacing(tuple(d["spacing"])); img.SetOrigin(tuple(d["origin"]))
            nifti.WriteImage(img, R.get_img_path(bids, acq, dirname).replace(" ", ""))
    monkeypatch.setattr(R, "_fit_subject", _oracle_fit_subject)
    R.process_t2maps(pd.DataFrame(rows), bids, [int(t) for t in d["te"]], "gaussian", R.t2map.fit_table("gaussian", True),
                     False, True, False, False, False, "g1")
    out_dir = os.path.join(bids, "prj-900", "derivatives", R.t2map_dirname, "sub-001", "ses-01", "anat")
    names = sorted(os.listdir(out_dir))
    assert names == sorted(os.path.basename(str(s)) for s in d["written"])
    for name in names:
        key = name.split("_sim-g1_")[1].split("map_")[0]
        img = nifti.ReadImage(os.path.join(out_dir, name))
        assert img.arr.dtype == np.float32 and np.array_equal(img.arr, d[key]), key
        assert np.allclose(img.GetSpacing(), d["spacing"]) and np.allclose(img.GetOrigin(), d["origin"])


def test_threaded_gzip_members_round_trip(tmp_path):
    """Volumes above 8 MB are written as several gzip members (one per thread): still an ordinary .gz for Python's
    gzip module and for this reader, both as a single image and through read_stack."""
    rng = np.random.default_rng(4)
    vol = rng.normal(size=(40, 256, 256)).astype(np.float32)  # 10.5 MB -> 2 data members + the header member
    p = str(tmp_path / "big.nii.gz")
    nifti.WriteImage(nifti.GetImageFromArray(vol), p, threads=4)
    raw = gzip.open(p).read()
    assert len(raw) == 352 + vol.nbytes and np.array_equal(np.frombuffer(raw[352:], np.float32).reshape(vol.shape), vol)
    assert open(p, "rb").read().count(b"\x1f\x8b\x08") >= 3
    assert np.array_equal(nifti.ReadImage(p).arr, vol)
    nifti.WriteImages([(nifti.GetImageFromArray(vol[:20]), str(tmp_path / "a.nii.gz")),
                       (nifti.GetImageFromArray(vol[20:]), str(tmp_path / "b.nii.gz"))])
    stack, _ = nifti.read_stack([str(tmp_path / "a.nii.gz"), str(tmp_path / "b.nii.gz")])
    assert np.array_equal(stack.reshape(vol.shape), vol)
