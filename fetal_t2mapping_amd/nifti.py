"""NIfTI-1 single-file I/O for the file edge of the T2-mapping driver.

The reference reads its per-TE reconstructions and masks and writes its four maps with SimpleITK
(run_t2mapping.py:374-377, utils/t2map_utils.py:18-29).  ``cli.py`` does the same when SimpleITK is
importable; this module is what it uses otherwise, and what feeds the streaming path: it exposes the
handful of SimpleITK calls the driver makes (``ReadImage``, ``GetArrayFromImage``,
``GetImageFromArray``, ``WriteImage``, ``Image.Get/Set{Spacing,Origin,Direction}``) on plain numpy,
plus ``read_stack``: all echoes of a subject decoded concurrently (zlib releases the GIL) straight
into one float32 ``(nTE, Z, Y, X)`` buffer -- which may be the numpy view of a pinned staging tensor,
so that gzip decode -> pinned memory -> HBM involves no intermediate copy and no ``(Z,Y,X,nTE)``
transpose (run_t2mapping.py:385-386 is never materialised).

Geometry follows ITK's conventions so that maps written here carry the geometry SimpleITK would have
written: arrays are ``(Z, Y, X)`` (x fastest on disk); spacing/origin/direction are in ITK's LPS frame
(the NIfTI RAS affine with its first two rows negated); the sform is used when ``sform_code > 0``,
otherwise the qform quaternion, otherwise ``pixdim`` alone; both forms are written.
Scope: NIfTI-1 ``.nii`` / ``.nii.gz``, 3-D (or 4-D with one volume), little- or big-endian, the scalar
datatypes 2/4/8/16/64/256/512/768, ``scl_slope``/``scl_inter`` applied on read.
"""
from __future__ import annotations

import struct
import zlib
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Sequence

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}
_HDR = 348
_CHUNK = 1 << 22


class Image:
    """Array plus ITK-style geometry (the subset of ``SimpleITK.Image`` the driver touches)."""

    def __init__(self, arr: np.ndarray, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
                 direction=(1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)):
        self.arr = arr
        self._spacing, self._origin, self._direction = tuple(map(float, spacing)), tuple(map(float, origin)), tuple(map(float, direction))

    def GetSpacing(self): return self._spacing
    def GetOrigin(self): return self._origin
    def GetDirection(self): return self._direction
    def GetSize(self): return tuple(int(s) for s in self.arr.shape[::-1])
    def SetSpacing(self, s): self._spacing = tuple(map(float, s))
    def SetOrigin(self, o): self._origin = tuple(map(float, o))
    def SetDirection(self, d): self._direction = tuple(map(float, d))


# ---- header --------------------------------------------------------------------------------------
class _Header:
    __slots__ = ("endian", "shape", "dtype", "vox_offset", "slope", "inter", "spacing", "origin", "direction")


def _quaternion_to_matrix(b, c, d):
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 1e-7 else 0.0
    if a2 <= 1e-7:  # special case of the NIfTI-1 standard: renormalise (b, c, d)
        s = 1.0 / np.sqrt(b * b + c * c + d * d)
        b, c, d = b * s, c * s, d * s
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])


def _matrix_to_quaternion(R):
    """Unit quaternion (b, c, d) with a >= 0 of a proper rotation matrix (nifti1_io's mat44_to_quatern)."""
    r11, r12, r13 = R[0]
    r21, r22, r23 = R[1]
    r31, r32, r33 = R[2]
    a = r11 + r22 + r33 + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        return 0.25 * (r32 - r23) / a, 0.25 * (r13 - r31) / a, 0.25 * (r21 - r12) / a
    xd, yd, zd = 1.0 + r11 - (r22 + r33), 1.0 + r22 - (r11 + r33), 1.0 + r33 - (r11 + r22)
    if xd > 1.0:
        b = 0.5 * np.sqrt(xd)
        c, d, a = 0.25 * (r12 + r21) / b, 0.25 * (r13 + r31) / b, 0.25 * (r32 - r23) / b
    elif yd > 1.0:
        c = 0.5 * np.sqrt(yd)
        b, d, a = 0.25 * (r12 + r21) / c, 0.25 * (r23 + r32) / c, 0.25 * (r13 - r31) / c
    else:
        d = 0.5 * np.sqrt(zd)
        b, c, a = 0.25 * (r13 + r31) / d, 0.25 * (r23 + r32) / d, 0.25 * (r21 - r12) / d
    if a < 0.0:
        b, c, d = -b, -c, -d
    return b, c, d


def _parse_header(raw: bytes) -> _Header:
    if len(raw) < _HDR:
        raise ValueError("not a NIfTI-1 file: header shorter than 348 bytes")
    for endian in ("<", ">"):
        if struct.unpack_from(endian + "i", raw, 0)[0] == _HDR:
            break
    else:
        raise ValueError("not a NIfTI-1 file: sizeof_hdr != 348")
    if raw[344:348] not in (b"n+1\0", b"ni1\0"):
        raise ValueError("not a NIfTI-1 file: bad magic")
    if raw[344:348] == b"ni1\0":
        raise ValueError("two-file NIfTI (.hdr/.img) is not supported; use single-file .nii[.gz]")
    dim = struct.unpack_from(endian + "8h", raw, 40)
    datatype, bitpix = struct.unpack_from(endian + "hh", raw, 70)
    pixdim = struct.unpack_from(endian + "8f", raw, 76)
    vox_offset, slope, inter = struct.unpack_from(endian + "3f", raw, 108)
    qform_code, sform_code = struct.unpack_from(endian + "hh", raw, 252)
    qb, qc, qd, qx, qy, qz = struct.unpack_from(endian + "6f", raw, 256)
    srow = np.array(struct.unpack_from(endian + "12f", raw, 280), np.float64).reshape(3, 4)
    nd = dim[0]
    if not 1 <= nd <= 7:
        raise ValueError("bad dim[0] in NIfTI header")
    shape = [int(d) for d in dim[1:1 + nd]]
    while len(shape) > 3 and shape[-1] == 1:
        shape.pop()
    if len(shape) > 3:
        raise ValueError(f"expected a 3-D volume, file has dimensions {shape}")
    while len(shape) < 3:
        shape.append(1)
    if datatype not in _DTYPES:
        raise ValueError(f"unsupported NIfTI datatype code {datatype}")
    h = _Header()
    h.endian = endian
    h.shape = tuple(shape[::-1])  # (Z, Y, X)
    h.dtype = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
    h.vox_offset = int(vox_offset) if vox_offset >= _HDR else 352
    h.slope, h.inter = (float(slope), float(inter)) if slope not in (0.0,) and np.isfinite(slope) else (1.0, 0.0)
    # RAS affine: sform if present, else qform, else pixdim on the diagonal
    if sform_code > 0:
        A = srow[:, :3].copy()
        t = srow[:, 3].copy()
    elif qform_code > 0:
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        R = _quaternion_to_matrix(qb, qc, qd)
        A = R * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac], np.float64)[None, :]
        t = np.array([qx, qy, qz], np.float64)
    else:
        A = np.diag([abs(pixdim[1]) or 1.0, abs(pixdim[2]) or 1.0, abs(pixdim[3]) or 1.0]).astype(np.float64)
        t = np.zeros(3)
    lps = np.diag([-1.0, -1.0, 1.0])
    A, t = lps @ A, lps @ t
    sp = np.linalg.norm(A, axis=0)
    sp[sp == 0] = 1.0
    h.spacing = tuple(float(s) for s in sp)
    h.origin = tuple(float(v) for v in t)
    h.direction = tuple(float(v) for v in (A / sp[None, :]).reshape(-1))
    return h


def _build_header(img: Image, dtype: np.dtype) -> bytes:
    z, y, x = img.arr.shape
    sp = np.array(img.GetSpacing(), np.float64)
    D = np.array(img.GetDirection(), np.float64).reshape(3, 3)
    lps = np.diag([-1.0, -1.0, 1.0])
    A = lps @ (D * sp[None, :])
    t = lps @ np.array(img.GetOrigin(), np.float64)
    R = A / sp[None, :]
    qfac = 1.0
    if np.linalg.det(R) < 0:  # improper: flip the third column, recorded in pixdim[0]
        R = R * np.array([1.0, 1.0, -1.0])[None, :]
        qfac = -1.0
    qb, qc, qd = _matrix_to_quaternion(R)
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, _HDR)
    struct.pack_into("<8h", hdr, 40, 3, x, y, z, 1, 1, 1, 1)
    struct.pack_into("<hh", hdr, 70, _CODES[dtype.str[1:]], dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, qfac, sp[0], sp[1], sp[2], 0.0, 0.0, 0.0, 0.0)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2  # xyzt_units: millimetres
    struct.pack_into("<hh", hdr, 252, 1, 1)  # qform_code, sform_code: scanner anatomical
    struct.pack_into("<6f", hdr, 256, qb, qc, qd, t[0], t[1], t[2])
    struct.pack_into("<12f", hdr, 280, *np.concatenate([A, t[:, None]], axis=1).reshape(-1))
    hdr[344:348] = b"n+1\0"
    return bytes(hdr)


# ---- reading -------------------------------------------------------------------------------------
def _is_gz(path: str) -> bool:
    with open(path, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def _decode_into(path: str, out: Optional[np.ndarray], want_dtype=None):
    """Decode `path`; voxel values go to `out` (flat view, converted to its dtype with the header's
    scaling applied) or to a fresh array of the file's dtype.  Streaming: the compressed file is never
    held decompressed as a whole next to the destination."""
    gz = _is_gz(path)
    with open(path, "rb") as f:
        if gz:
            dec = zlib.decompressobj(wbits=31)
            pending = bytearray()

            def pull(nbytes):  # exactly nbytes of decompressed data (or fewer at EOF)
                nonlocal dec
                while len(pending) < nbytes:
                    chunk = f.read(_CHUNK)
                    if not chunk:
                        break
                    data = dec.decompress(chunk)
                    pending.extend(data)
                    while dec.eof and dec.unused_data:  # concatenated gzip members
                        rest = dec.unused_data
                        dec = zlib.decompressobj(wbits=31)
                        pending.extend(dec.decompress(rest))
                got = bytes(pending[:nbytes])
                del pending[:nbytes]
                return got
        else:
            def pull(nbytes):
                return f.read(nbytes)

        head = pull(352)
        h = _parse_header(head)
        skip = h.vox_offset - len(head)
        if skip > 0:
            pull(skip)
        n = int(np.prod(h.shape))
        scaled = not (h.slope == 1.0 and h.inter == 0.0)
        if out is None:
            dst_dtype = np.dtype(want_dtype) if want_dtype is not None else (np.dtype(np.float32) if scaled and h.dtype.kind != "f" else h.dtype.newbyteorder("="))
            out = np.empty(n, dst_dtype)
        flat = out.reshape(-1)
        if flat.size != n:
            raise ValueError(f"{path}: volume has {n} voxels, destination has {flat.size}")
        step = max(1, _CHUNK // h.dtype.itemsize)
        done = 0
        while done < n:
            cnt = min(step, n - done)
            raw = pull(cnt * h.dtype.itemsize)
            if len(raw) != cnt * h.dtype.itemsize:
                raise ValueError(f"{path}: file ends after {done} of {n} voxels")
            src = np.frombuffer(raw, h.dtype, cnt)
            if scaled:
                flat[done:done + cnt] = src.astype(np.float64) * h.slope + h.inter
            else:
                flat[done:done + cnt] = src  # converts dtype / byte order in the same pass
            done += cnt
    return h, out


def ReadImage(path: str) -> Image:
    h, data = _decode_into(path, None)
    return Image(data.reshape(h.shape), h.spacing, h.origin, h.direction)


def ReadGeometry(path: str) -> Image:
    """Spacing / origin / direction of the volume at `path` without decoding it (the first 352 bytes): an
    :class:`Image` over an empty array.  (The maps carry the geometry of the last echo's image,
    run_t2mapping.py:377 / utils/t2map_utils.py:22-24; a rank that writes them need not have decoded that echo.)"""
    with open(path, "rb") as f:
        head = f.read(4096)
    if head[:2] == b"\x1f\x8b":
        head = zlib.decompressobj(wbits=31).decompress(head, 352)
    h = _parse_header(head)
    return Image(np.zeros((0, 0, 0), np.float32), h.spacing, h.origin, h.direction)


def GetArrayFromImage(img: Image) -> np.ndarray:
    return img.arr


def GetImageFromArray(arr) -> Image:
    arr = np.asarray(arr)
    if arr.ndim != 3:
        raise ValueError("expected a (Z, Y, X) array")
    return Image(arr)


def _gzip_member(buf, level: int) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    return c.compress(buf) + c.flush()


def WriteImage(img: Image, path: str, compresslevel: int = 1, threads: int = 4) -> None:
    """Write ``.nii`` or ``.nii.gz`` (by extension).  gzip level 1: the maps are float32 noise-like
    data on which higher levels cost several times the time for a few per cent of size.  The voxel
    data is compressed as several gzip members by a few threads (zlib releases the GIL; a file of
    concatenated members is an ordinary ``.gz`` to zlib's ``gzread``, Python's ``gzip`` and this reader)."""
    arr = np.ascontiguousarray(img.arr)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype.str[1:] not in _CODES:
        arr = arr.astype(np.float32)
    arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
    hdr = _build_header(Image(arr, img.GetSpacing(), img.GetOrigin(), img.GetDirection()), arr.dtype)
    data = memoryview(arr).cast("B")
    if not path.endswith(".gz"):
        with open(path, "wb") as f:
            f.write(hdr)
            f.write(data)
        return
    n = len(data)
    parts = max(1, min(threads, n // (4 << 20)))
    step = -(-n // parts) if n else 0
    chunks = [data[i:i + step] for i in range(0, n, step)] if n else []
    if parts > 1:
        with ThreadPoolExecutor(parts) as pool:
            members = list(pool.map(lambda b: _gzip_member(b, compresslevel), chunks))
    else:
        members = [_gzip_member(b, compresslevel) for b in chunks]
    with open(path, "wb") as f:
        f.write(_gzip_member(hdr, compresslevel))
        for mbr in members:
            f.write(mbr)


def WriteImages(items, compresslevel: int = 1, threads: int = 4) -> None:
    """``WriteImage`` for several ``(image, path)`` pairs at once (the four maps of a subject)."""
    items = list(items)
    with ThreadPoolExecutor(max(1, min(threads, len(items)))) as pool:
        list(pool.map(lambda it: WriteImage(it[0], it[1], compresslevel, threads), items))


def read_stack(paths: Sequence[str], out: Optional[np.ndarray] = None, dtype=np.float32, threads: int = 8):
    """Decode the volumes at `paths` concurrently into one ``(len(paths), Z, Y, X)`` array of `dtype`.

    `out` may be a preallocated array (e.g. ``pinned_tensor.numpy()``) of exactly that shape, or a flat
    one of that many elements.  Returns ``(stack, images)`` where ``images[i]`` is an :class:`Image`
    viewing ``stack[i]`` with the geometry of file i.  All volumes must have the same shape.
    """
    if not paths:
        raise ValueError("no paths")
    with open(paths[0], "rb") as f:
        head = f.read(4096)
    if head[:2] == b"\x1f\x8b":
        head = zlib.decompressobj(wbits=31).decompress(head, 352)
    shape = _parse_header(head).shape
    n = int(np.prod(shape))
    if out is None:
        out = np.empty((len(paths),) + shape, dtype)
    if out.size != len(paths) * n:
        raise ValueError(f"destination has {out.size} elements, need {len(paths)} x {n}")
    stack = out.reshape((len(paths),) + shape)

    def one(i):
        h, _ = _decode_into(paths[i], stack[i])
        if h.shape != shape:
            raise ValueError(f"{paths[i]}: shape {h.shape} differs from {paths[0]}: {shape}")
        return Image(stack[i], h.spacing, h.origin, h.direction)

    if len(paths) == 1 or threads <= 1:
        images = [one(i) for i in range(len(paths))]
    else:
        with ThreadPoolExecutor(min(threads, len(paths))) as pool:
            images = list(pool.map(one, range(len(paths))))
    return stack, images
