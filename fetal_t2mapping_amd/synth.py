"""Seeded synthetic multi-echo data of the shapes BASELINE.json names.

There is no dataset on the path (no network, the reference ships none); every test, fixture and
bench run uses signals generated here: Rician-magnitude noise over ``k * exp(-TE / T2)``.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 20250103  # seed = 20250103 + config index


def te_vector(n_te: int, low_field: bool = True, integer: bool = False) -> np.ndarray:
    """Echo times in ms.  3 echoes: the reference defaults (run_t2mapping.py:540-545);
    otherwise evenly spaced over the acquired 114..299 ms range."""
    if n_te == 3:
        return np.array([114.0 if low_field else 115.0, 202.0, 299.0])
    te = np.linspace(114.0, 299.0, n_te)
    return np.round(te) if integer else te


def voxels(rng: np.random.Generator, te, m: int, k_range=(700.0, 3000.0), t2_range=(20.0, 1000.0),
           sigmas=(5.0, 20.0, 50.0)):
    """``(m, nTE)`` float32 rows + the (k, T2, sigma) they were drawn from."""
    te = np.asarray(te, np.float64)
    k = rng.uniform(*k_range, size=m)
    # log-uniform T2 so short and long relaxation times are both well represented
    t2 = np.exp(rng.uniform(np.log(t2_range[0]), np.log(t2_range[1]), size=m))
    sg = rng.choice(np.asarray(sigmas, np.float64), size=m)
    clean = k[:, None] * np.exp(-te[None, :] / t2[:, None])
    n1 = rng.normal(size=clean.shape) * sg[:, None]
    n2 = rng.normal(size=clean.shape) * sg[:, None]
    y = np.hypot(clean + n1, n2).astype(np.float32)
    return y, np.stack([k, t2, sg], axis=1)


def edge_rows(te, low_field: bool = True) -> tuple[np.ndarray, list[str]]:
    """Rows that exercise bounds, degenerate and non-finite inputs."""
    te = np.asarray(te, np.float64)
    n = te.size
    klb = 600.0 if low_field else 850.0
    rows = {
        "bright_decay": 16000.0 * np.exp(-te / 250.0),          # S(TE0) > 10000: no-prior lb > ub
        "zeros": np.zeros(n),
        "flat_100": np.full(n, 100.0),
        "rising": np.linspace(50.0, 120.0, n),
        "nan_first": np.r_[np.nan, np.linspace(10.0, 5.0, n - 1)],
        "inf_last": np.r_[np.linspace(900.0, 300.0, n - 1), np.inf],
        "t2_below_lb": 2000.0 * np.exp(-te / 5.0),
        "t2_above_ub": 1500.0 * np.exp(-te / 5000.0),
        "k_below_lb": 0.2 * klb * np.exp(-te / 120.0),
        "k_at_ub": 9000.0 * np.exp(-te / 80.0) * np.exp(te[0] / 80.0),
        "negative": np.linspace(-5.0, 3.0, n),
        "clean_wm": 1000.0 * np.exp(-te / 110.0),
        "clean_csf": 2500.0 * np.exp(-te / 1500.0),
        "noise_only": np.full(n, 25.0) + np.arange(n) % 2,
    }
    names = list(rows)
    return np.stack([rows[k] for k in names]).astype(np.float32), names


def brain_volume(shape, n_te: int, seed: int, low_field: bool = True, sigma: float = 20.0,
                 fill: float = 0.45):
    """Small numpy phantom: ellipsoidal 'brain' (T2 40..400 ms) with a CSF pocket (600..2000 ms)
    over a Rayleigh background.  Returns ``(echoes (nTE,Z,Y,X) f32, mask (Z,Y,X) u8, te f64)``."""
    rng = np.random.default_rng(seed)
    z, y, x = shape
    te = te_vector(n_te, low_field)
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, z), np.linspace(-1, 1, y), np.linspace(-1, 1, x),
                             indexing="ij")
    # semi-axes chosen so the ellipsoid fills `fill` of the box: (4/3)pi abc / 8 = fill
    a = (fill * 6.0 / np.pi) ** (1.0 / 3.0)
    r2 = (zz / a) ** 2 + (yy / a) ** 2 + (xx / a) ** 2
    mask = r2 <= 1.0
    csf = r2 <= 0.05
    k = rng.uniform(700.0, 3000.0, size=shape)
    t2 = rng.uniform(40.0, 400.0, size=shape)
    t2[csf] = rng.uniform(600.0, 2000.0, size=int(csf.sum()))
    clean = np.where(mask, k, 0.0)[None] * np.exp(-te[:, None, None, None] / t2[None])
    n1 = rng.normal(scale=sigma, size=clean.shape)
    n2 = rng.normal(scale=sigma, size=clean.shape)
    echoes = np.hypot(clean + n1, n2).astype(np.float32)
    return echoes, mask.astype(np.uint8), te


def brain_volume_torch(shape, n_te: int, seed: int, device, sigma: float = 20.0, fill: float = 0.45,
                       z_offset: int = 0, z_total: int | None = None):
    """Bench-size generator on the device (256^3 x 8 TE is too slow as a numpy pass).

    Same distribution as ``brain_volume``: ellipsoidal mask filling ``fill`` of the (z_total, Y, X)
    box, CSF pocket in the centre, Rician noise.  ``z_offset``/``z_total`` let one rank generate
    its Z-slab of a taller volume.  Returns ``(echoes (nTE, Z*Y*X) f32, mask (Z*Y*X,) u8, te f64
    ndarray)`` with the tensors on `device`.
    """
    import torch

    z, y, x = shape
    z_total = z if z_total is None else z_total
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    te = te_vector(n_te, True)
    n_vox = z * y * x
    a = (fill * 6.0 / np.pi) ** (1.0 / 3.0)
    lin = lambda n, off, tot: ((torch.arange(n, device=device, dtype=torch.float32) + off) * (2.0 / max(tot - 1, 1)) - 1.0) / a  # noqa: E731
    r2 = (lin(z, z_offset, z_total)[:, None, None] ** 2 + lin(y, 0, y)[None, :, None] ** 2
          + lin(x, 0, x)[None, None, :] ** 2).reshape(-1)
    mask = r2 <= 1.0
    csf = r2 <= 0.05
    k = torch.empty(n_vox, device=device).uniform_(700.0, 3000.0, generator=g)
    t2 = torch.empty(n_vox, device=device).uniform_(40.0, 400.0, generator=g)
    t2 = torch.where(csf, torch.empty(n_vox, device=device).uniform_(600.0, 2000.0, generator=g), t2)
    k = torch.where(mask, k, torch.zeros_like(k))
    echoes = torch.empty((n_te, n_vox), dtype=torch.float32, device=device)
    for i in range(n_te):
        clean = k * torch.exp(-float(te[i]) / t2)
        n1 = torch.randn(n_vox, generator=g, device=device) * sigma
        n2 = torch.randn(n_vox, generator=g, device=device) * sigma
        echoes[i] = torch.hypot(clean + n1, n2)
    return echoes, mask.to(torch.uint8), te


def phantom_volume(shape=(20, 64, 64), n_te: int = 6, seed: int = SEED_BASE, low_field: bool = False,
                   sigma: float = 15.0, k: float = 2000.0):
    """NIST-phantom-like volume (BASELINE.json config 1: 64x64x20 x 6 TE): one cylinder per vial with
    the NMR ground-truth T2 of the reference's table (run_t2mapping.py:19,24), Rician noise.
    Returns ``(echoes (nTE,Z,Y,X) f32, mask (Z,Y,X) u8, label (Z,Y,X) i16, te, t2_truth list)``."""
    gt = [594, 416, 284, 221, 167, 122, 80, 53, 41] if low_field else [1044, 624, 428, 258, 186, 137, 90, 63, 44, 27, 19,
                                                                        15, 10, 8]
    rng = np.random.default_rng(seed)
    z, y, x = shape
    te = te_vector(n_te, low_field)
    label = np.zeros(shape, np.int16)
    yy, xx = np.meshgrid(np.arange(y), np.arange(x), indexing="ij")
    n = len(gt)
    for i in range(n):
        ang = 2 * np.pi * i / n
        cy, cx = y / 2 + 0.36 * y * np.sin(ang), x / 2 + 0.36 * x * np.cos(ang)
        disc = (yy - cy) ** 2 + (xx - cx) ** 2 <= (0.055 * min(y, x)) ** 2
        label[z // 4: z - z // 4, disc] = i + 1
    body = (yy - y / 2) ** 2 + (xx - x / 2) ** 2 <= (0.47 * min(y, x)) ** 2
    mask = np.broadcast_to(body, shape).astype(np.uint8).copy()
    t2 = np.full(shape, 2200.0)  # water-like fill between the vials
    for i, v in enumerate(gt):
        t2[label == i + 1] = v
    clean = np.where(mask != 0, k, 0.0)[None] * np.exp(-te[:, None, None, None] / t2[None])
    n1 = rng.normal(scale=sigma, size=clean.shape)
    n2 = rng.normal(scale=sigma, size=clean.shape)
    return np.hypot(clean + n1, n2).astype(np.float32), mask, label, te, gt
