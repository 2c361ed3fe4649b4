"""Convergence-study figures of the reference driver.

The reference draws three PNGs per (sub, ses) after the fit (run_t2mapping.py:465-468 ->
utils/t2map_utils.py:115-292): objective value per iteration for 50 random voxels, step length per
iteration for 20 random voxels (log axis), and iterations-vs-final-objective for every fitted voxel,
all coloured by the voxel's T2 on the ``jet`` map, into ``<bids>/<prj>/ada/convergence_analysis``.
There the per-iteration history is a Python list of dicts kept for EVERY voxel; here the fit kernel
returns ``nit`` / ``fun`` maps for all voxels and the history of the sampled voxels only is captured by
re-running those 70 voxels through the trace entry point (``t2fit_voxels_trace_host``; the fit is
deterministic, so their traces are the ones the full run took).  Same file names, same content.

The reference samples with the unseeded global ``random`` module; ``seed`` makes the choice repeatable.
"""
from __future__ import annotations

import os
import random
from typing import Optional, Sequence

import numpy as np

N_CURVES_OBJECTIVE = 50  # utils/t2map_utils.py:126 (the file name still says 20)
N_CURVES_STEP = 20       # utils/t2map_utils.py:213


def set_ada_path(bids_path: str, prj: str) -> str:
    """run_t2mapping.py:113-117 (creates missing parents too)."""
    path = os.path.join(bids_path, prj, "ada/convergence_analysis")
    os.makedirs(path, exist_ok=True)
    return path


def sample_voxels(n_masked: int, count: int, rng: random.Random) -> list:
    """``random.sample(range(n), count)`` as in the reference, which needs ``n >= count``; smaller masks
    (the reference raises ValueError there) are drawn in full."""
    return rng.sample(range(n_masked), min(count, n_masked))


def _curves(path, title, ylabel, curves: Sequence[Sequence[float]], t2_values: Sequence[float], log_y: bool):
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from matplotlib import cm

    fig, ax = plt.subplots(figsize=(12, 6))
    lo, hi = (min(t2_values), max(t2_values)) if len(t2_values) else (0.0, 1.0)
    norm = plt.Normalize(vmin=lo, vmax=hi)
    for ys, t2v in zip(curves, t2_values):
        ax.plot(range(len(ys)), ys, color=cm.jet(norm(t2v)))
    bar = fig.colorbar(cm.ScalarMappable(cmap=cm.jet, norm=norm), ax=ax)
    bar.set_label("T2 Value")
    ax.set_xlabel("Iteration")
    ax.set_ylabel(ylabel)
    ax.set_title(title)
    ax.grid(True)
    if log_y:
        ax.set_yscale("log")
    fig.tight_layout()
    fig.savefig(path)
    plt.close(fig)


def _scatter(path, nit, fun, t2_values):
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from matplotlib import cm

    fig, ax = plt.subplots(figsize=(10, 8))
    norm = plt.Normalize(vmin=float(np.min(t2_values)), vmax=float(np.max(t2_values))) if len(t2_values) else plt.Normalize(0, 1)
    ax.scatter(nit, fun, c=t2_values, cmap=cm.jet, norm=norm)
    bar = fig.colorbar(cm.ScalarMappable(cmap=cm.jet, norm=norm), ax=ax)
    bar.set_label("T2 Value")
    ax.set_xlabel("Number of Iterations")
    ax.set_ylabel("Final Loss Function Value")
    ax.set_title("Final Number of Iterations vs Final Loss Value (Colored by T2 Value)")
    ax.grid(True)
    fig.tight_layout()
    fig.savefig(path)
    plt.close(fig)


def pick_voxels(n_masked: int, seed: Optional[int] = None):
    """The two samples of the figures, as positions in ``mask_indices``: (50 for the objective curves, 20 for the step
    lengths), drawn like the reference draws them (two ``random.sample`` calls)."""
    rng = random.Random(seed)
    return sample_voxels(n_masked, N_CURVES_OBJECTIVE, rng), sample_voxels(n_masked, N_CURVES_STEP, rng)


def convergence_study(ada_path: str, echo_vols: Optional[Sequence[np.ndarray]], mask_indices: np.ndarray, t2_map: np.ndarray,
                      nit_map: np.ndarray, fun_map: np.ndarray, TEeffs, fit: str, fit_params: dict, prior: bool, norm: bool,
                      sub: str, ses: str, sim, *, solver: str = "lbfgsb", precision: str = "f64", device: int = 0,
                      seed: Optional[int] = None, trace_cap: int = 256, picks=None, rows: Optional[np.ndarray] = None,
                      numpy_legacy: bool = False) -> list:
    """Write the three figures; returns their paths.

    ``echo_vols``: the nTE volumes (only the sampled voxels' rows are gathered, the (N, nTE) stack of
    run_t2mapping.py:411 is not built); ``mask_indices``: flat indices of the fitted voxels;
    ``t2_map`` / ``nit_map`` / ``fun_map``: maps of the finished fit (N elements each).  A caller that does not hold
    every echo (one volume shared by several ranks) draws the sample itself (``pick_voxels``) and hands in ``picks``
    and the sampled voxels' ``rows`` (len(picks[0]) + len(picks[1]), nTE) instead of ``echo_vols``.
    """
    from . import t2map

    mask_indices = np.asarray(mask_indices, np.int64)
    t2_map, nit_map, fun_map = (np.asarray(a).reshape(-1) for a in (t2_map, nit_map, fun_map))
    pick_f, pick_s = picks if picks is not None else pick_voxels(len(mask_indices), seed)
    picks = list(pick_f) + list(pick_s)
    infos = []
    if picks:
        sel = mask_indices[picks]
        if rows is None:
            rows = np.stack([np.asarray(v).reshape(-1)[sel] for v in echo_vols], axis=1).astype(np.float32)
        rows = np.ascontiguousarray(rows, np.float32)
        infos = t2map.fit_voxels_trace(np.arange(len(sel)), fit, fit_params, TEeffs, rows, prior, norm,
                                       trace_cap=trace_cap, solver=solver, precision=precision, device=device,
                                       numpy_legacy=numpy_legacy)[5]
    t2_of = lambda sel: [float(t2_map[mask_indices[i]]) for i in sel]  # noqa: E731
    out = [os.path.join(ada_path, f"convergence_20_random_voxels_colored_by_t2_{sub}_{ses}_sim-{sim}_{fit}.png"),
           os.path.join(ada_path, f"step_size_convergence_20_random_voxels_colored_by_t2_{sub}_{ses}_sim-{sim}.png"),
           os.path.join(ada_path, f"scatter_iterations_vs_loss_colored_by_t2_{sub}_{ses}_sim-{sim}.png")]
    _curves(out[0], "Convergence of 20 Random Voxels Colored by T2 Value", "Objective Function Value (Loss)",
            [[e["f_val"] for e in info] for info in infos[: len(pick_f)]], t2_of(pick_f), log_y=False)
    _curves(out[1], "Step Size Convergence of 20 Random Voxels Colored by T2 Value", "Step Size",
            [[e["step_size"] for e in info] for info in infos[len(pick_f):]], t2_of(pick_s), log_y=True)
    _scatter(out[2], nit_map[mask_indices], fun_map[mask_indices], t2_map[mask_indices])
    print(f"Convergence figures saved to {ada_path}")
    return out
