"""CPU checks of the per-lane solver logic through the host-side lane simulator
(tests/hostsim: the same headers the HIP kernels are compiled from, built with g++).
The simulator is test infrastructure only -- the product package has no CPU path."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from hostsim import sim

FILES = sorted(glob.glob(os.path.join(GOLDEN, "voxels_*.npz")))
PICK = [f for f in FILES if os.path.basename(f)[7:-4] in (
    "lf_gaussian_prior_te3", "hf_gaussian_noprior_te6", "lf_gaussian_rician_prior_te8",
    "hf_gaussian_rician_noprior_te3", "lf_rician_prior_te6", "hf_rician_noprior_te8")]


@pytest.mark.parametrize("path", PICK, ids=lambda p: os.path.basename(p)[7:-4])
def test_lbfgsb_lane_solver_tracks_reference(path):
    """Every voxel of a fixture: at least as close to the reference as the worst of the reference's own 24 one-ulp
    perturbed runs, less one point (tests/golden/make_noise_floor.py), in T2 and in the iteration count."""
    d = np.load(path)
    floor = np.load(os.path.join(GOLDEN, "noise_floor.npz"))
    name = os.path.basename(path)[7:-4]
    cfg = sim.config(str(d["mode"]), bool(d["low_field"]), d["te"], prior=bool(d["prior"]), solver="lbfgsb")
    o = sim.fit_rows(cfg, d["y"])
    assert np.array_equal(o["status"] == 4, d["raised"])
    fit = ~d["raised"] & np.isfinite(d["fun"])
    dt = np.abs(o["x"][fit, 1] - d["x"][fit, 1])
    assert np.mean(dt <= 1.0) >= float(floor[name + "/frac_1ms_min"]) - 0.01
    assert np.mean(o["nit"][fit] == d["nit"][fit]) >= float(floor[name + "/nit_equal_min"]) - 0.01
    assert np.median(dt) <= 0.02
    assert np.mean((o["status"][fit] == 1) == d["success"][fit]) >= 0.995
    bad = ~d["raised"] & ~np.isfinite(d["fun"])
    n_par = d["x"].shape[1]
    assert np.allclose(o["x"][bad][:, :n_par], d["x"][bad]) and np.all(o["nit"][bad] == 0)


def test_lbfgsb_lane_solver_on_the_stable_sets():
    """Where the reference's answer does not depend on the last bit of exp() / log() / i0e() -- the stable set of each
    fixture: all 24 perturbed runs of the reference reproduce the golden row -- the lane solver must reproduce it
    too: over all 36 fixtures (3105 voxels) T2 within 1 ms on >= 99.9 %, `success` equal on all, `nit` equal on
    >= 99.5 %.  (The HIP path is held to the same bar on the GPU: test_gpu_parity.py::test_lbfgsb_stable_set.)"""
    floor = np.load(os.path.join(GOLDEN, "noise_floor.npz"))
    n = n_t2 = n_nit = n_ok = 0
    for path in FILES:
        d = np.load(path)
        name = os.path.basename(path)[7:-4]
        rows = np.flatnonzero(floor[name + "/stable"])
        cfg = sim.config(str(d["mode"]), bool(d["low_field"]), d["te"], prior=bool(d["prior"]), solver="lbfgsb")
        o = sim.fit_rows(cfg, d["y"][rows])
        n += len(rows)
        n_t2 += int(np.sum(np.abs(o["x"][:, 1] - d["x"][rows, 1]) > 1.0))
        n_nit += int(np.sum(o["nit"] != d["nit"][rows]))
        n_ok += int(np.sum((o["status"] == 1) != d["success"][rows]))
    assert n >= 3000
    assert n_ok == 0 and n_t2 <= 1e-3 * n and n_nit <= 5e-3 * n, (n, n_t2, n_nit, n_ok)


def test_lbfgsb_lane_solver_on_the_frozen_stack_stable_sets():
    """The reference under the stack it freezes (numpy 1.26 promotion rules, Fortran L-BFGS-B:
    tests/golden/make_golden_frozen.py).  Lane solver with cfg.numpy_legacy = 1 on the rows stable under both stacks'
    perturbations (rician: under the frozen stack's; its numpy-2 trajectory is a different one): per model T2 within
    1 ms on >= 99.9 %, `success` equal on all, `nit` equal on >= 99.5 %.  Residual map in its float32 form within
    1e-3 of the frozen stack's (glibc's expf against numpy's)."""
    floor = np.load(os.path.join(GOLDEN, "noise_floor.npz"))
    tally = {}
    for path in FILES:
        d = np.load(path)
        name = os.path.basename(path)[7:-4]
        fz = np.load(os.path.join(GOLDEN, f"frozen_voxels_{name}.npz"))
        mode = str(d["mode"])
        rows = np.flatnonzero(fz["stable"] & (floor[name + "/stable"] if mode != "rician" else True))
        cfg = sim.config(mode, bool(d["low_field"]), d["te"], prior=bool(d["prior"]), solver="lbfgsb", numpy_legacy=True)
        o = sim.fit_rows(cfg, d["y"][rows])
        t = tally.setdefault(mode, [0, 0, 0, 0])
        t[0] += len(rows)
        t[1] += int(np.sum(np.abs(o["x"][:, 1] - fz["x"][rows, 1]) > 1.0))
        t[2] += int(np.sum(o["nit"] != fz["nit"][rows]))
        t[3] += int(np.sum((o["status"] == 1) != fz["success"][rows]))
        okr = np.flatnonzero(~fz["raised"] & np.isfinite(fz["res"]))
        x = fz["x"][okr]
        res = sim.residuals(cfg, d["y"][okr], x[:, 0], x[:, 1], x[:, 2] if x.shape[1] == 3 else np.zeros(len(okr)))
        assert np.max(np.abs(res - fz["res"][okr])) <= 1e-3, name
    for mode, (n, n_t2, n_nit, n_ok) in tally.items():
        assert n >= 900, (mode, n)
        assert n_ok == 0 and n_t2 <= 1e-3 * n and n_nit <= 5e-3 * n, (mode, n, n_t2, n_nit, n_ok)


@pytest.mark.parametrize("path", [f for f in FILES if "gaussian_prior_te6" in f or "gaussian_rician_prior_te3" in f],
                         ids=lambda p: os.path.basename(p)[7:-4])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_lm_lane_solver_reaches_minimum(path, precision):
    d = np.load(path)
    mode = str(d["mode"])
    cfg = sim.config(mode, bool(d["low_field"]), d["te"], prior=True, solver="lm", precision=precision)
    o = sim.fit_rows(cfg, d["y"])
    good = ~d["raised"] & np.isfinite(d["f_tight"]) & np.isfinite(o["fun"])
    dt = np.abs(o["x"][good, 1] - d["x_tight"][good, 1])
    assert np.mean(dt <= 1.0) >= (0.99 if mode == "gaussian" else 0.97)


def test_residual_map_matches_reference():
    """compute_residuals restated per lane (float64 prediction stored as float32, float32 residuals,
    numpy's pairwise float32 row sum): evaluated on the reference's own parameters it must give the
    reference's float32 residuals bit-for-bit, except where exp() rounds differently (<= 1e-3)."""
    for path in FILES[::5]:
        d = np.load(path)
        cfg = sim.config(str(d["mode"]), bool(d["low_field"]), d["te"], prior=bool(d["prior"]), solver="lbfgsb")
        rows = np.where(~d["raised"] & np.isfinite(d["res"]))[0]
        x = d["x"][rows]
        sg = x[:, 2] if x.shape[1] == 3 else np.zeros(len(rows))
        res = sim.residuals(cfg, d["y"][rows], x[:, 0], x[:, 1], sg)
        assert np.mean(res == d["res"][rows]) >= 0.9, path
        assert np.max(np.abs(res - d["res"][rows])) <= 1e-3, path


@pytest.mark.parametrize("path", [f for f in FILES if "_gaussian_prior_" in f or "_gaussian_noprior_" in f],
                         ids=lambda p: os.path.basename(p)[7:-4])
def test_loglin_lane_equals_closed_form_oracle(path):
    """T2FIT_SOLVER_LOGLIN (extension named by BASELINE.json config 2; the reference has no log-linear
    routine, so parity with it is unpinned): the lane code must equal the closed form of
    oracle.loglinear_fit -- float64 sums over a float32 logarithm, so T2 within 1e-4 relative -- and
    recover a noise-free decay inside the bounds, where it also meets the reference's converged fit."""
    from oracle import t2fit_oracle as oracle

    d = np.load(path)
    prior = bool(d["prior"])
    table = oracle.fit_table("gaussian", bool(d["low_field"]))
    cfg = sim.config("gaussian", bool(d["low_field"]), d["te"], prior=prior, solver="loglin")
    o = sim.fit_rows(cfg, d["y"])
    rows = np.where(~d["raised"] & np.all(np.isfinite(d["y"]), axis=1))[0]
    want, ok = oracle.loglinear_fit(d["y"][rows], d["te"], table, prior=prior)
    got = o["x"][rows]
    assert np.array_equal(o["status"][rows] == 1, ok)
    assert np.all(o["nit"][rows] == 0) and np.all(got[:, 2] == 0)
    assert np.allclose(got[:, 1], want[:, 1], rtol=1e-4, atol=1e-6)
    assert np.allclose(got[:, 0], want[:, 0], rtol=1e-4, atol=1e-6)
    # rows scipy refuses (lb > ub) and non-finite rows behave as in the other solvers
    assert np.array_equal(o["status"] == 4, d["raised"])
    bad = ~d["raised"] & ~np.all(np.isfinite(d["y"]), axis=1)
    assert np.all(o["status"][bad] == 3) and np.allclose(o["x"][bad][:, :2], d["x"][bad])
    # noise-free white-matter decay (k 1000, T2 110 ms): exact, and equal to the reference's result
    names = [str(n) for n in d["edge_names"]]
    i = names.index("clean_wm")
    if prior or d["y"][i, 0] <= 1000.0:
        assert abs(o["x"][i, 1] - 110.0) < 1e-2 and abs(o["x"][i, 0] - 1000.0) < 0.2
        assert abs(o["x"][i, 1] - d["x"][i, 1]) < 2e-2
    # objective value reported is the reference's objective at the returned point
    te = d["te"]
    for r in rows[:20]:
        f = np.mean((d["y"][r].astype(np.float64) - o["x"][r, 0] * np.exp(-te / o["x"][r, 1])) ** 2)
        assert np.isclose(o["fun"][r], f, rtol=1e-12)


def test_lbfgsb_lane_reproduces_the_notebook_known_answer():
    """The reference's one committed known-answer (tests/test_oracle_golden.py NOTEBOOK_*): the lane solver walks the
    same 13 iterations as scipy on the printed ROI means, nine echoes (eight unrolled + one in the tail loop)."""
    import copy

    from oracle import t2fit_oracle as oracle
    from test_oracle_golden import NOTEBOOK_MEAN, NOTEBOOK_PARAMS, NOTEBOOK_TE

    want = oracle.fit_voxel(0, "gaussian", copy.deepcopy(NOTEBOOK_PARAMS), NOTEBOOK_TE, NOTEBOOK_MEAN[None, :], True, False)
    cfg = sim.config("gaussian", True, NOTEBOOK_TE, solver="lbfgsb")
    cfg.x0[0], cfg.x0[1] = 630.0, 165.0
    cfg.lb[0], cfg.ub[0], cfg.lb[1], cfg.ub[1] = float(NOTEBOOK_MEAN[0]), 1e4, 10.0, 600.0
    cfg.ftol = 1e-6
    o = sim.fit_rows(cfg, NOTEBOOK_MEAN[None, :])
    assert o["nit"][0] == want[2] == 13 and o["status"][0] == 1
    assert abs(o["x"][0, 1] - want[0][1]) < 1e-3 and abs(o["x"][0, 0] - want[0][0]) < 1e-2


@pytest.mark.parametrize("dim", [2, 3])
def test_correction_pair_ring_keeps_the_direction_of_s(dim):
    """The ring stores s as a direction, s / s_0 = (1, s_1/s_0, ...) (t2fit_lbfgsb.h store_s / load_s; the BFGS update
    does not change when s is scaled).  What comes back times s_0 must be s, every entry to 1 ulp, over magnitudes
    1e-300 .. 1e300 and mixed signs; a first component that is zero or below 2^-400 of the largest is replaced by
    +-2^-400 of the largest, so that no ratio exceeds 2^400 (and the direction moves by 2^-400 at most)."""
    rng = np.random.default_rng(5)
    s = rng.normal(size=(4000, dim)) * 10.0 ** rng.integers(-12, 12, size=(4000, dim))
    s[:200, 0] = 0.0                      # a variable at its bound does not move
    s[200:300, 1:] = 0.0                  # only the first moves
    s[300:400] = np.abs(s[300:400, :1])   # all equal
    s[400:420] *= 1e-290
    s[420:440] *= 1e280
    s[440:460, 0] *= 1e-200               # a first component far below the others
    s[460:480, 0] = -0.0
    got = sim.pair_roundtrip(s)
    big = np.max(np.abs(s), axis=1)
    least = big * 2.0 ** -400
    small = np.abs(s[:, 0]) < least
    p0 = np.where(small, np.where(s[:, 0] < 0, -least, least), s[:, 0])
    assert np.array_equal(got[:, 0], np.ones(len(s)))
    assert np.all(np.abs(got) <= 2.0 ** 400 * (1 + 1e-15))
    with np.errstate(over="ignore", under="ignore"):
        back = got[:, 1:] * p0[:, None]
    assert np.allclose(back, s[:, 1:], rtol=4e-16, atol=0.0)
    assert np.array_equal(got[:, 1:] == 0.0, s[:, 1:] == 0.0)
    assert small[:200].all() and small[440:480].all() and not small[200:440].any()


# ---- the Rician evaluation as loops (round 3): each piece against the form it restates --------------------------
def test_log_i0e_four_wide_loop_equals_the_cephes_form():
    """t2_log_i0e4 (table-driven Chebyshev loops, four arguments at once, both series wave-uniform) against
    log(i0e(x)) of the one-value Cephes form, bit for bit: both ranges, the boundary 8, mixed groups, 0, negative
    arguments, huge arguments; and against scipy to 1e-13."""
    from scipy.special import i0e

    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0, 8, (500, 4)), rng.uniform(8, 2000, (500, 4)), 10.0 ** rng.uniform(-6, 6, (500, 4)),
                        -10.0 ** rng.uniform(-3, 3, (100, 4)),
                        np.array([[0.0, 8.0, np.nextafter(8.0, 9.0), 1e300], [7.999, 8.001, -8.0, -0.0]])])
    out, ref = sim.log_i0e4(x)
    assert np.array_equal(out, ref)
    assert np.allclose(out, np.log(i0e(x)), rtol=1e-13, atol=1e-15)
    nan_out, nan_ref = sim.log_i0e4(np.array([[np.nan, 1.0, 100.0, np.inf]]))
    assert np.array_equal(nan_out, nan_ref, equal_nan=True) and np.isnan(nan_out[0, 0])


def test_lean_log_is_within_one_ulp():
    """t2_log_lean (fdlibm's __ieee754_log restated: what the Rician lane applies to i0e's values) against a 50-digit
    logarithm: below 1 ulp on the range i0e can return ((0, 1]) and far beyond it, exact at 1."""
    from decimal import Decimal, getcontext

    getcontext().prec = 50
    rng = np.random.default_rng(13)
    x = np.concatenate([rng.uniform(0.0, 1.0, 4000), 10.0 ** rng.uniform(-12, 0, 2000), 10.0 ** rng.uniform(0, 12, 500),
                        1.0 + rng.uniform(-1e-3, 1e-3, 1000), np.array([1.0, 0.5, 2.0, 0.7071067811865476, 1e-300, 1e300])])
    x = x[x > 0]
    got = sim.log_lean(x)
    worst = 0.0
    for xi, gi in zip(x[::7], got[::7]):  # (Decimal is slow: every seventh point)
        exact = Decimal(float(xi)).ln()
        ulp = Decimal(float(np.spacing(abs(gi)))) if gi != 0 else Decimal(5e-324)
        worst = max(worst, float(abs(Decimal(float(gi)) - exact) / ulp))
    assert worst < 1.0, worst
    assert sim.log_lean(np.array([1.0]))[0] == 0.0
    assert np.allclose(got, np.log(x), rtol=3e-16, atol=3e-19)


def test_i0e_shared_loop_equals_the_cephes_form():
    """t2_i0e4_by_lane -- the 30-step loop a wave runs when some of its lanes need the [0, 8] series and others the
    (8, inf) one, each lane picking its own series' coefficient (the (8, inf) table behind five zero steps) -- against
    the one-value Cephes form, bit for bit, for rows on either side, at the boundary and at the extremes; scipy to 1e-14."""
    from scipy.special import i0e

    rng = np.random.default_rng(12)
    lo = rng.uniform(0, 8, (800, 4))
    hi = np.concatenate([rng.uniform(8, 50, (400, 4)), 10.0 ** rng.uniform(1, 8, (400, 4))])
    hi[hi <= 8.0] = 9.0
    edge = np.array([[0.0, 8.0, 7.999999999, 1e-300], [np.nextafter(8.0, 9.0), 8.5, 1e300, 1e10], [-3.0, -0.0, -8.0, 2.5],
                     [-9.0, -1e5, 40.0, 8.000001]])
    x = np.concatenate([lo, hi, edge])
    out, ref = sim.i0e4_by_lane(x)
    assert np.array_equal(out, ref)
    assert np.allclose(out, i0e(x), rtol=1e-14, atol=0)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8, 9, 11, 15, 16, 17, 24, 31, 32])
def test_echo_loop_row_sums_are_numpys(n):
    """RowSums4 fed item by item == np.sum of each column, bit for bit (numpy's pairwise order: left to right below
    eight items, eight interleaved partial sums + tree + tail from eight on), in the run-time form for every n and
    in the compile-time echo-count form for 3..8."""
    rng = np.random.default_rng(n)
    for _ in range(200):
        t = rng.normal(size=(n, 4)) * 10.0 ** rng.integers(-4, 5, size=(n, 4))
        want = np.array([np.sum(np.ascontiguousarray(t[:, j])) for j in range(4)])
        assert np.array_equal(sim.rowsums4(t, False), want)
        if 3 <= n <= 8:
            assert np.array_equal(sim.rowsums4(t, True), want)


@pytest.mark.parametrize("n_te", [3, 5, 6, 8, 9, 16, 20, 32])
@pytest.mark.parametrize("legacy", [False, True])
def test_rician_echo_loop_equals_the_statement_by_statement_objective(n_te, legacy):
    """Lbfgsb<RICIAN>::eval (one loop over the echoes, four objective values at once) against objective_t (one
    objective at a time, the reference's statements one by one): f(x) bit for bit, in the run-time echo-count form and
    in the specialised forms (whose sample rotation must come back to where it started); the forward-difference
    gradient against differences of the oracle's objective."""
    from oracle import t2fit_oracle as oracle

    rng = np.random.default_rng(100 + n_te)
    te = np.round(np.linspace(30.0, 400.0, n_te))
    cfg = sim.config("rician", True, te, numpy_legacy=legacy)
    obj = oracle._OBJ_LEGACY["rician"] if legacy else oracle._OBJ["rician"]
    for _ in range(40):
        k, t2v, sg = rng.uniform(560, 890), rng.uniform(20, 500), rng.choice([3.0, 20.0, 40.0, 300.0, 900.0])
        clean = k * np.exp(-te / t2v)
        row = np.hypot(clean + rng.normal(size=n_te) * sg, rng.normal(size=n_te) * sg).astype(np.float32)
        x = np.array([rng.uniform(551, 899), rng.uniform(11, 599), rng.uniform(2.1, 999)])
        for special in ([False, True] if n_te <= 8 else [False]):
            out, ref = sim.rician_eval(cfg, row, x, special)
            assert out[0] == ref, (n_te, special, out[0], ref)
        with np.errstate(all="ignore"):
            f0 = obj(x, te, row)
            grad = [(obj(x + 1e-8 * np.eye(3)[j], te, row) - f0) / ((x[j] + 1e-8) - x[j]) for j in range(3)]
        assert np.isclose(out[0], f0, rtol=1e-6)  # np.log of a float32 array is not glibc's logf to the last bit: a constant shift
        assert np.allclose(out[1:], grad, rtol=1e-5, atol=2e-3)  # differences of nearly equal numbers: libm last bits


def test_shared_seed_square_roots_are_the_correctly_rounded_ones():
    """t2_sqrt_from_seed_seq / t2_sqrt_near_seq / t2_sqrt_from_h_seq (t2fit_lane.h): the FMA sequences behind the square roots of
    an evaluation, here on the CPU from a float-precision seed (coarser than v_rsq_f64's): the base root, the roots of three
    arguments within 2^-21 of the base argument started from the base's refined reciprocal root, and the root rebuilt from
    the refined reciprocal root alone -- all bit for bit numpy's (IEEE) sqrt, on two million arguments across the
    radicands' range, at the edge of the allowed distance, and at exact squares and their neighbours."""
    rng = np.random.default_rng(5)
    n = 500_000
    x = 10.0 ** rng.uniform(-2, 10, n)
    rel = np.concatenate([rng.uniform(-1, 1, (n // 2, 3)) * 1e-8, rng.uniform(-1, 1, (n - n // 2, 3)) * 2.0 ** -21])
    a = x[:, None] * (1.0 + rel)
    sq = np.floor(rng.uniform(1, 3e4, 2000)) ** 2  # exact squares: neighbours one ulp either side must round the right way
    x = np.concatenate([x, sq])
    a = np.concatenate([a, np.stack([np.nextafter(sq, 0), sq, np.nextafter(sq, np.inf)], axis=1)])
    base, near, from_h = sim.sqrt_near(x, a)
    assert np.array_equal(base, np.sqrt(x))
    assert np.array_equal(near, np.sqrt(a))
    assert np.array_equal(from_h, np.sqrt(a))


def test_i0e_shared_reciprocal_root_equals_independent_divisions():
    """t2_i0e4_by_lane(near=True) -- 32 / x and the division by sqrt(x) of the (8, inf) series through one shared reciprocal
    square root -- against the same loop with its independent divisions and roots, bit for bit, for rows of four arguments
    within 1e-8 (the reference's step) and within 2^-21 of each other on either side of 8."""
    rng = np.random.default_rng(6)
    base = np.concatenate([rng.uniform(8.0001, 60, 60000), 10.0 ** rng.uniform(1, 7, 60000), rng.uniform(1e-3, 7.9999, 20000)])
    rel = np.where(rng.random((len(base), 1)) < 0.5, 1e-8, 2.0 ** -21) * rng.uniform(-1, 1, (len(base), 4))
    rel[:, 0] = 0.0
    x = base[:, None] * (1.0 + rel)
    out, ref = sim.i0e4_by_lane_near(x)
    assert np.array_equal(out, ref)
