#!/usr/bin/env python3
"""Headline benchmark: Mvoxel/s of the per-voxel T2 fit on a synthetic 256^3 x 8 TE volume.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over the volume: the fit kernel over voxels x 8 echoes already resident in HBM
(t2fit_volume_dev) and, for N > 1, the RCCL all-gather of the four output maps that BASELINE.json's north_star asks
for.  Default: STRONG scaling, as BASELINE.json's metric words it ("256^3 x 8TE ... 1/2/4/8 GPU"): ONE 256^3 volume,
cut over the N ranks in chunks of 16 Ki voxels dealt rank by rank (fetal_t2mapping_amd/dist.py: every rank gets the
same share of every region of the mask, contiguous Z-slabs of the ellipsoid do not), each rank fits its share, one
all-gather puts the four maps of the whole volume on every rank.  `--scaling weak`: every rank fits its own 256^3
volume (round 1's mode).  `value` = dense voxels of the volume(s) / wall time (max over ranks) in Mvoxel/s.

The JSON line also carries
  roofline     : achieved algorithmic HBM GB/s of the dominant kernel, the fit (45 B/voxel at 8 TE: 4*nTE samples + 1
                 mask byte + the three float32 parameter maps it writes) over its mean launch duration on rank 0,
                 measured with HIP events on the launch stream inside the library, against 8 TB/s; `epilogue` beside it
                 is the streaming pass that follows every fit and writes the fourth map, `res` (49 B/voxel: samples,
                 mask, the three maps read back, res written).  SURVEY.md 8d's 49 B/voxel is the two together.
  cpu_baseline : the CPU oracle (oracle/t2fit_oracle.py: the reference's scipy L-BFGS-B loop restated) timed on this
                 host's cores over a bounded sample of the same masked voxels; on the same basis as `value` (dense
                 voxels/s at the bench volume's mask fill), with the fitted-voxel rates of both beside it.
  cpu_baseline_native : SURVEY.md 8d (ii): the build's own lane solver (the headers the kernels are compiled from, built
                 by g++ into tests/hostsim: test infrastructure, never loaded by the product) on all host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# one BLAS/OpenMP thread per process, set before numpy loads: the CPU baseline runs one scipy fit
# per worker process and oversubscribed BLAS threads slow it by >10x
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may actually use: affinity, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("T2FIT_BENCH_CORES", "64"))))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
CLOCK_MHZ, N_SIMD = 2400.0, 256 * 4  # MI355X_MICROARCH.md: 2400 MHz max clock, 256 CUs x 4 SIMD16
VEC_F64_PEAK_TFLOPS = 256 * 4 * 16 * 2 * 2.4e9 / 1e12  # vector (non-MFMA) float64 FMA peak: 78.6 TFLOP/s


def alu_view(key, kernel_ms):
    """What actually bounds the iterative fits: VALU issue slots.  Instruction counts per launch come from separate
    rocprofv3 --pmc passes of this same workload (profiles/instr_mix.json, deterministic per launch); the time is live."""
    path = os.path.join(REPO, "profiles", "instr_mix.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        m = json.load(f).get(key)
    if not m:
        return None
    cycles = kernel_ms * 1e-3 * CLOCK_MHZ * 1e6 * N_SIMD
    flops = (2 * m.get("valu_fma_f64", 0) + m.get("valu_mul_f64", 0) + m.get("valu_add_f64", 0)
             + m.get("valu_trans_f64", 0)) * 64 * m["lanes_active"]
    return {"valu_wave_insts_per_launch": m["valu"], "lanes_active": m["lanes_active"],
            "valu_issue_utilisation": round(m["valu"] * 4 / cycles, 3),  # >= 4 cycles per wave64 VALU instruction
            "f64_tflops_active_lanes": round(flops / (kernel_ms * 1e-3) / 1e12, 2),
            "vector_f64_peak_tflops": round(VEC_F64_PEAK_TFLOPS, 1),
            "source": "profiles/instr_mix.json (rocprofv3 --pmc SQ_INSTS_VALU*, own passes) / live kernel time"}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--shape", type=int, nargs=3, default=[256, 256, 256], help="volume Z Y X (per rank with --scaling weak)")
    p.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                   help="strong (default): one volume cut over the ranks; weak: one volume per rank")
    p.add_argument("--partition", default="cyclic", choices=["cyclic", "slab"],
                   help="strong scaling: chunks dealt rank by rank (balanced, default) or contiguous flat ranges")
    p.add_argument("--n-te", type=int, default=8)
    p.add_argument("--fit", default="gaussian_rician", choices=["gaussian", "gaussian_rician", "rician"])
    p.add_argument("--solver", default="lbfgsb", choices=["lbfgsb", "lm", "loglin"])
    p.add_argument("--precision", default="f64", choices=["f64", "f32"])
    p.add_argument("--no-prior", action="store_true")
    p.add_argument("--no-gather", action="store_true", help="skip the all-gather of the maps (N > 1)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration (0 = skip)")
    p.add_argument("--no-also", action="store_true", help="skip the secondary (converged LM float32) measurement")
    p.add_argument("--reserve-cus", type=int, default=0,
                   help="N > 1 with the all-gather: CUs the persistent fit kernel leaves free for RCCL's kernels in the timed "
                        "steps (0 = none, the default).  Whether that pays cannot be measured on one GPU "
                        "(profiles/r02_overlap_*.jsonl: kernels that need little LDS run beside the fit anyway, the fit's "
                        "workgroups hold 150 of a CU's 160 KiB), so the N > 1 run times the other setting too, after the "
                        "timed steps, and reports it beside `value` as `reserve_cus_ab`")
    p.add_argument("--pipeline", choices=["auto", "on", "off"], default="auto",
                   help="consecutive steps alternate between two streams, so that the drain of one fit launch (a few waves "
                        "finishing their last voxels, about 1 ms whatever the share size) overlaps the start of the next: "
                        "auto = on for N > 1 (where a rank's share is small and the drain is a third of the launch), off for N = 1")
    p.add_argument("--pipeline-streams", type=int, default=2, help="streams the steps rotate over when pipelined (N = 1 runs; N > 1 uses two)")
    p.add_argument("--reserve-cus-ab", type=int, default=8, help="the other setting measured for `reserve_cus_ab` (N > 1)")
    return p.parse_args()


def cpu_baseline(echoes_rows, te, fit, prior, seconds):
    """Reference-equivalent scipy loop (the oracle) on all host cores over a bounded voxel sample."""
    import multiprocessing as mp

    import numpy as np

    from oracle import t2fit_oracle as O

    cores = host_cores()
    log(f"cpu baseline: {cores} worker processes")
    table = O.fit_table(fit, True)
    # calibrate on a few voxels, then size the sample for ~`seconds` of wall time
    t0 = time.perf_counter()
    probe = min(40, echoes_rows.shape[0])
    O.fit_volume(echoes_rows[:probe], np.arange(probe), te, fit, table, prior=prior)
    per_voxel = (time.perf_counter() - t0) / probe
    n = int(min(echoes_rows.shape[0], max(cores * 50, seconds * cores / per_voxel)))
    log(f"cpu baseline: {per_voxel * 1e3:.2f} ms/voxel single process, timing {n} voxels")
    rows = echoes_rows[:n]
    pool = mp.get_context("fork").Pool(cores)
    try:
        t0 = time.perf_counter()
        O.fit_volume(rows, np.arange(n), te, fit, table, prior=prior, pool=pool)
        dt = time.perf_counter() - t0
    finally:
        pool.close()  # workers leave on their own (no SIGTERM: under a profiler that reads as an abort)
        pool.join()
    return {"fitted_value": n / dt / 1e6, "unit": "Mvoxel/s", "cores": cores, "kind": "port",
            "sample": f"{n} masked voxels of a 6-slice slab of the same synthetic distribution, scipy {__import__('scipy').__version__} "
                      f"L-BFGS-B loop (oracle/t2fit_oracle.py) on a {cores}-process pool, {dt:.1f} s"}


def cpu_baseline_native(echoes_rows, te, fit, prior, seconds):
    """the build's own C++ lane solver on every host core.  tests/hostsim compiles the very headers
    the HIP kernels are made of with g++ (test infrastructure: the product never loads it); ctypes releases the GIL, so
    a thread per core runs its own block of rows."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    sys.path.insert(0, os.path.join(REPO, "tests"))
    try:
        from hostsim import sim
        sim.lib()
    except Exception as e:  # not built (no g++ on this box): the figure is optional
        log(f"native cpu baseline skipped: {e}")
        return None
    cores = host_cores()
    cfg = sim.config(fit, True, te, prior=prior, solver="lbfgsb")
    t0 = time.perf_counter()
    sim.fit_rows(cfg, echoes_rows[:2000])
    per_voxel = (time.perf_counter() - t0) / min(2000, len(echoes_rows))
    n = int(min(len(echoes_rows), max(cores * 2000, seconds * cores / per_voxel)))
    blocks = np.array_split(np.arange(n), cores * 4)
    with ThreadPoolExecutor(cores) as pool:
        t0 = time.perf_counter()
        list(pool.map(lambda b: sim.fit_rows(cfg, echoes_rows[b[0]: b[-1] + 1]), [b for b in blocks if len(b)]))
        dt = time.perf_counter() - t0
    return {"fitted_value": n / dt / 1e6, "unit": "Mvoxel/s", "cores": cores, "kind": "port",
            "sample": f"{n} masked voxels of the same slab, the lane solver of fetal_t2mapping_amd/csrc/t2fit_lbfgsb.h compiled "
                      f"by g++ (tests/hostsim), one thread per core, {dt:.1f} s"}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import fetal_t2mapping_amd as t2
    from fetal_t2mapping_amd import _abi, synth
    from fetal_t2mapping_amd import dist as t2dist
    from fetal_t2mapping_amd._lib import check, require_gpu

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        a.gpus = world
    cpu = cpu_native = None
    if world == 1 and a.cpu_seconds > 0:
        # timed BEFORE this process touches the GPU (the worker pool forks); a thin slab of the
        # same synthetic distribution: same TE vector, k/T2/noise ranges and mask shape
        ev, mv, te_c = synth.brain_volume((6, a.shape[1], a.shape[2]), a.n_te, synth.SEED_BASE + 3)
        rows = np.ascontiguousarray(ev.reshape(a.n_te, -1)[:, mv.reshape(-1) != 0].T)
        cpu = cpu_baseline(rows, te_c, a.fit, not a.no_prior, a.cpu_seconds)
        cpu_native = cpu_baseline_native(rows, te_c, a.fit, not a.no_prior, min(5.0, a.cpu_seconds))
    log("cpu baseline done" if cpu else "no cpu baseline")
    # (the pool's host driver shares device memory between processes through dmabuf only; already exported there)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = require_gpu()
    # T2FIT_BENCH_BACKEND=gloo is a REHEARSAL of the N > 1 control flow on a one-GPU box: every rank
    # uses cuda:0 and the gather is staged through the host.  Numbers from it mean nothing.
    backend = os.environ.get("T2FIT_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    z, y, x = a.shape
    n_vol = z * y * x
    strong = a.scaling == "strong"
    # strong: every rank generates the SAME volume (same seed) and keeps its share; weak: one volume per rank
    echoes, mask, te = synth.brain_volume_torch((z, y, x), a.n_te, synth.SEED_BASE + 3 + (0 if strong else rank), dev)
    masked_vol = int(mask.sum().item())
    src_index = None
    if strong and world > 1:
        if a.partition == "cyclic":
            n_mine = t2dist.cyclic_len(n_vol, world)
            idx = t2dist.cyclic_index(n_vol, rank, world)
        else:
            n_mine = t2dist.slab_len(n_vol, world)
            lo, hi = t2dist.slab_range(n_vol, rank, world)
            idx = np.full(n_mine, -1, np.int64)
            idx[: hi - lo] = np.arange(lo, hi)
        it = torch.from_numpy(np.where(idx >= 0, idx, 0)).to(dev)
        real = torch.from_numpy(idx >= 0).to(dev)
        echoes = echoes[:, it].contiguous()                       # this rank's share, resident before the timed region
        mask = (mask[it] * real.to(torch.uint8)).contiguous()     # padding slots: mask 0
        del it, real
        if a.partition == "cyclic":   # rows of the gathered [world * slots] chunk table in voxel order (dist.gather_maps_cyclic)
            slots = n_mine // t2dist.CHUNK
            c = np.arange(slots * world, dtype=np.int64)
            g = c // world
            src_index = torch.from_numpy(((c % world + t2dist._rotation(g, world)) % world) * slots + g).to(dev)
    else:
        n_mine = n_vol
    n_vox = n_mine
    masked_mine = int(mask.sum().item())
    log(f"rank {rank}: {n_vox} voxels x {a.n_te} TE on device ({masked_mine} in the mask), scaling {a.scaling}")
    table = t2.fit_table(a.fit, True)
    cfg = t2.make_config(a.fit, table, te, prior=not a.no_prior, norm=False, solver=a.solver, precision=a.precision)
    # packed output [4, n_vox]: t2, k, sigma, res -- one all-gather moves all four maps.
    # Two packed buffers / two gather targets alternate so that the RCCL all-gather of step i (RCCL's own
    # stream, xGMI) overlaps the fit kernel of step i+1; a buffer is reused only after the gather that
    # read it has been waited for on the compute stream.
    do_gather = world > 1 and not a.no_gather
    pipelined = a.pipeline == "on" or (a.pipeline == "auto" and world > 1)
    n_streams = max(2, a.pipeline_streams) if (pipelined and not do_gather) else 2
    side_pipeline = world == 1 and not pipelined and a.solver == "lbfgsb" and not a.no_also
    n_buf = n_streams if pipelined else (2 if (do_gather or side_pipeline) else 1)
    packed = [torch.empty((4, n_vox), dtype=torch.float32, device=dev) for _ in range(n_buf)]
    gathered = [torch.empty((world, 4, n_vox), dtype=torch.float32, device=dev) for _ in range(2)] if do_gather else None
    ordered = ([torch.empty((4, world * n_vox), dtype=torch.float32, device=dev) for _ in range(2)]
               if (do_gather and src_index is not None) else None)
    # The reference-trajectory kernel is persistent and its workgroups fill every CU's LDS; t2fit_set_reserve_cus makes
    # the library launch it over (CUs - reserve) CUs so that RCCL's kernels find free CUs while it runs.
    can_reserve = do_gather and a.solver == "lbfgsb" and "T2FIT_PERSISTENT_BLOCKS" not in os.environ
    reserved = a.reserve_cus if can_reserve else 0
    lib.t2fit_set_reserve_cus(reserved)
    maps_b = []
    for pk in packed:
        mb = _abi.T2FitMaps()
        mb.t2, mb.k, mb.sigma, mb.res = (pk[j].data_ptr() for j in range(4))
        maps_b.append(mb)
    maps = maps_b[0]
    pending = [None] * max(2, n_buf)
    # step i runs on stream i % 2 when pipelined (its fit, its all-gather dependency, its reordering), else on the current stream
    streams = [torch.cuda.Stream() for _ in range(n_streams)] if pipelined else [torch.cuda.current_stream()]
    lib.t2fit_set_timing(1)
    kernel_ms, epilogue_ms, step_wall = [], [], []
    step_no = [0]

    def stream_of(b):
        return streams[b % len(streams)]

    def reorder(b):
        """strong scaling, cyclic partition: the gathered chunk table back into voxel order (one gather of 64 KiB rows)."""
        if ordered is not None:
            slots = n_vox // t2dist.CHUNK
            by_chunk = gathered[b].view(world, 4, slots, t2dist.CHUNK).permute(1, 0, 2, 3).reshape(4, world * slots, t2dist.CHUNK)
            torch.index_select(by_chunk, 1, src_index, out=ordered[b].view(4, world * slots, t2dist.CHUNK))

    def step(record):
        b = step_no[0] % (len(packed) if (pipelined or do_gather) else 1)
        step_no[0] += 1
        with torch.cuda.stream(stream_of(b)):
            if pending[b] is not None:
                pending[b].wait()  # stream-side wait: buffer b is free again
                reorder(b)
                pending[b] = None
            check(lib.t2fit_volume_dev(C.byref(cfg), echoes.data_ptr(), _abi.LAYOUT_TE_MAJOR, mask.data_ptr(), n_vox,
                                       C.byref(maps_b[b]), C.c_void_p(stream_of(b).cuda_stream)))
            if do_gather:
                if rehearsal:
                    parts = [torch.empty((4, n_vox), dtype=torch.float32) for _ in range(world)]
                    dist.all_gather(parts, packed[b].cpu())
                    gathered[b].copy_(torch.stack(parts))
                    reorder(b)
                else:
                    pending[b] = dist.all_gather_into_tensor(gathered[b].view(-1), packed[b].view(-1), async_op=True)
        if record and not pipelined:
            kernel_ms.append(lib.t2fit_last_kernel_ms())  # syncs on the kernel's stop event only
            step_wall.append(time.perf_counter())

    def drain():
        for b in range(len(pending)):
            if pending[b] is not None:
                with torch.cuda.stream(stream_of(b)):
                    pending[b].wait()
                    reorder(b)
                pending[b] = None

    for _ in range(a.warmup):
        step(False)
        drain()
        torch.cuda.synchronize()
        log("warmup step done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    drain()  # every all-gather (and reordering) of the timed steps has completed inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if pipelined:  # kernel times of the last timed launches, read now that they are done (no stall inside the region)
        kernel_ms = [lib.t2fit_kernel_ms(k) for k in range(min(a.steps, 16))]
    epilogue_ms = [lib.t2fit_epilogue_ms(k) for k in range(min(a.steps, 16))]
    # per-step wall times (one stream: each step ends where the host has seen its fit kernel finish; the 0.2 ms epilogue of
    # step i runs under the launch of step i + 1, so the last one is inside `elapsed` but in no step of this list)
    step_ms = [1e3 * (b - a_) for a_, b in zip([t0] + step_wall[:-1], step_wall)] if step_wall else []
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # N = 1: the same K steps rotating over two streams, after the timed region: the drain of a launch (the last voxels in
    # flight, ~0.8 ms) then overlaps the start of the next launch.  Reported beside `value`, never as `value`: the headline
    # stays the plain one-stream figure whose kernel time is the roofline's.
    piped = None
    if side_pipeline:
        pipelined = True
        streams[:] = [torch.cuda.Stream(), torch.cuda.Stream()]
        for _ in range(2):
            step(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step(False)
        torch.cuda.synchronize()
        piped = {"ms_per_step": round((time.perf_counter() - t1) / a.steps * 1e3, 4)}
        pipelined = False
        streams[:] = [torch.cuda.current_stream()]
        step_no[0] = 0
    # the other reserve-CUs setting, same steps, after the timed region (N > 1 only): reported, never `value`
    reserve_ab = None
    if can_reserve and a.reserve_cus_ab != reserved and a.reserve_cus_ab >= 0:
        lib.t2fit_set_reserve_cus(a.reserve_cus_ab)
        step(False)
        drain()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step(False)
        drain()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t1
        tt = torch.tensor([e2], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        reserve_ab = {"cus_left_free_for_rccl": a.reserve_cus_ab, "ms_per_step": round(float(tt.item()) / a.steps * 1e3, 4)}
        lib.t2fit_set_reserve_cus(reserved)

    # strong scaling self-check (outside the timed region): the maps every rank now holds for the whole volume are,
    # bit for bit, the maps of a single-GPU fit of that volume (rank 0 fits it alone and compares)
    verified = None
    if strong and do_gather and rank == 0:
        e_all, m_all, _ = synth.brain_volume_torch((z, y, x), a.n_te, synth.SEED_BASE + 3, dev)
        alone = torch.empty((4, n_vol), dtype=torch.float32, device=dev)
        ma = _abi.T2FitMaps()
        ma.t2, ma.k, ma.sigma, ma.res = (alone[j].data_ptr() for j in range(4))
        check(lib.t2fit_volume_dev(C.byref(cfg), e_all.data_ptr(), _abi.LAYOUT_TE_MAJOR, m_all.data_ptr(), n_vol, C.byref(ma),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        last = (step_no[0] - 1) % len(packed)
        if ordered is not None:
            reorder(last)
            whole = ordered[last][:, :n_vol]
        else:
            whole = gathered[last].permute(1, 0, 2).reshape(4, world * n_vox)[:, :n_vol]
        verified = bool(((whole == alone) | (whole.isnan() & alone.isnan())).all().item())
        del e_all, m_all, alone

    # secondary measurements, same data on this rank: the converged bounded-LM solver (north_star's "per-lane
    # Levenberg-Marquardt") in float32 and in the reference's float64, and the closed form; never `value`
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def measure(cfg_x):
        """Mean kernel time (HIP events on the launch stream) and mean wall time per launch of one more solver."""
        n2 = max(3, min(10, a.steps))
        ks = []
        for i in range(n2 + 1):
            check(lib.t2fit_volume_dev(C.byref(cfg_x), echoes.data_ptr(), _abi.LAYOUT_TE_MAJOR, mask.data_ptr(), n_vox,
                                       C.byref(maps), st))
            k = lib.t2fit_last_kernel_ms()
            if i:
                ks.append(k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n2):
            check(lib.t2fit_volume_dev(C.byref(cfg_x), echoes.data_ptr(), _abi.LAYOUT_TE_MAJOR, mask.data_ptr(), n_vox,
                                       C.byref(maps), st))
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t1) / n2
        kk = float(np.mean(ks))
        return {"per_gpu_value": round(n_vox / dt2 / 1e6, 3), "per_gpu_fitted_value": round(masked_mine / dt2 / 1e6, 3),
                "unit": "Mvoxel/s", "ms_per_step": round(dt2 * 1e3, 4), "kernel_ms": round(kk, 4),
                # bytes the timed kernel moves: the iterative fits write three maps (res comes from their epilogue pass)
                "roofline_frac": round((4 * a.n_te + (17 if cfg_x.solver == _abi.SOLVER_LOGLIN else 13)) * n_vox / (kk * 1e-3) / 1e9
                                       / HBM_PEAK_GBS, 6)}

    also = also_f64 = also_loglin = also_rician = also_noprior = None
    if a.solver == "lbfgsb" and a.fit != "rician" and not a.no_also:
        # the reference's other configurations on the same stack, so that a regression there is visible in this line:
        # the Rician-likelihood objective (--rician) and the paper's bounds (--no_prior)
        cfg_r = t2.make_config("rician", t2.fit_table("rician", True), te, prior=not a.no_prior, norm=False, solver="lbfgsb")
        also_rician = {"solver": "lbfgsb", "fit": "rician", "dtype": "f64", **measure(cfg_r),
                       "note": "reference-trajectory L-BFGS-B on the Rician negative log-likelihood (run_t2mapping.py:157-177), same stack"}
        cfg_np = t2.make_config(a.fit, table, te, prior=a.no_prior, norm=False, solver="lbfgsb")
        also_noprior = {"solver": "lbfgsb", "fit": a.fit, "dtype": "f64", "prior": bool(a.no_prior), **measure(cfg_np),
                        "note": "the same objective under the other bounds setting (k >= S(TE0), T2 in [10, 2000] when prior is "
                                "false: run_t2mapping.py:243-245, the paper's in-vivo configuration)"}
        note = ("converged bounded LM of the same objective; differs from the reference's early-stopped result by design "
                "(DESIGN.md section 2), no all-gather in this figure")
        cfg2 = t2.make_config(a.fit, table, te, prior=not a.no_prior, norm=False, solver="lm", precision="f32")
        also = {"solver": "lm", "dtype": "f32", **measure(cfg2), "note": note}
        cfg2d = t2.make_config(a.fit, table, te, prior=not a.no_prior, norm=False, solver="lm", precision="f64")
        also_f64 = {"solver": "lm", "dtype": "f64", **measure(cfg2d), "note": note + "; the reference's precision"}
        # the one fit on the path that IS bound by HBM: closed-form log-linear 2-parameter fit of the same stack
        # (BASELINE.json config 2 names it; the reference has no such routine), fit + residual map in one pass
        cfg3 = t2.make_config("gaussian", t2.fit_table("gaussian", True), te, prior=not a.no_prior, norm=False,
                              solver="loglin")
        also_loglin = {"solver": "loglin", "fit": "gaussian", "dtype": "f64 sums over f32 log", **measure(cfg3),
                       "kernel": "loglin_volume_kernel",
                       "note": "closed-form weighted log-linear 2-parameter fit, one streaming pass (fit + residual map); "
                               "an extension, not the reference's solver: never `value`"}
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        total_vox = n_vol if strong else world * n_vol
        total_masked = masked_vol if strong else world * masked_vol  # (weak: every rank's mask has the same size)
        value = total_vox / (elapsed / a.steps) / 1e6
        fitted_value = total_masked / (elapsed / a.steps) / 1e6
        fill = masked_vol / n_vol
        iterative = a.solver != "loglin" and not (a.solver == "lm" and os.environ.get("T2FIT_ONE_SHOT", "0") != "0")
        # the iterative fits write t2 / k / sigma; `res` (and its zeros outside the mask) comes from the epilogue pass
        bytes_per_voxel = 4 * a.n_te + 1 + (12 if iterative else 16)
        k_ms = float(np.mean(kernel_ms))
        e_ms = float(np.mean([v for v in epilogue_ms if v >= 0] or [0.0]))
        achieved = bytes_per_voxel * n_vox / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        key = f"{a.fit}/{a.solver}/{a.precision}/{z}x{y}x{x}x{a.n_te}"
        if os.path.exists(tpath) and n_vox == n_vol:  # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/README.md)
            with open(tpath) as f:
                traffic = json.load(f).get(key)
        part = "one volume" if world == 1 else (f"one volume in {a.partition} shares x{world}" if strong else f"one volume per rank x{world}")
        out = {
            "metric": "Mvoxel/s T2 fit, 256\u00b3\u00d78TE 3-param LM, 1/2/4/8 GPU; % HBM roofline",  # BASELINE.json's metric, verbatim
            "value": round(value, 3), "unit": "Mvoxel/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median": round(float(np.median(step_ms)), 4) if step_ms else None,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f64" if a.solver == "lbfgsb" else a.precision, "data": "synthetic",
            "basis": f"`value` counts every voxel of the volume (SURVEY.md 8d: dense voxels/s); mask fill {fill:.3f}: "
                     f"`fitted_value` counts the voxels inside the mask, the only ones that are fitted",
            "fitted_value": round(fitted_value, 3),
            "gathered_maps_equal_single_gpu_fit": verified,
            "config": {"workload": f"{z}x{y}x{x} voxels x {a.n_te} TE ({part}), {a.fit} objective, "
                                   f"{ {'lbfgsb': 'reference-trajectory L-BFGS-B', 'lm': 'bounded LM', 'loglin': 'closed-form log-linear'}[a.solver]} solver, "
                                   f"{'prior' if not a.no_prior else 'no-prior'} bounds, mask fill {fill:.2f}",
                       "solver": a.solver, "fit": a.fit, "n_te": a.n_te, "voxels_total": total_vox,
                       "masked_voxels_total": total_masked, "voxels_rank0": n_vox, "masked_voxels_rank0": masked_mine,
                       "parallelism": part + (" + all-gather of 4 maps (overlapped with the next fit)" if do_gather else ""),
                       "cus_left_free_for_rccl": reserved,
                       "steps_pipelined_over_two_streams": pipelined,
                       "metric_note": "BASELINE.json words the metric '3-param LM'; the reference's solver is scipy L-BFGS-B "
                                      "(SURVEY.md F1) and `value` is the solver that reproduces the reference's maps; the "
                                      "converged LM kernel north_star describes is measured in the same run under `also` "
                                      "(float32) and `also_lm_f64` (the reference's precision)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": "loglin_volume_kernel" if a.solver == "loglin" else "fit_persistent_kernel",
                         "kernel_ms": round(k_ms, 4), "kernel_ms_median": round(float(np.median(kernel_ms)), 4),
                         "epilogue": ({"kernel": "residuals_kernel", "kernel_ms": round(e_ms, 4), "bytes_per_voxel": 4 * a.n_te + 17,
                                       "achieved": round((4 * a.n_te + 17) * n_vox / (e_ms * 1e-3) / 1e9, 3) if e_ms > 0 else None,
                                       "note": "streaming pass after every iterative fit: residual map (4 B/voxel written) from the "
                                               "samples, the mask and the three parameter maps read back; inside `ms_per_step`"}
                                      if iterative else None),
                         "kernel_ms_note": ("steps alternate between two streams: a launch's start-to-end time includes the "
                                            "drain of the launch before it, which it overlaps" if pipelined else None),
                         "bytes_per_voxel": bytes_per_voxel, "voxels_per_launch": n_vox,
                         "note": ("one streaming pass, HBM bound" if a.solver == "loglin" else
                                  "the fit is float64 VALU-issue bound (about 3200 wave instructions per evaluation round of a wave, 2650 of them vector; "
                                  "eight one-wave workgroups per CU, two per SIMD), not HBM bound: see DESIGN.md section 6 "
                                  "and `alu`; bytes_per_voxel counts what THIS kernel moves (samples, mask, three maps)")},
        }
        if piped is not None:
            piped["value"] = round(total_vox / (piped["ms_per_step"] * 1e-3) / 1e6, 3)
            piped["note"] = ("the same steps rotating over two streams (the drain of a launch overlaps the start of the next one), "
                             "timed after the steps of `value`")
            out["steps_over_two_streams"] = piped
        if reserve_ab is not None:
            reserve_ab["value"] = round(total_vox / (reserve_ab["ms_per_step"] * 1e-3) / 1e6, 3)
            reserve_ab["note"] = "the same steps with the other --reserve-cus setting, timed after the steps of `value`"
            out["reserve_cus_ab"] = reserve_ab
        alu = alu_view(key, k_ms) if n_vox == n_vol else None
        if alu is not None:
            out["alu"] = alu
        if cpu is not None:
            cpu["value"] = cpu["fitted_value"] / fill
            cpu["basis"] = (f"`value` = dense-volume equivalent: the fitted-voxel rate divided by the bench volume's mask fill "
                            f"{fill:.3f} (same basis as the headline `value`); `fitted_value` = masked voxels/s as timed")
            out["cpu_baseline"] = cpu
        if cpu_native is not None:
            cpu_native["value"] = cpu_native["fitted_value"] / fill
            out["cpu_baseline_native"] = cpu_native
        if also_rician is not None:
            out["also_rician"] = also_rician
        if also_noprior is not None:
            out["also_noprior"] = also_noprior
        if also is not None:
            out["also"] = also
        if also_f64 is not None:
            out["also_lm_f64"] = also_f64
        if also_loglin is not None:
            out["also_loglin"] = also_loglin
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
