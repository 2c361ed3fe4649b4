"""Timeline of fetal_t2mapping_amd.stream on one GPU: per-stage durations from HIP events and host clocks.
Diagnostic for config 5 (why is a streamed subject slower than max(H2D, fit, D2H)?)."""
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fetal_t2mapping_amd as t2  # noqa: E402
from fetal_t2mapping_amd import _abi, synth  # noqa: E402
from fetal_t2mapping_amd._lib import check, require_gpu  # noqa: E402
from fetal_t2mapping_amd.t2map import make_config  # noqa: E402

solver = sys.argv[1] if len(sys.argv) > 1 else "lm"
precision = sys.argv[2] if len(sys.argv) > 2 else "f32"
shape = (256, 256, 256)
n = 256 ** 3
dev = torch.device("cuda", 0)
e, m, te = synth.brain_volume_torch(shape, 8, synth.SEED_BASE + 5, dev)
lib = require_gpu()
cfg = make_config("gaussian_rician", t2.fit_table("gaussian_rician", True), te, True, False, solver, precision)
h_in = e.reshape(-1).cpu().pin_memory()
h_mask = m.reshape(-1).cpu().pin_memory()
h_out = torch.empty(4 * n, dtype=torch.float32).pin_memory()
d_in = [torch.empty_like(e.reshape(-1)) for _ in range(2)]
d_mask = [torch.empty_like(m.reshape(-1)) for _ in range(2)]
d_out = [torch.empty(4 * n, dtype=torch.float32, device=dev) for _ in range(2)]
compute = torch.cuda.current_stream()
s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()


def ev():
    return torch.cuda.Event(enable_timing=True)


def fit(j):
    maps = _abi.T2FitMaps()
    base = d_out[j].data_ptr()
    maps.t2, maps.k, maps.sigma, maps.res = (base + 4 * n * q for q in range(4))
    check(lib.t2fit_volume_dev(C.byref(cfg), d_in[j].data_ptr(), _abi.LAYOUT_TE_MAJOR, d_mask[j].data_ptr(), n,
                               C.byref(maps), C.c_void_p(compute.cuda_stream)))


out = {}
# each stage alone
for name, fn, st in (("h2d", lambda: (d_in[0].copy_(h_in, non_blocking=True), d_mask[0].copy_(h_mask, non_blocking=True)), s_in),
                     ("fit", lambda: fit(0), compute),
                     ("d2h", lambda: h_out.copy_(d_out[0], non_blocking=True), s_out)):
    ts = []
    for _ in range(4):
        a, b = ev(), ev()
        with torch.cuda.stream(st):
            a.record(st); fn(); b.record(st)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out[name + "_alone_ms"] = [round(t, 2) for t in ts]
# pairs overlapped
for name, pair in (("h2d+fit", ("h2d", "fit")), ("h2d+d2h", ("h2d", "d2h")), ("fit+d2h", ("fit", "d2h")), ("all", ("h2d", "fit", "d2h"))):
    evs = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if "h2d" in pair:
        with torch.cuda.stream(s_in):
            evs["h2d"] = (ev(), ev()); evs["h2d"][0].record(s_in)
            d_in[1].copy_(h_in, non_blocking=True); d_mask[1].copy_(h_mask, non_blocking=True)
            evs["h2d"][1].record(s_in)
    t1 = time.perf_counter()
    if "fit" in pair:
        evs["fit"] = (ev(), ev()); evs["fit"][0].record(compute); fit(0); evs["fit"][1].record(compute)
    t2_ = time.perf_counter()
    if "d2h" in pair:
        with torch.cuda.stream(s_out):
            evs["d2h"] = (ev(), ev()); evs["d2h"][0].record(s_out)
            h_out.copy_(d_out[0], non_blocking=True)
            evs["d2h"][1].record(s_out)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    out[name] = {k: round(v[0].elapsed_time(v[1]), 2) for k, v in evs.items()}
    out[name]["host_enqueue_ms"] = [round((b - a) * 1e3, 2) for a, b in ((t0, t1), (t1, t2_), (t2_, t3))]
    out[name]["wall_ms"] = round((t4 - t0) * 1e3, 2)
print(json.dumps(out))
