"""Which kernels of another stream run WHILE the persistent reference-trajectory fit is in flight, and does leaving
whole CUs free (T2FIT_RESERVE_CUS) change that?  Side-stream kernels: a device-to-device copy (no LDS), a softmax over
rows (LDS + registers, the footprint class of a collective's kernel).  The side work is enqueued 2 ms after the fit
was launched (host sleep), so host-side enqueue time cannot be mistaken for waiting; HIP events bracket everything.

    python tools/overlap_check2.py           # runs itself with T2FIT_RESERVE_CUS = 0, 8, 16, 32
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if len(sys.argv) == 1:
    for r in ("0", "8", "16", "32"):
        env = dict(os.environ, T2FIT_RESERVE_CUS=r)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
    sys.exit(0)

import torch  # noqa: E402

import fetal_t2mapping_amd as t2  # noqa: E402
from fetal_t2mapping_amd import _abi, synth  # noqa: E402
from fetal_t2mapping_amd._lib import check, require_gpu  # noqa: E402

lib = require_gpu()
dev = torch.device("cuda", 0)
shape = (256, 256, 256)
n = 256 ** 3
e, m, te = synth.brain_volume_torch(shape, 8, synth.SEED_BASE + 3, dev)
cfg = t2.make_config("gaussian_rician", t2.fit_table("gaussian_rician", True), te)
out = torch.empty((4, n), dtype=torch.float32, device=dev)
maps = _abi.T2FitMaps()
maps.t2, maps.k, maps.sigma, maps.res = (out[j].data_ptr() for j in range(4))
a = torch.randn(1 << 26, dtype=torch.float32, device=dev)  # 256 MiB
b = torch.empty_like(a)
rows = a.view(1 << 16, 1 << 10)
sa, sb = torch.cuda.current_stream(), torch.cuda.Stream()


def fit():
    check(lib.t2fit_volume_dev(C.byref(cfg), e.data_ptr(), _abi.LAYOUT_TE_MAJOR, m.data_ptr(), n, C.byref(maps),
                               C.c_void_p(sa.cuda_stream)))


def side(kind):
    if kind == "copy":
        b.copy_(a)
    else:
        torch.softmax(rows, dim=1, out=b.view_as(rows))


res = {"reserve_cus": int(os.environ.get("T2FIT_RESERVE_CUS", "0"))}
for kind in ("copy", "softmax"):
    fit(); side(kind); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        t0.record(sb); side(kind); t1.record(sb)
    torch.cuda.synchronize()
    alone = t0.elapsed_time(t1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(sa)
    h0 = time.perf_counter()
    fit()
    enqueue_ms = (time.perf_counter() - h0) * 1e3
    ev[1].record(sa)
    time.sleep(0.002)
    with torch.cuda.stream(sb):
        ev[2].record(sb); side(kind); ev[3].record(sb)
    torch.cuda.synchronize()
    res[kind] = {"alone_ms": round(alone, 3), "fit_enqueue_host_ms": round(enqueue_ms, 3), "fit_ms": round(ev[0].elapsed_time(ev[1]), 2),
                 "side_start_ms_after_fit_start": round(ev[0].elapsed_time(ev[2]), 2),
                 "side_end_ms_after_fit_start": round(ev[0].elapsed_time(ev[3]), 2)}
print(json.dumps(res))
