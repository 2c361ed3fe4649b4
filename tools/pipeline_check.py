"""Does the drain of one fit launch (a few waves finishing their last voxels, ~1 ms) overlap the start of the next when
consecutive launches alternate between two streams?  Wall time of 12 launches, one stream vs two.
    python tools/pipeline_check.py Z Y X"""
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from fetal_t2mapping_amd import _abi, synth  # noqa: E402
from fetal_t2mapping_amd import t2map as t2  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 256, 256)
lib = t2.require_gpu()
dev = torch.device("cuda", 0)
n = shape[0] * shape[1] * shape[2]
e, m, te = synth.brain_volume_torch(shape, 8, synth.SEED_BASE + 3, dev)
cfg = t2.make_config("gaussian_rician", t2.fit_table("gaussian_rician", True), te, prior=True, norm=False, solver="lbfgsb", precision="f64")
outs = [torch.empty((4, n), dtype=torch.float32, device=dev) for _ in range(2)]
maps = []
for o in outs:
    mb = _abi.T2FitMaps()
    mb.t2, mb.k, mb.sigma, mb.res = (o[j].data_ptr() for j in range(4))
    maps.append(mb)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
lib.t2fit_set_timing(0)


def run(n_streams, steps=12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        b = i % n_streams
        rc = lib.t2fit_volume_dev(C.byref(cfg), C.c_void_p(e.data_ptr()), 0, C.c_void_p(m.data_ptr()), C.c_int64(n),
                                  C.byref(maps[b]), C.c_void_p(streams[b].cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


run(1, 3)
res = {"shape": shape, "one_stream_ms_per_launch": round(run(1), 4), "two_streams_ms_per_launch": round(run(2), 4),
       "one_stream_again": round(run(1), 4), "two_streams_again": round(run(2), 4)}
same = bool(torch.equal(outs[0], outs[1]) or ((outs[0] == outs[1]) | (outs[0].isnan() & outs[1].isnan())).all())
res["maps_equal"] = same
print(json.dumps(res))
