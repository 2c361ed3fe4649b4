"""Does RCCL's all-gather get to run while the persistent reference-trajectory fit is in flight?  (VERDICT r01 item 8:
the assumption behind bench.py --reserve-cus.)  One rank, one GPU: `all_gather_into_tensor` of the packed maps of a
256^3 volume (268 MB) is started on RCCL's stream right after a fit is launched on the compute stream; HIP events
bracket both.  Reported once per T2FIT_RESERVE_CUS value (own process each), with a rocprofv3-free hint of whether
RCCL ran a kernel at all for world size 1 (it may use a plain device copy).

    python tools/overlap_check_rccl.py       # runs itself with T2FIT_RESERVE_CUS = 0 and 16
"""
import ctypes as C
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if len(sys.argv) == 1:
    for r in ("0", "16"):
        env = dict(os.environ, T2FIT_RESERVE_CUS=r, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
    sys.exit(0)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import fetal_t2mapping_amd as t2  # noqa: E402
from fetal_t2mapping_amd import _abi, synth  # noqa: E402
from fetal_t2mapping_amd._lib import check, require_gpu  # noqa: E402

lib = require_gpu()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
shape = (256, 256, 256)
n = 256 ** 3
e, m, te = synth.brain_volume_torch(shape, 8, synth.SEED_BASE + 3, dev)
cfg = t2.make_config("gaussian_rician", t2.fit_table("gaussian_rician", True), te)
out = torch.empty((4, n), dtype=torch.float32, device=dev)
prev = torch.randn((4, n), dtype=torch.float32, device=dev)  # "the maps of the previous step"
gathered = torch.empty((1, 4, n), dtype=torch.float32, device=dev)
maps = _abi.T2FitMaps()
maps.t2, maps.k, maps.sigma, maps.res = (out[j].data_ptr() for j in range(4))
sa = torch.cuda.current_stream()


def fit():
    check(lib.t2fit_volume_dev(C.byref(cfg), e.data_ptr(), _abi.LAYOUT_TE_MAJOR, m.data_ptr(), n, C.byref(maps),
                               C.c_void_p(sa.cuda_stream)))


fit()
dist.all_gather_into_tensor(gathered.view(-1), prev.view(-1))
torch.cuda.synchronize()
# the gather alone
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record(sa)
w = dist.all_gather_into_tensor(gathered.view(-1), prev.view(-1), async_op=True)
w.wait()
t1.record(sa)
torch.cuda.synchronize()
alone = t0.elapsed_time(t1)
# the gather started right behind the launch of a fit
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
side = torch.cuda.Stream()
ev[0].record(sa)
fit()
ev[1].record(sa)
# The collective is issued with `side` as the current stream: torch makes RCCL's stream wait for the CURRENT stream
# (that is how bench.py's gather of step i waits for fit i); issued from the compute stream it would simply queue
# behind the fit just launched.  Here the gather's input does not depend on that fit, as in bench.py where
# gather i runs beside fit i+1.
with torch.cuda.stream(side):
    w = dist.all_gather_into_tensor(gathered.view(-1), prev.view(-1), async_op=True)  # RCCL's own stream
    w.wait()          # stream-side wait on `side`, not on the compute stream
    ev[2].record(side)
torch.cuda.synchronize()
ok = bool(torch.equal(gathered[0], prev))
print(json.dumps({"reserve_cus": int(os.environ.get("T2FIT_RESERVE_CUS", "0")), "world_size": 1,
                  "fit_ms": round(ev[0].elapsed_time(ev[1]), 2), "gather_alone_ms": round(alone, 3),
                  "gather_done_ms_after_fit_launch": round(ev[0].elapsed_time(ev[2]), 2),
                  "gather_overlapped_the_fit": ev[0].elapsed_time(ev[2]) < 0.8 * ev[0].elapsed_time(ev[1]),
                  "gathered_equals_input": ok}))
dist.destroy_process_group()
