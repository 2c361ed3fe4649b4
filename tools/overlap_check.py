"""Does a kernel of another stream get to run while the persistent reference-trajectory fit is in flight?
(the assumption behind bench.py --reserve-cus for N > 1: RCCL's all-gather kernel beside the next fit).  A
device-to-device copy kernel (torch copy_ of 1 GiB, stream B) is launched right after the fit (stream A); its
HIP-event start/end times are reported relative to the fit's, once per T2FIT_RESERVE_CUS value (own process each).

    python tools/overlap_check.py            # runs itself with T2FIT_RESERVE_CUS = 0 and 16
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
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
b = torch.empty_like(a)
sa, sb = torch.cuda.current_stream(), torch.cuda.Stream()


def fit():
    check(lib.t2fit_volume_dev(C.byref(cfg), e.data_ptr(), _abi.LAYOUT_TE_MAJOR, m.data_ptr(), n, C.byref(maps),
                               C.c_void_p(sa.cuda_stream)))


fit(); b.copy_(a); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record(sa); fit(); ev[1].record(sa)
with torch.cuda.stream(sb):
    ev[2].record(sb); b.copy_(a); ev[3].record(sb)
torch.cuda.synchronize()
print(json.dumps({"reserve_cus": int(os.environ.get("T2FIT_RESERVE_CUS", "0")), "fit_ms": round(ev[0].elapsed_time(ev[1]), 2),
                  "copy_kernel_start_ms_after_fit_start": round(ev[0].elapsed_time(ev[2]), 2),
                  "copy_kernel_end_ms_after_fit_start": round(ev[0].elapsed_time(ev[3]), 2)}))
