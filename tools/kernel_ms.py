"""Kernel time of the headline fit (256^3 x 8 TE, gaussian_rician, L-BFGS-B) with whatever library T2FIT_LIB selects,
through the raw C ABI (works with libraries of ABI 2 and 3): mean of 10 launches after 3 warm-ups."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from fetal_t2mapping_amd import _abi, synth  # noqa: E402

lib = C.CDLL(os.environ.get("T2FIT_LIB") or os.path.join(REPO, "fetal_t2mapping_amd", "lib", "libt2fit_hip.so"))
lib.t2fit_last_kernel_ms.restype = C.c_double
dev = torch.device("cuda", 0)
n = 256 ** 3
e, m, te = synth.brain_volume_torch((256, 256, 256), 8, synth.SEED_BASE + 3, dev)
cfg = _abi.T2FitConfig()
assert lib.t2fit_config_default(C.byref(cfg), 1, 1) == 0
cfg.n_te = 8
for i, t in enumerate(te):
    cfg.te_ms[i] = float(t)
out = torch.empty((4, n), dtype=torch.float32, device=dev)
maps = _abi.T2FitMaps()
maps.t2, maps.k, maps.sigma, maps.res = (out[j].data_ptr() for j in range(4))
lib.t2fit_set_timing(1)
ks = []
for i in range(13):
    rc = lib.t2fit_volume_dev(C.byref(cfg), C.c_void_p(e.data_ptr()), 0, C.c_void_p(m.data_ptr()), C.c_int64(n), C.byref(maps), None)
    assert rc == 0, rc
    k = lib.t2fit_last_kernel_ms()
    if i >= 3:
        ks.append(k)
torch.cuda.synchronize()
print(sys.argv[1] if len(sys.argv) > 1 else "lib", "kernel_ms mean %.3f min %.3f" % (sum(ks) / len(ks), min(ks)),
      "t2 checksum %.6f" % float(out[0].double().sum()))
