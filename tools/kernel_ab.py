"""Kernel time and map digest of one L-BFGS-B fit configuration with whatever library T2FIT_LIB selects, through the
raw C ABI (so that an older build of the library can be measured beside the current one in the same call):

    python tools/kernel_ab.py [label] [--fit rician] [--shape 180 256 256] [--nte 6] [--no_prior] [--legacy] [--extras]

Prints the mean / min kernel time of 10 launches after 3 warm-ups (the library's own HIP events) and the SHA-256 of the
four maps: a restructured kernel must leave the digest as it was."""
import argparse
import ctypes as C
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from fetal_t2mapping_amd import _abi, synth  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("label", nargs="?", default="lib")
p.add_argument("--fit", default="gaussian_rician")
p.add_argument("--shape", nargs=3, type=int, default=[256, 256, 256])
p.add_argument("--nte", type=int, default=8)
p.add_argument("--no_prior", action="store_true")
p.add_argument("--legacy", action="store_true")
p.add_argument("--extras", action="store_true")
p.add_argument("--reps", type=int, default=10)
p.add_argument("--solver", default="lbfgsb", choices=["lbfgsb", "lm", "loglin"])
p.add_argument("--precision", default="f64", choices=["f64", "f32"])
a = p.parse_args()

lib = C.CDLL(os.environ.get("T2FIT_LIB") or os.path.join(REPO, "fetal_t2mapping_amd", "lib", "libt2fit_hip.so"))
lib.t2fit_last_kernel_ms.restype = C.c_double
dev = torch.device("cuda", 0)
shape = tuple(a.shape)
n = shape[0] * shape[1] * shape[2]
e, m, te = synth.brain_volume_torch(shape, a.nte, synth.SEED_BASE + 3, dev)
cfg = _abi.T2FitConfig()
assert lib.t2fit_config_default(C.byref(cfg), _abi.MODELS[a.fit], 1) == 0
cfg.n_te = a.nte
cfg.no_prior = int(a.no_prior)
cfg.numpy_legacy = int(a.legacy)
cfg.solver = _abi.SOLVERS[a.solver]
cfg.precision = _abi.PRECISIONS[a.precision]
if a.solver == "lm":
    cfg.maxiter = 0
for i, t in enumerate(te):
    cfg.te_ms[i] = float(t)
out = torch.zeros((4, n), dtype=torch.float32, device=dev)
maps = _abi.T2FitMaps()
maps.t2, maps.k, maps.sigma, maps.res = (out[j].data_ptr() for j in range(4))
if a.extras:
    nit = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    maps.nit, maps.status = nit.data_ptr(), st.data_ptr()
lib.t2fit_set_timing(1)
ks = []
for i in range(3 + a.reps):
    rc = lib.t2fit_volume_dev(C.byref(cfg), C.c_void_p(e.data_ptr()), 0, C.c_void_p(m.data_ptr()), C.c_int64(n), C.byref(maps), None)
    assert rc == 0, rc
    k = lib.t2fit_last_kernel_ms()
    if i >= 3:
        ks.append(k)
torch.cuda.synchronize()
h = hashlib.sha256(out.cpu().numpy().tobytes())
if a.extras:
    h.update(nit.cpu().numpy().tobytes())
    h.update(st.cpu().numpy().tobytes())
print(f"{a.label}: {a.solver}{'/' + a.precision if a.solver == 'lm' else ''} {a.fit} {'noprior' if a.no_prior else 'prior'}{' legacy' if a.legacy else ''} {shape[0]}x{shape[1]}x{shape[2]}x{a.nte}"
      f" kernel_ms mean {sum(ks) / len(ks):.3f} min {min(ks):.3f} maps sha256 {h.hexdigest()[:24]}", flush=True)
