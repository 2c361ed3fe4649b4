"""Turn the per-kernel PMC summaries of a profile round (gpurun_out/pmc_<tag>_{lbfgsb,lmf32,loglin}.txt, written by
tools/profile_r02.sh) into profiles/traffic.json (HBM bytes per launch) and profiles/instr_mix.json (VALU instruction
counts per launch), which bench.py reads for `roofline.traffic` and the `alu` view.

    python tools/make_profile_json.py <tag>
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
KEYS = {"lbfgsb": "gaussian_rician/lbfgsb/f64/256x256x256x8", "rician": "rician/lbfgsb/f64/180x256x256x6", "lmf32": "gaussian_rician/lm/f32/256x256x256x8",
        "lmf64": "gaussian_rician/lm/f64/256x256x256x8", "loglin": "gaussian/loglin/f64/256x256x256x8"}


def read(name):
    out = {}
    for line in open(os.path.join(REPO, "gpurun_out", f"pmc_{tag}_{name}.txt")):
        parts = line.split()
        c = [p for p in parts if p.startswith(("SQ", "FETCH", "WRITE", "TCC"))]
        m = [p for p in parts if p.startswith("mean=")]
        if c and m:
            out[c[0]] = float(m[0][5:])
    return out


res = read("residuals")
# FETCH_SIZE calibration for one-dword-per-lane reads, on the kernel whose read volume is known exactly
# (residuals_kernel at 256^3 x 8 TE, mask fill 7,463,192: samples + mask + three maps of the fitted voxels)
n, m = 256 ** 3, 7463192
known_read = m * 32 + n + 3 * 4 * m
cal4 = known_read / (res["FETCH_SIZE"] * 1024)
traffic = {"_comment": "HBM bytes per launch of the dominant kernel, 1 x MI355X, from separate rocprofv3 --pmc FETCH_SIZE / "
                       "WRITE_SIZE passes (profiles/%s_pmc_*.txt).  WRITE_SIZE is exact.  FETCH_SIZE under-reports on gfx950: "
                       "one-dword-per-lane reads are calibrated on residuals_kernel, whose read volume is known exactly "
                       "(fetch_calibration); 16-byte-per-lane reads report exactly one half (MI355X_MICROARCH.md)." % tag,
           "fetch_calibration": round(cal4, 4)}
mix = {"_comment": "VALU wave-instructions per launch by class (rocprofv3 --pmc, own passes), 256^3 x 8 TE, mask fill 0.44; "
                   "lanes_active = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)."}
for name, key in KEYS.items():
    if not os.path.exists(os.path.join(REPO, "gpurun_out", f"pmc_{tag}_{name}.txt")):
        continue
    c = read(name)
    if name == "loglin":  # echo planes read 16 B per lane (reported 1/2), the mask 4 B per lane
        mask_reported = n / cal4
        fetch = (c["FETCH_SIZE"] * 1024 - mask_reported) * 2 + n
    else:
        fetch = c["FETCH_SIZE"] * 1024 * cal4
    traffic[key] = int(fetch + c["WRITE_SIZE"] * 1024)
    mix[key] = {k[len("SQ_INSTS_"):].lower(): int(v) for k, v in c.items() if k.startswith("SQ_INSTS_")}
    mix[key]["lanes_active"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4)
json.dump(traffic, open(os.path.join(REPO, "profiles", "traffic.json"), "w"), indent=1)
json.dump(mix, open(os.path.join(REPO, "profiles", "instr_mix.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(json.dumps({k: (v if isinstance(v, str) else {a: v[a] for a in ("valu", "valu_fma_f64", "lanes_active") if a in v}) for k, v in mix.items()}, indent=1))
