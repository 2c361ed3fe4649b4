#!/usr/bin/env python3
"""Static attribution of a kernel's ISA to source lines / functions.

    hipcc -O3 --offload-arch=gfx950 -gline-tables-only -save-temps --cuda-device-only -c t2fit_kernels.hip
    python tools/isa_attrib.py <file.s> <kernel-name-substring> [--by-line]

Counts instructions per source file:line range (from .loc directives) and per class (f64 arithmetic, moves,
selects, SGPR spill traffic v_readlane/v_writelane, AGPR spill traffic v_accvgpr_*, scalar, branches, waits).
Static counts: loops are counted once.  Used to find where the non-arithmetic issue slots of the L-BFGS-B lane
kernel come from (profiles/r02_*_isa_mix.txt).
"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_readlane", "v_writelane")):
        return "sgpr_spill"
    if op.startswith("v_accvgpr"):
        return "agpr_spill"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_mov"):
        return "v_mov"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64")):
        return "f64_arith"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "f64_trans"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_"):
        return "valu_other"
    return "other"


def main():
    path, want = sys.argv[1], sys.argv[2]
    by_line = "--by-line" in sys.argv
    files = {}
    cur = None
    loc = ("?", 0)
    per_loc = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if want in m.group(1) else None
            continue
        if cur is None:
            m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
            if m:
                files[int(m.group(1))] = m.group(3) or m.group(2)
            continue
        s = line.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)).split("/")[-1], int(m.group(2)))
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        key = loc if by_line else (loc[0], loc[1] // 10 * 10)
        per_loc[key][c] += 1
        total[c] += 1
    n = sum(total.values())
    print(f"kernel *{want}*: {n} instructions")
    for k, v in total.most_common():
        print(f"  {k:12s} {v:6d} {100.0 * v / n:5.1f}%")
    print("per source location (file, line bucket): total | f64 | spill(sgpr) | spill(agpr) | mov+cnd+cmp | salu+branch")
    for key in sorted(per_loc):
        c = per_loc[key]
        t = sum(c.values())
        if t < 15:
            continue
        print(f"  {key[0]:22s} {key[1]:5d}  {t:5d} | {c['f64_arith'] + c['f64_trans']:5d} | {c['sgpr_spill']:4d} | {c['agpr_spill']:4d} | "
              f"{c['v_mov'] + c['v_cndmask'] + c['v_cmp']:5d} | {c['salu'] + c['branch']:5d}")


if __name__ == "__main__":
    main()
