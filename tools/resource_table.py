#!/usr/bin/env python3
"""Per-kernel register / spill table from a `hipcc ... -Rpass-analysis=kernel-resource-usage` log.

    python tools/resource_table.py build.log [substring]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    seen = set()
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = b.split(" [")[0].strip()
        if name in seen:
            continue
        seen.add(name)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").split("(t2fit::LaneParams")[0].replace("void ", "")
        if want not in dem:
            continue

        def g(key):
            m = re.search(key + r": (\d+)", b)
            return m.group(1) if m else "?"

        print(f"{dem[:92]:92s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} SGPR {g('SGPRs'):>3} sgprSpill {g('SGPRs Spill'):>3} "
              f"vgprSpill {g('VGPRs Spill'):>3} scratch {g('ScratchSize .bytes/lane.'):>4} waves/SIMD {g('Occupancy .waves/SIMD.')}"
              f" LDS {g('LDS Size .bytes/block.')}")


if __name__ == "__main__":
    main()
