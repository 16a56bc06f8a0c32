#!/usr/bin/env python3
"""tools/kernel_resources.py [ASM|all] [substring] -- registers, spills, LDS and occupancy of every kernel in the device
assembly (lidar_feature_extraction_amd/_build/lfx_api_gfx950.s and the four lfx_unit_v*_gfx950.s; `make -C
lidar_feature_extraction_amd/csrc asmfile`)."""
import re
import subprocess
import sys

import glob
path = sys.argv[1] if len(sys.argv) > 1 else "all"
want = sys.argv[2] if len(sys.argv) > 2 else ""
paths = [path] if path != "all" else ["lidar_feature_extraction_amd/_build/lfx_api_gfx950.s"] + sorted(glob.glob("lidar_feature_extraction_amd/_build/lfx_unit_v?_gfx950.s"))
text = "\n".join(open(p).read() for p in paths)
rows = []
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: +\d+", text, re.S):
    blk = m.group(0)
    def f(k):
        mm = re.search(r"\.%s: +(\S+)" % k, blk)
        return mm.group(1) if mm else "?"
    name = f("name")
    rows.append((name, f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_count"), f("sgpr_spill_count"),
                 f("group_segment_fixed_size"), f("private_segment_fixed_size")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-90s %5s %5s %5s %5s %7s %7s" % ("kernel", "vgpr", "vspil", "sgpr", "sspil", "lds", "scratch"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("void lfx::", ""))
    if want in n:
        print("%-90s %5s %5s %5s %5s %7s %7s" % ((n[:90],) + r[1:]))
