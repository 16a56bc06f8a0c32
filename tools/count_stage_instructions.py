"""Static instruction counts per stage of a unit kernel, from the marked listing (`make -C lidar_feature_extraction_amd/csrc marks`):
    python tools/count_stage_instructions.py [kernel-name-substring]
Stages are the LFX_STAMP points of unit_body (0 entry, 1 geometry, 2 loads, 3 range, 4 order check + links + jumps,
5 occlusion + reach, 6 curvature, 7 order masks, 8 edge pass, 9 surface pass, 10 parallel beam + labels + records).
Counts are of the listing (every chunk of an unrolled stage once, a loop body once), not of executed instructions."""
import collections
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else "ring_unit_org_kernelILi0ELi5ELb0"
text = open(os.path.join(root, "lidar_feature_extraction_amd", "_build", "lfx_unit_v0_gfx950_marks.s")).read().splitlines()
inside, stage = False, "pre"
cnt = collections.defaultdict(collections.Counter)
for line in text:
    if not inside:
        if re.match(r"^_ZN3lfx\w*%s\w*:" % re.escape(want), line):
            inside = True
        continue
    if line.startswith(".Lfunc_end"):           # (not the first s_endpgm: an early exit may sit anywhere in the listing)
        break
    m = re.search(r"; LFX_MARK (\d+)", line)
    if m:
        stage = "after %s" % m.group(1)
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if not m:
        continue
    op = m.group(1)
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_waitcnt", "s_nop", "s_cbranch", "s_branch"))
            else "lds" if op.startswith("ds_") else "branch" if op.startswith(("s_cbranch", "s_branch")) else "vmem" if op.startswith(("global_", "flat_", "buffer_"))
            else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "smem")
    cnt[stage][kind] += 1
    if kind == "valu":
        cnt[stage]["f64" if "f64" in op else "v32"] += 1
keys = ["valu", "f64", "salu", "lds", "branch", "vmem", "smem", "wait"]
print("%-10s" % "stage" + "".join("%8s" % k for k in keys))
tot = collections.Counter()
for st in sorted(cnt, key=lambda s: (-1 if s == "pre" else int(s.split()[1]))):
    print("%-10s" % st + "".join("%8d" % cnt[st][k] for k in keys))
    tot.update(cnt[st])
print("%-10s" % "total" + "".join("%8d" % tot[k] for k in keys))
