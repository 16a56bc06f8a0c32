#!/bin/bash
# tools/pmc2.sh LIB "COUNTER COUNTER ..." [kernel-substring] -- on the GPU box: one rocprofv3 --pmc pass (kernel trace only
# beside it) over a short bench run with library LIB; prints the per-wave average of each counter for the matching kernel.
LIB=$1; C="$2"; KN="${3:-ring_unit_org_kernel}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc2_$$
LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/$LIB timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pmc2_run.txt 2>&1
python3 - "$OUT" "$KN" "$LIB" <<'PY'
import csv, glob, sys, collections
out, kn, lib = sys.argv[1:4]
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kn in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
w = acc.get("SQ_WAVES", 0) / max(cnt.get("SQ_WAVES", 1), 1) or 1
print(lib, " ".join("%s=%.0f" % (k.replace("SQ_", ""), acc[k] / cnt[k] / (1 if k in ("SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES") else w)) for k in sorted(acc)))
PY
rm -rf $OUT
