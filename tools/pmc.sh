#!/bin/bash
# tools/pmc.sh "COUNTER COUNTER ..." [kernel-name-substring] -- on the GPU box: one rocprofv3 --pmc pass over a
# short bench run; prints the per-dispatch average of each counter for the matching kernel.
set -e
C="$1"; KN="${2:-ring_unit_kernel}"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$$
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pmc_run.txt 2>&1
python3 - "$OUT" "$KN" <<'PY'
import csv, glob, sys, collections
out, kn = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kn in row["Kernel_Name"] and "ring_unit_kernel<true" not in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
for k in sorted(acc):
    print("%-28s %14.0f per dispatch (%d dispatches)" % (k, acc[k] / cnt[k], cnt[k]))
PY
