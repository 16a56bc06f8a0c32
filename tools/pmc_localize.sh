#!/bin/bash
# tools/pmc_localize.sh [LIB] -- on the GPU box: SQ counters (rocprofv3 --pmc with --kernel-trace only) and durations of the
# scan_to_map / align_* kernels for one iteration of tools/localize_bench.py; LIB: a library under _lib/ (A/B builds)
export TMPDIR=/tmp
OUT=gpurun_out/locpmc
mkdir -p $OUT
LIB=${1:-liblfx}
export LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/$LIB.so
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/localize_bench.py --batch ${BATCH:-1} --cpu-scans 0 --kd-scans 0 --max-iter 1 --steps 2 > $OUT/run1.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "scan_to_map" not in k and "align_" not in k and "map_search" not in k and "rows_from" not in k and "voxel" not in k: continue
        k = k[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "scan_to_map" not in k and "align_" not in k and "map_search" not in k and "rows_from" not in k and "voxel" not in k: continue
        dur[k[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in acc:
    print(k)
    print("  durations us:", [round(d) for d in dur[k]])
    for c in acc[k]:
        print("   %-22s %14.0f per launch" % (c, acc[k][c] / cnt[k][c]))
PY
rm -rf $OUT/p1
