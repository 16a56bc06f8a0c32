#!/bin/bash
# tools/trace_localize.sh -- on the GPU box: the kernel timeline of ONE lfx_localize_batch call (one 64 x 1800 scan)
export TMPDIR=/tmp
OUT=gpurun_out/loctrace
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 tools/localize_bench.py --batch 1 --map-scans 40 --steps 3 --cpu-scans 0 --kd-scans 0 > $OUT/run.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]))
for f in glob.glob(out + "/t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# the last call = the events after the last voxel_downsample_kernel start
last = max(i for i, e in enumerate(ev) if "voxel_downsample" in e[2])
t0 = ev[last][0]
prev_end = t0
for s, e, n in ev[last:last + 40]:
    print("%8.1f us  +%6.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n))
    prev_end = max(prev_end, e)
PY
tail -2 $OUT/run.txt
