#!/bin/bash
# tools/holes_subbatch.sh -- on the GPU box: a stream with holes (5 % of the returns written as (0, 0, 0) and filtered, or
# missing) through the bucketing route in batches small enough for the staged arrays of one batch (16 B per point) to stay
# in the 256 MB Infinity Cache between the bucketing kernel and the unit kernel, one and two streams (the second
# context's bucketing pass under the first one's unit kernel).  Prints scans/s per (batch, streams).
mkdir -p gpurun_out
OUT=gpurun_out/holes_subbatch.txt
: > $OUT
for mode in "--drop-zero" ""; do
  for b in 1024 256 128 64 48 32; do
    for st in 1 2 3; do
      steps=$(( 8192 / b )); [ $steps -lt 8 ] && steps=8
      timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --unique 8 --steps $steps --warmup 4 --repeats 3 --streams $st \
        --drop-fraction 0.05 $mode > gpurun_out/holes_tmp.json 2> gpurun_out/holes_err.txt || { echo "batch $b streams $st failed"; tail -3 gpurun_out/holes_err.txt; continue; }
      python - "$mode" $b $st <<'PY' | tee -a $OUT
import json, sys
d = json.loads(open("gpurun_out/holes_tmp.json").read().strip().splitlines()[-1])
k = d["roofline"]["kernel_us_per_launch"]
print("%-12s batch %5s streams %s  %9.0f scans/s  %s  parity %s" % (sys.argv[1] or "(missing)", sys.argv[2], sys.argv[3], d["value"],
      " ".join("%s=%.0f" % (n.replace("_kernel", "").replace("ring_", ""), v) for n, v in k.items() if v > 8), d["parity_spot_check"]))
PY
    done
  done
done
