#!/bin/bash
# tools/ldsgran.sh -- on the GPU box: what a few hundred bytes of LDS more per workgroup cost the 5-chunk organised-scan kernel
# (LFX_DEBUG_UNIT_LDS_PAD 0 / 1024 / 1200 / 1300 / 1700 on the headline configuration).  Round 4: +1.2 % for every non-zero pad --
# the seventh workgroup per CU is lost with the first kilobyte (allocation granularity) and is worth that much to this variant.
mkdir -p gpurun_out
: > gpurun_out/ldsgran.txt
for round in 1 2; do
  for pad in 0 1024 1200 1300 1700; do
    LFX_DEBUG_UNIT_LDS_PAD=$pad timeout -k 10 180 python bench.py --no-cpu-baseline --no-side-configs --steps 30 --warmup 5 > gpurun_out/g_tmp.json 2> gpurun_out/g_err.txt || { tail -3 gpurun_out/g_err.txt; exit 1; }
    python - $pad <<'PY' | tee -a gpurun_out/ldsgran.txt
import json, sys
d = json.loads(open("gpurun_out/g_tmp.json").read().strip().splitlines()[-1])
print("pad %5s unit %7.1f us  %8.0f scans/s" % (sys.argv[1], d["roofline"]["kernel_us_per_launch"]["ring_unit_org_kernel"], d["value"]))
PY
  done
done
