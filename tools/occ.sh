#!/bin/bash
# tools/occ.sh -- on the GPU box: unit-kernel time against workgroups per CU (LFX_DEBUG_UNIT_LDS_PAD adds unused LDS)
mkdir -p gpurun_out
for pad in 0 4500 10000 18000 32000; do
  LFX_DEBUG_UNIT_LDS_PAD=$pad timeout -k 10 180 python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/occ_tmp.json 2> gpurun_out/occ_err.txt || { tail -3 gpurun_out/occ_err.txt; exit 1; }
  python - $pad <<'PY' | tee -a gpurun_out/occ.txt
import json, sys
d = json.loads(open("gpurun_out/occ_tmp.json").read().strip().splitlines()[-1])
print("lds pad", sys.argv[1], "unit_us", d["roofline"]["kernel_us_per_launch"]["ring_unit_kernel"])
PY
done
