#!/bin/bash
# tools/occ8.sh -- on the GPU box: what the EIGHTH workgroup per CU is worth to this kernel family, measured where it is
# admitted: rings of 1 400 columns take the 4-chunk variant of ring_unit_org_kernel (18 048 B of LDS, <= 64 vector and 78
# scalar registers: 8 workgroups per CU); LFX_DEBUG_UNIT_LDS_PAD then takes workgroups away again (2 600 B -> 7, 5 400 -> 6,
# 9 300 -> 5).  Same instruction stream, same data, only the waves in flight differ.
mkdir -p gpurun_out
: > gpurun_out/occ8.txt
for round in 1 2 3; do
  for pad in 0 2600 5400 9300; do
    LFX_DEBUG_UNIT_LDS_PAD=$pad timeout -k 10 180 python bench.py --no-cpu-baseline --cols 1400 --steps 30 --warmup 5 > gpurun_out/occ_tmp.json 2> gpurun_out/occ_err.txt || { tail -3 gpurun_out/occ_err.txt; exit 1; }
    python - $pad <<'PY' | tee -a gpurun_out/occ8.txt
import json, sys
d = json.loads(open("gpurun_out/occ_tmp.json").read().strip().splitlines()[-1])
pad = int(sys.argv[1])
wgs = (160 * 1024) // (18048 + pad)
print("lds pad %5d  -> %d workgroups per CU   unit kernel %7.1f us   %8.0f scans/s   parity %s" % (pad, min(wgs, 8), d["roofline"]["kernel_us_per_launch"]["ring_unit_org_kernel"], d["value"], d["parity_spot_check"]))
PY
  done
done
