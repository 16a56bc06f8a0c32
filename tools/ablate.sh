#!/bin/bash
# tools/ablate.sh -- on the GPU box, after `make -C lidar_feature_extraction_amd/csrc ablate` here: unit-kernel time
# with stages switched off (liblfx_ablate.so honours LFX_DEBUG_UNIT_FLAGS; results are
# then wrong on purpose): 1 edge pass, 64 surface pass, 256 no occlusion, 512 no parallel-beam, 1024 no records.
mkdir -p gpurun_out
for f in 65 64 1 0 321 577 1089 1856; do
  LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/liblfx_ablate.so LFX_DEBUG_UNIT_FLAGS=$f timeout -k 10 180 python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/abl_tmp.json 2> gpurun_out/abl_err.txt || { tail -3 gpurun_out/abl_err.txt; exit 1; }
  python - $f <<'PY' | tee -a gpurun_out/ablate.txt
import json, sys
d = json.loads(open("gpurun_out/abl_tmp.json").read().strip().splitlines()[-1])
print("flags", sys.argv[1], "unit_us", d["roofline"]["kernel_us_per_launch"]["ring_unit_kernel"], "parity", d["parity_spot_check"])
PY
done
