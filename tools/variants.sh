#!/bin/bash
# tools/variants.sh LIB... -- on the GPU box: bench each _lib/<LIB> twice, print scans/s and per-kernel microseconds
mkdir -p gpurun_out
for i in 1 2; do
  for lib in "$@"; do
    LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/$lib timeout -k 10 180 python bench.py --no-cpu-baseline --steps 40 --warmup 6 \
      > gpurun_out/var_tmp.json 2> gpurun_out/var_err.txt || { tail -5 gpurun_out/var_err.txt; exit 1; }
    python - $lib <<'PY'
import json, sys
d = json.loads(open("gpurun_out/var_tmp.json").read().strip().splitlines()[-1])
k = d["roofline"]["kernel_us_per_launch"]
print(sys.argv[1], d["value"], d["ms_per_step"], d["parity_spot_check"], " ".join("%s=%.0f" % (n.replace("_kernel", "").replace("ring_", ""), v) for n, v in k.items() if v > 0))
PY
  done
done
