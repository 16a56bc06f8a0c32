#!/bin/bash
# tools/ab_run.sh [ROUNDS] [bench args...] -- on the GPU box: alternate _lib/ab_A.so (built by tools/ab.sh)
# and _lib/liblfx.so through bench.py, print scans/s and the per-kernel microseconds of each run.
R=${1:-3}; shift
mkdir -p gpurun_out
for i in $(seq $R); do
  for lib in ab_A.so liblfx.so; do
    LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/$lib timeout -k 10 180 python bench.py --no-cpu-baseline --steps 40 --warmup 6 "$@" \
      > gpurun_out/ab_tmp.json 2> gpurun_out/ab_err.txt || { tail -5 gpurun_out/ab_err.txt; exit 1; }
    python - $lib <<'PY' | tee -a gpurun_out/ab.txt
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
k = d["roofline"]["kernel_us_per_launch"]
print(sys.argv[1], d["value"], d["ms_per_step"], " ".join("%s=%.0f" % (n.replace("_kernel", "").replace("ring_", ""), v) for n, v in k.items() if v > 0))
PY
  done
done
