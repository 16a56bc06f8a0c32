#!/bin/bash
# tools/localize_check.sh [outdir] -- on the GPU box: the localizer's tests, its timing for one scan and for 64, the timeline of one call
OUT=${1:-gpurun_out/loc}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_align_gpu.py tests/test_downsample_gpu.py tests/test_residuals_gpu.py tests/test_map_gpu.py tests/test_cpp_host.py -x -q -m gpu > $OUT/tests.txt 2>&1 || { tail -30 $OUT/tests.txt; exit 1; }
tail -1 $OUT/tests.txt
timeout -k 10 200 python tools/localize_bench.py --batch 1 --map-scans 40 --steps 20 --cpu-scans 0 --kd-scans 0 > $OUT/batch1.json 2> $OUT/batch1.err || { tail $OUT/batch1.err; exit 1; }
timeout -k 10 200 python tools/localize_bench.py --batch 64 --map-scans 40 --steps 5 --cpu-scans 0 --kd-scans 0 > $OUT/batch64.json 2> $OUT/batch64.err || { tail $OUT/batch64.err; exit 1; }
python - $OUT <<'PY'
import json, sys
for n in ("batch1", "batch64"):
    d = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
    print(n, "ms/batch", d["localize_ms_per_batch"], "ms/scan", d["localize_ms_per_scan"], "iterations", d["iterations_mean"], "codes", d["codes"], "pose error", d["pose_error_after"])
PY
bash tools/trace_localize.sh | grep -v rocprofv3
