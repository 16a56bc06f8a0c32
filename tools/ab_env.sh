#!/bin/bash
# tools/ab_env.sh ROUNDS "ENV=.. ENV=.." "ENV=.." ... -- [bench args after --]: on the GPU box, alternate bench.py runs under the
# given environments (one quoted string per arm; "" = none) and print scans/s and per-kernel microseconds of each run.
R=${1:-2}; shift
ARMS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARMS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for i in $(seq $R); do
  for arm in "${ARMS[@]}"; do
    env $arm timeout -k 10 240 python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_err.txt || { echo "arm [$arm] failed"; tail -5 gpurun_out/ab_err.txt; exit 1; }
    python - "$arm" <<'PY' | tee -a gpurun_out/ab_env.txt
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
k = d["roofline"]["kernel_us_per_launch"]
print("[%s]" % sys.argv[1], d["value"], d["ms_per_step"], d["parity_spot_check"], " ".join("%s=%.0f" % (n.replace("_kernel", "").replace("ring_", ""), v) for n, v in k.items() if v > 0))
PY
  done
done
