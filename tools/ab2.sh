#!/bin/bash
# tools/ab2.sh LIB [LIB ...] -- on the GPU box: time each library (lidar_feature_extraction_amd/_lib/LIB) through bench.py,
# ROUNDS times in turn on the one device; prints scans/s and the organised-scan kernel's microseconds per launch.
R=${ROUNDS:-2}
for i in $(seq $R); do
  for lib in "$@"; do
    LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 --warmup 6 $BENCH_ARGS 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_us_per_launch']; print(sys.argv[1], d['value'], ' '.join('%s=%.0f' % (n.replace('_kernel','').replace('ring_',''), v) for n, v in k.items() if v > 7))" $lib
  done
done
