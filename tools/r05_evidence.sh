#!/bin/bash
# tools/r05_evidence.sh OUTDIR -- on the GPU box: every number DESIGN.md quotes for round 5, one command each: the profile of the
# default bench command (tools/profile.sh: kernel stats, HBM traffic, SQ counters, instruction mix, the bench line with its
# side configurations), the stage cycles of the unit kernel (diagnostic build), the memory-side model (tools/membench) and
# the bench lines of the other streams.
OUT=${1:-gpurun_out/r05_final}
mkdir -p $OUT
bash tools/profile.sh $OUT > $OUT/profile_log.txt 2>&1 || { echo "profile.sh failed"; tail -5 $OUT/profile_log.txt; }
L=$PWD/lidar_feature_extraction_amd/_lib
if [ -f $L/liblfx_stamps.so ]; then
  LFX_LIB_PATH=$L/liblfx_stamps.so timeout -k 10 200 python tools/stamps.py --skew --by-ring --json $OUT/stage_cycles.json > $OUT/stage_cycles.txt 2>&1 || echo "stamps failed"
fi
[ -x tools/membench/membench ] && timeout -k 10 300 tools/membench/membench > $OUT/membench.txt 2>&1
run() { name=$1; shift; timeout -k 10 400 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; }; echo "$name: $(python3 -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print(d.get('value'), d.get('value_min'), d.get('value_max'), d['roofline']['frac'], d['roofline']['whole_path_frac'], d.get('parity_spot_check'), d.get('gather_ms_per_step'), (d.get('box') or {}).get('copy_gbs'))
except Exception as e: print('unreadable', e)
")"; }
B="python bench.py --no-cpu-baseline"
run bench_launch_yaml $B --params launch_yaml
run bench_rotated $B --start-col 517
run bench_reversed $B --reverse
run bench_ragged_5pct $B --drop-fraction 0.05
run bench_zeros_5pct $B --drop-fraction 0.05 --drop-zero
run bench_shuffled $B --shuffle --batch 256 --steps 5 --warmup 2
run bench_force_gather_pairs $B --force-gather
run bench_force_gather_dst0 $B --force-gather --gather-dst 0
run bench_force_gather_lanes2 $B --force-gather --gather-lanes 2
run bench_64x3600_long_form $B --cols 3600 --batch 512
run bench_64x4500 $B --cols 4500 --batch 256
run bench_cfg3_128x2048x32_long $B --rings 128 --cols 2048 --batch 32 --steps 200 --warmup 20
rm -f $OUT/*.err
