#!/bin/bash
# tools/r06_evidence.sh OUTDIR -- on the GPU box: every number DESIGN.md quotes for round 6, one command each: the profile of the
# default bench command (tools/profile.sh: kernel stats, HBM traffic, SQ counters, the bench line with its side
# configurations), the same for the stream with zero records (the holes form), the stage cycles of both forms of the unit
# kernel (diagnostic build) and the bench lines of the other streams.
OUT=${1:-gpurun_out/r06_final}
mkdir -p $OUT
bash tools/profile.sh $OUT > $OUT/profile_log.txt 2>&1 || { echo "profile.sh failed"; tail -5 $OUT/profile_log.txt; }
bash tools/profile.sh $OUT/zeros --drop-zero --drop-fraction 0.05 --steps 8 --warmup 3 > $OUT/profile_zeros_log.txt 2>&1 || { echo "profile.sh (zeros) failed"; tail -5 $OUT/profile_zeros_log.txt; }
L=$PWD/lidar_feature_extraction_amd/_lib
if [ -f $L/liblfx_stamps.so ]; then
  LFX_LIB_PATH=$L/liblfx_stamps.so timeout -k 10 200 python tools/stamps.py --json $OUT/stage_cycles.json > $OUT/stamps_plain.txt 2>&1 || echo "stamps failed"
  LFX_LIB_PATH=$L/liblfx_stamps.so timeout -k 10 200 python tools/stamps.py --holes 0.05 > $OUT/stamps_holes.txt 2>&1 || echo "stamps (holes) failed"
fi
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
run bench_zeros_15pct $B --drop-fraction 0.15 --drop-zero
run bench_zeros_1pct $B --drop-fraction 0.01 --drop-zero
run bench_shuffled $B --shuffle --batch 256 --steps 5 --warmup 2
run bench_force_gather_pairs $B --force-gather
run bench_force_gather_dst0 $B --force-gather --gather-dst 0
run bench_cfg3_128x2048x32_long $B --rings 128 --cols 2048 --batch 32 --steps 200 --warmup 20
rm -f $OUT/*.err
