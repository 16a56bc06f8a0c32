#!/bin/bash
# tools/r03_evidence.sh OUTDIR -- on the GPU box: every number DESIGN.md quotes for round 3, one command each.
OUT=${1:-gpurun_out/r03_final}
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 400 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; }; echo "$name: $(python3 -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print(d.get('value'), d.get('value_min'), d.get('value_max'), d['roofline']['frac'], d['roofline']['whole_path_frac'], d.get('parity_spot_check'))
except Exception as e: print('unreadable', e)
")"; }
B="python bench.py --no-cpu-baseline"
run bench_cfg0_16x900 $B --rings 16 --cols 900
run bench_cfg1_16x1800 $B --rings 16 --cols 1800
run bench_cfg3_128x2048x32 $B --rings 128 --cols 2048 --batch 32 --steps 200 --warmup 20
run bench_rotated $B --start-col 517
run bench_reversed $B --reverse
run bench_ragged_5pct $B --drop-fraction 0.05
run bench_zeros_5pct $B --drop-fraction 0.05 --drop-zero
run bench_shuffled $B --shuffle --batch 256 --steps 5 --warmup 2
run bench_streams2 $B --streams 2
LFX_DEBUG_STREAM=1 run bench_stream_kernel $B
run bench_force_gather $B --force-gather
run bench_force_gather_rotate $B --force-gather --gather-dst rotate
timeout -k 10 300 python tools/localize_bench.py --batch 1 --map-scans 40 --steps 9 --cpu-scans 0 --kd-scans 2 > $OUT/localize_batch1.json 2> $OUT/localize_batch1.err; tail -c 600 $OUT/localize_batch1.json
timeout -k 10 300 python tools/localize_bench.py --batch 64 --map-scans 40 --steps 3 --cpu-scans 0 --kd-scans 0 > $OUT/localize_batch64.json 2> $OUT/localize_batch64.err; tail -c 300 $OUT/localize_batch64.json
