#!/bin/bash
# tools/r04_evidence.sh OUTDIR -- on the GPU box: every number DESIGN.md quotes for round 4 that the default bench line (with
# its "configs" list: BASELINE.json's other shapes and the zero-filled stream) does not already carry, one command each.
OUT=${1:-gpurun_out/r04_final}
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 400 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; }; echo "$name: $(python3 -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print(d.get('value'), d.get('value_min'), d.get('value_max'), d['roofline']['frac'], d['roofline']['whole_path_frac'], d.get('parity_spot_check'), d.get('gather_ms_per_step'))
except Exception as e: print('unreadable', e)
")"; }
B="python bench.py --no-cpu-baseline"
run bench_rotated $B --start-col 517
run bench_reversed $B --reverse
run bench_ragged_5pct $B --drop-fraction 0.05
run bench_shuffled $B --shuffle --batch 256 --steps 5 --warmup 2
run bench_streams2 $B --streams 2
run bench_force_gather $B --force-gather --gather-dst 0
run bench_force_gather_rotate $B --force-gather --gather-dst rotate
run bench_64x3600_long_form $B --cols 3600 --batch 512
run bench_cfg3_128x2048x32_long $B --rings 128 --cols 2048 --batch 32 --steps 200 --warmup 20
timeout -k 10 600 python tools/stress.py 12000 404 > $OUT/stress_12000_cases.txt 2>&1; tail -1 $OUT/stress_12000_cases.txt
