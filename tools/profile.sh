#!/bin/bash
# tools/profile.sh OUTDIR -- on the GPU box: the evidence behind bench.py's "roofline" object.
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (per-kernel durations)       -> kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (HBM traffic)      -> pmc_traffic.json
#   3. the plain bench line, CPU baseline included                                         -> bench.json
set -e
OUT=${1:-gpurun_out/profile}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/trace_run.txt 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/pmc_fetch_run.txt 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $OUT/pmc_write_run.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections, re
out = sys.argv[1]
def mean_per_kernel(d, counter):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            m = re.search(r"lfx::(\w+)", name)
            key = m.group(1) if m else name[:40]
            if key == "ring_unit_kernel" and "<true" in name:
                key = "ring_unit_kernel(second pass)"
            acc[key] += float(row["Counter_Value"]); cnt[key] += 1
    return {k: round(acc[k] / cnt[k], 1) for k in acc}
fetch = mean_per_kernel(out + "/pmc_fetch", "FETCH_SIZE")
write = mean_per_kernel(out + "/pmc_write", "WRITE_SIZE")
hbm = {k: int(2 * fetch.get(k, 0) * 1024 + write.get(k, 0) * 1024) for k in set(fetch) | set(write)}
json.dump({"batch": 1024, "rings": 64, "cols": 1800,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB per dispatch, mean over dispatches); "
                   "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md: it counts 128-B requests at 64 B)",
           "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "hbm_bytes_per_launch": hbm}, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(hbm))
PY
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench_err.txt
cat $OUT/bench.json
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
