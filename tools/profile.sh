#!/bin/bash
# tools/profile.sh OUTDIR [bench args...] -- on the GPU box: the evidence behind bench.py's "roofline" object.
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (per-kernel durations)       -> kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (HBM traffic)      -> pmc_traffic.json
#   3. rocprofv3 --pmc SQ_* (instruction mix, busy and wait cycles of the dominant kernel) -> sq_counters.json
#   4. the plain bench line, CPU baseline included                                         -> bench.json
# (counter passes carry --kernel-trace only; never a trace domain beside --pmc)
set -e
OUT=${1:-gpurun_out/profile}
shift || true
ARGS="$@"
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline $ARGS > $OUT/trace_run.txt 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 $ARGS > $OUT/pmc_fetch_run.txt 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 $ARGS > $OUT/pmc_write_run.txt 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 $ARGS > $OUT/pmc_sq1_run.txt 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 $ARGS > $OUT/pmc_sq2_run.txt 2>&1
# the executed vector-instruction mix (for the vector-issue floor: a SIMD-32 takes a wave's 32-bit instruction in 2 cycles, an
# f64 one in 4, a transcendental in 8; MI355X_MICROARCH.md)
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --kernel-trace --output-format csv -d $OUT/pmc_sq3 -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 $ARGS > $OUT/pmc_sq3_run.txt 2>&1 || echo "instruction-mix pass failed"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq4 -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 $ARGS > $OUT/pmc_sq4_run.txt 2>&1 || echo "instruction-mix pass 2 failed"
python3 - $OUT "$ARGS" <<'PY'
import csv, glob, json, sys, collections, re
out, args = sys.argv[1], sys.argv[2].split()
def opt(name, default):
    return int(args[args.index(name) + 1]) if name in args else default
def short(name):
    m = re.search(r"lfx::(\w+)", name)
    key = m.group(1) if m else name[:40]
    if key == "ring_unit_kernel" and "<true" in name:
        key = "ring_unit_kernel(second pass)"
    return key
def mean_per_kernel(d, counter=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if counter and row["Counter_Name"] != counter:
                continue
            key = short(row["Kernel_Name"])
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[key][row["Counter_Name"]] += 1
    return {k: {c: round(acc[k][c] / cnt[k][c], 1) for c in acc[k]} for k in acc}
fetch = {k: v["FETCH_SIZE"] for k, v in mean_per_kernel(out + "/pmc_fetch", "FETCH_SIZE").items()}
write = {k: v["WRITE_SIZE"] for k, v in mean_per_kernel(out + "/pmc_write", "WRITE_SIZE").items()}
hbm = {k: int(2 * fetch.get(k, 0) * 1024 + write.get(k, 0) * 1024) for k in set(fetch) | set(write)}
import hashlib, os
h = hashlib.sha256()
for name in ("lfx_kernels_common.hpp", "lfx_kernels_unit.hpp", "lfx_kernels_extract.hpp"):
    h.update(open(os.path.join("lidar_feature_extraction_amd", "csrc", name), "rb").read())
json.dump({"batch": opt("--batch", 1024), "rings": opt("--rings", 64), "cols": opt("--cols", 1800), "drop_zero": "--drop-zero" in args,
           "kernels_sha256": h.hexdigest(),      # bench.py quotes these bytes only for the sources they were measured on
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB per dispatch, mean over dispatches); "
                   "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md: it counts 128-B requests at 64 B)",
           "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "hbm_bytes_per_launch": hbm,
           "hbm_bytes_per_launch_total": sum(v for k, v in hbm.items() if k.startswith(("ring_", "feature_", "batch_", "fallback_", "grid_", "scan_count")))},
          open(out + "/pmc_traffic.json", "w"), indent=1)
sq = mean_per_kernel(out + "/pmc_sq1")
for extra in ("/pmc_sq2", "/pmc_sq3", "/pmc_sq4"):
    for k, v in mean_per_kernel(out + extra).items():
        sq.setdefault(k, {}).update(v)
keep = {k: v for k, v in sq.items() if k.startswith(("ring_unit", "ring_stream", "ring_scatter", "feature_compact", "grid_count", "scan_count"))}
for k, v in keep.items():
    w = v.get("SQ_WAVES", 0) or 1
    v["per_wave"] = {c: round(v[c] / w, 1) for c in v if c.startswith(("SQ_INSTS", "SQ_WAVE_CYCLES", "SQ_ACTIVE", "SQ_WAIT"))}
    if "SQ_INSTS_VALU_FMA_F64" in v:
        # vector-issue floor of the kernel: executed instructions by cost class over the device's 1024 SIMDs
        f64 = v.get("SQ_INSTS_VALU_ADD_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0) + v.get("SQ_INSTS_VALU_FMA_F64", 0) + v.get("SQ_INSTS_VALU_INT64", 0)
        trans = v.get("SQ_INSTS_VALU_TRANS_F64", 0) + v.get("SQ_INSTS_VALU_TRANS_F32", 0)
        rest = max(v.get("SQ_INSTS_VALU", 0) - f64 - trans, 0)
        cycles = (2 * rest + 4 * f64 + 8 * trans) / 1024.0
        v["valu_issue_floor"] = {"f64_class": f64, "transcendental": trans, "other_32_bit": rest, "simd_cycles": round(cycles, 0),
                                 "us_at_2100_mhz": round(cycles / 2100.0, 1),
                                 "note": "2 / 4 / 8 cycles per wave instruction on a SIMD-32 (32-bit / f64 and 64-bit integer / transcendental), spread over 1024 SIMDs"}
json.dump({"note": "rocprofv3 --pmc, two passes of eight SQ counters, mean per dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)",
           "kernels": keep}, open(out + "/sq_counters.json", "w"), indent=1)
print(json.dumps(hbm))
PY
timeout -k 10 400 python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench_err.txt
cat $OUT/bench.json
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_sq4
