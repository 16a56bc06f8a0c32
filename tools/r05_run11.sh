mkdir -p gpurun_out/r05b
timeout -k 10 1000 python -m pytest tests/test_gather_gpu.py tests/test_gather_shim_gpu.py tests/test_bench_two_ranks_gpu.py -x -q -m gpu > gpurun_out/r05b/gather_tests.txt 2>&1; rc=$?; tail -30 gpurun_out/r05b/gather_tests.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --force-gather > gpurun_out/r05b/bench_force_gather_pairs.json 2> gpurun_out/r05b/fg.err; rc=$?; tail -3 gpurun_out/r05b/fg.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05b/bench_force_gather_pairs.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["sharding"], d.get("gather_ms_per_step"), d["box"])
PY
exit $rc
