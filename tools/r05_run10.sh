mkdir -p gpurun_out/r05b
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05b/gpu_suite.txt 2>&1; rc=$?; tail -5 gpurun_out/r05b/gpu_suite.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err; rc=$?; tail -3 gpurun_out/r05b/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05b/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_path_frac"], d["roofline"].get("frac_of_box_copy"), d["box"], d["parity_spot_check"])
print(d["roofline"]["kernel_us_per_launch"])
for c in d["configs"]:
    print(c.get("workload","")[:60], c.get("value"), c.get("ms_per_step"), c.get("frac"), c.get("whole_path_frac"), c.get("parity_spot_check"), c.get("error"))
PY
exit $rc
