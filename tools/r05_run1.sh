set -e
mkdir -p gpurun_out/r05a
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/dpp tools/probes/dpp_wave_shift.hip && /tmp/dpp > gpurun_out/r05a/dpp.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05a/smoke.txt 2>&1 || { cat gpurun_out/r05a/smoke.txt; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05a/parity.txt 2>&1 || { tail -40 gpurun_out/r05a/parity.txt; exit 1; }
tail -3 gpurun_out/r05a/parity.txt
tools/ab_run.sh 2
