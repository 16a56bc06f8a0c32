mkdir -p gpurun_out/r05c
L=$PWD/lidar_feature_extraction_amd/_lib
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05c/parity.txt 2>&1; rc=$?; tail -4 gpurun_out/r05c/parity.txt; [ $rc -eq 0 ] || exit $rc
tools/ab_env.sh 3 "LFX_LIB_PATH=$L/ab_E.so" "LFX_LIB_PATH=$L/liblfx.so" -- --params launch_yaml
tools/ab_env.sh 1 "LFX_LIB_PATH=$L/ab_E.so" "LFX_LIB_PATH=$L/liblfx.so"
