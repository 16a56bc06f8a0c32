mkdir -p gpurun_out/r05c
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05c/parity.txt 2>&1; rc=$?; tail -8 gpurun_out/r05c/parity.txt; [ $rc -eq 0 ] || exit $rc
L=$PWD/lidar_feature_extraction_amd/_lib
tools/ab_env.sh 3 "LFX_LIB_PATH=$L/ab_D.so" "LFX_LIB_PATH=$L/liblfx.so"
