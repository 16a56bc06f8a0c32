mkdir -p gpurun_out/r05a
L=$PWD/lidar_feature_extraction_amd/_lib
tools/ab_env.sh 2 "LFX_LIB_PATH=$L/liblfx.so" "LFX_LIB_PATH=$L/v_pairs.so" || exit 1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05a/gpu_suite.txt 2>&1; rc=$?; tail -5 gpurun_out/r05a/gpu_suite.txt; exit $rc
