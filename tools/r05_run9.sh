mkdir -p gpurun_out/r05a
L=$PWD/lidar_feature_extraction_amd/_lib
tools/ab_env.sh 3 "LFX_LIB_PATH=$L/ab_C.so" "LFX_LIB_PATH=$L/liblfx.so" "LFX_LIB_PATH=$L/v_w6.so" "LFX_LIB_PATH=$L/v_w6.so LFX_DEBUG_UNIT_LDS_PAD=4500"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
