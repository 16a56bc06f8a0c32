mkdir -p gpurun_out/r05a
L=$PWD/lidar_feature_extraction_amd/_lib
tools/ab_env.sh 3 "LFX_LIB_PATH=$L/liblfx.so" "LFX_LIB_PATH=$L/v_quads.so" "LFX_DEBUG_UNIT_LDS_PAD=4500" "LFX_DEBUG_UNIT_LDS_PAD=10000" || exit 1
LFX_LIB_PATH=$L/liblfx_stamps.so timeout -k 10 200 python tools/stamps.py > gpurun_out/r05a/stamps_turned.txt 2>&1; cat gpurun_out/r05a/stamps_turned.txt
