mkdir -p gpurun_out/r05a
L=$PWD/lidar_feature_extraction_amd/_lib
tools/ab_env.sh 2 "LFX_LIB_PATH=$L/liblfx.so" "LFX_LIB_PATH=$L/v_ablate.so" "LFX_LIB_PATH=$L/v_ablate.so LFX_DEBUG_UNIT_FLAGS=1024" "LFX_LIB_PATH=$L/liblfx.so LFX_DEBUG_UNIT_LDS_PAD=4500"
