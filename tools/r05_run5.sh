L=$PWD/lidar_feature_extraction_amd/_lib
tools/ab_env.sh 3 "LFX_LIB_PATH=$L/ab_A.so" "LFX_LIB_PATH=$L/ab_B.so" "LFX_LIB_PATH=$L/liblfx.so"
