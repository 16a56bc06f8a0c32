set -e
O=gpurun_out/r05a; mkdir -p $O
L=$PWD/lidar_feature_extraction_amd/_lib
LFX_LIB_PATH=$L/ab_A_stamps.so timeout -k 10 200 python tools/stamps.py --by-ring > $O/stamps_ring_old.txt 2>&1
LFX_LIB_PATH=$L/liblfx_stamps.so timeout -k 10 200 python tools/stamps.py --by-ring > $O/stamps_ring_new.txt 2>&1
paste $O/stamps_ring_old.txt $O/stamps_ring_new.txt | tail -18
