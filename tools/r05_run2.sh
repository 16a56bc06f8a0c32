set -e
O=gpurun_out/r05a; mkdir -p $O
L=$PWD/lidar_feature_extraction_amd/_lib
LFX_LIB_PATH=$L/ab_A_stamps.so timeout -k 10 200 python tools/stamps.py > $O/stamps_old.txt 2>&1
LFX_LIB_PATH=$L/liblfx_stamps.so timeout -k 10 200 python tools/stamps.py > $O/stamps_new.txt 2>&1
paste $O/stamps_old.txt $O/stamps_new.txt
tools/pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" ring_unit_org > $O/pmc_new.txt 2>&1
cat $O/pmc_new.txt
