set -e
O=gpurun_out/r05a; mkdir -p $O
L=$PWD/lidar_feature_extraction_amd/_lib
C="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
export LFX_LIB_PATH=$L/ab_A.so
tools/pmc.sh "$C" ring_unit_org > $O/pmc2_old.txt 2>&1
export LFX_LIB_PATH=$L/liblfx.so
tools/pmc.sh "$C" ring_unit_org > $O/pmc2_new.txt 2>&1
paste $O/pmc2_old.txt $O/pmc2_new.txt
unset LFX_LIB_PATH
tools/ab_run.sh 3
