#!/bin/bash
# tools/ab.sh REV  -- build the kernels of git revision REV as _lib/ab_A.so next to the working tree's
# build (_lib/liblfx.so) so that both can be timed on ONE device in one gpurun call:
#   for i in 1 2 3; do LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/ab_A.so python bench.py ...; python bench.py ...; done
set -e
REV=${1:-HEAD}
T=$(mktemp -d)
mkdir -p $T/lidar_feature_extraction_amd/csrc $T/include
git show $REV:lidar_feature_extraction_amd/csrc/lfx_kernels.hpp > $T/lidar_feature_extraction_amd/csrc/lfx_kernels.hpp
git show $REV:lidar_feature_extraction_amd/csrc/lfx_api.hip > $T/lidar_feature_extraction_amd/csrc/lfx_api.hip
git show $REV:include/lfx.h > $T/include/lfx.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared \
  -o lidar_feature_extraction_amd/_lib/ab_A.so $T/lidar_feature_extraction_amd/csrc/lfx_api.hip
rm -rf $T
echo built ab_A.so from $REV
