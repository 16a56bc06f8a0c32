#!/bin/bash
# tools/ab.sh REV [NAME]  -- build the library of git revision REV as _lib/NAME.so (default ab_A) next to the working tree's
# build (_lib/liblfx.so) so that both can be timed on ONE device in one gpurun call:
#   tools/ab_env.sh 3 "LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/ab_A.so" ""
set -e
REV=${1:-HEAD}; NAME=${2:-ab_A}
T=$(mktemp -d)
git archive $REV lidar_feature_extraction_amd/csrc include | tar -x -C $T
if [ -f $T/lidar_feature_extraction_amd/csrc/lfx_wire.hip ]; then
  make -s -j5 -C $T/lidar_feature_extraction_amd/csrc ../_lib/liblfx.so      # (the library only: the callers under examples/ are not in the archive)
  cp $T/lidar_feature_extraction_amd/_lib/liblfx.so lidar_feature_extraction_amd/_lib/$NAME.so
else    # revisions of the single-file layout (rounds 1 and 2)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared \
    -o lidar_feature_extraction_amd/_lib/$NAME.so $T/lidar_feature_extraction_amd/csrc/lfx_api.hip
fi
rm -rf $T
echo built $NAME.so from $REV
