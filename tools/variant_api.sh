#!/bin/bash
# tools/variant_api.sh NAME "EXTRA FLAGS" [UNIT] -- an A/B build that differs from the working tree's library in ONE translation
# unit (default lfx_unit_v0: the unit kernels of the default parameter variant): that unit compiled with the extra flags,
# linked with the other units' objects as they stand -> _lib/NAME.so
# (use with tools/ab_env.sh "LFX_LIB_PATH=.../NAME.so"; several can be built side by side)
set -e
NAME=$1; EXTRA=$2; UNIT=${3:-lfx_unit_v0}
D=lidar_feature_extraction_amd/csrc; B=lidar_feature_extraction_amd/_build; L=lidar_feature_extraction_amd/_lib
mkdir -p $B/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w $EXTRA -c -o $B/$NAME/$UNIT.o $D/$UNIT.hip
OBJS=""
for u in lfx_api lfx_unit_v0 lfx_unit_v1 lfx_unit_v2 lfx_unit_v3 lfx_wire lfx_gather lfx_downsample lfx_localize; do
  if [ $u == $UNIT ]; then OBJS="$OBJS $B/$NAME/$u.o"; else OBJS="$OBJS $B/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $L/$NAME.so $OBJS
echo built $L/$NAME.so
