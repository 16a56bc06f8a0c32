#!/bin/bash
# tools/variant_api.sh NAME "EXTRA FLAGS" -- an A/B build that differs from the working tree's library in the extraction unit
# only: lfx_api.hip compiled with the extra flags, linked with the other units' objects as they stand -> _lib/NAME.so
# (use with tools/ab_env.sh "LFX_LIB_PATH=.../NAME.so"; several can be built side by side)
set -e
NAME=$1; EXTRA=$2
D=lidar_feature_extraction_amd/csrc; B=lidar_feature_extraction_amd/_build; L=lidar_feature_extraction_amd/_lib
mkdir -p $B/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w $EXTRA -c -o $B/$NAME/lfx_api.o $D/lfx_api.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $L/$NAME.so $B/$NAME/lfx_api.o $B/lfx_wire.o $B/lfx_gather.o $B/lfx_downsample.o $B/lfx_localize.o
echo built $L/$NAME.so
