#!/bin/bash
# tools/smoke_lib.sh LIB... -- on the GPU box: the 16x900 smoke scan against the oracle under each _lib/<LIB>, 60 s each; stops at the first failure
for lib in "$@"; do
  LFX_LIB_PATH=$PWD/lidar_feature_extraction_amd/_lib/$lib timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$lib.log 2>&1 || { echo "smoke under $lib failed"; tail -3 gpurun_out/smoke_$lib.log; exit 1; }
  echo "$lib: $(tail -1 gpurun_out/smoke_$lib.log)"
done
