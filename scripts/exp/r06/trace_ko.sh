#!/bin/bash
# round 6: the main sweep's whole-array launch with and without its hit test / its deliveries (measurement-only library variants)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_trace_ko
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OSC_CREATE_STREAM=0
for lib in "" _nodeliver _noepi; do
for nrg in 1 2; do
  export OSC_KNN_PANEL_NRG=$nrg
  export OSC_LIB_PATH=$ROOT/oscillink_amd/liboscillink_hip$lib.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l${lib}_nrg$nrg -- python3 $ROOT/scripts/knn_only.py ${1:-100000} ${2:-768} ${3:-32} > $OUT/l${lib}_nrg$nrg.log 2>&1
  f=$(find $OUT/l${lib}_nrg$nrg -name "*kernel_stats.csv" | head -1)
  echo "== lib=$lib nrg=$nrg"; grep "k_panel<12" $f | cut -c1-150
done
done
