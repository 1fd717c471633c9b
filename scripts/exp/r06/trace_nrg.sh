#!/bin/bash
# round 6: kernel trace of one config-3 lattice build with one / two row groups per wave in the main sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_trace_nrg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for nrg in 1 2; do
  export OSC_KNN_PANEL_NRG=$nrg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nrg$nrg -- python3 $ROOT/scripts/knn_only.py ${1:-100000} ${2:-768} ${3:-32} > $OUT/nrg$nrg.log 2>&1
  f=$(find $OUT/nrg$nrg -name "*kernel_stats.csv" | head -1)
  echo "== nrg=$nrg"; head -8 $f | cut -c1-160
done
