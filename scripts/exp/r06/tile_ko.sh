#!/bin/bash
# Round 6: measurement-only knockouts of the tile core's main sweep (k_tile_thr2) at config 5's shape: library variants
# -DOSC_TILE2_NODMA (stale stages), -DOSC_TILE2_NOBAR (no per-step wait + barrier), -DOSC_TILE2_NORD (fragments read once per K step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_t2ko; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OSC_CREATE_STREAM=0
for lib in "" _t2nodma _t2nobar _t2nord _t2none _t2nodel _t2noepi; do
  export OSC_LIB_PATH=$ROOT/oscillink_amd/liboscillink_hip$lib.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l$lib -- python3 $ROOT/scripts/knn_only.py 200000 1536 64 > $OUT/l$lib.log 2>&1
  f=$(find $OUT/l$lib -name "*kernel_stats.csv" | head -1)
  echo "== lib=$lib"; grep "k_tile_thr2" $f | cut -c1-140
done
