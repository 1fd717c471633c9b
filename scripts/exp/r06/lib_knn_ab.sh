#!/bin/bash
# kernel times of one whole-array config-3 build under library variants (oscillink_amd/liboscillink_hip_NAME.so): lib_knn_ab.sh NAME ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export OSC_CREATE_STREAM=0
for rep in 1 2; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset OSC_LIB_PATH; else export OSC_LIB_PATH=$ROOT/oscillink_amd/liboscillink_hip_$v.so; fi
  O=$ROOT/gpurun_out/libab_${v}_$rep; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $ROOT/scripts/knn_only.py > $O/run.log 2>&1
  f=$(find $O/t -name "*kernel_stats.csv" | sort | tail -1)
  echo "== $v (rep $rep): $(grep build_ms $O/run.log)"
  grep "panel_select\|panel_image\|k_knn_rescore_pair" $f | sed 's/(.*)"/"/' | cut -d, -f1,4 | tr '\n' ' '; echo
done
done
