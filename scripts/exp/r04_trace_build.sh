#!/bin/bash
# kernel table of one lattice build: bash scripts/exp/r04_trace_build.sh N D k
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tb; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/scripts/knn_only.py $1 $2 $3 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 $R/scripts/exp/kernel_stats_table.py $O/tr 16
rm -rf $O/tr
