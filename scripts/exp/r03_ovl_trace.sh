#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for M in 1 0; do
  rm -rf $O/r03_ovl$M
  OSC_SHARD_TIMES_RCCL=1 OSC_COMM_OVERLAP=$M rocprofv3 --kernel-trace --output-format csv -d $O/r03_ovl$M -- python3 $R/scripts/exp/trace_window.py 8 > $O/r03_ovl$M.log 2>&1 || { tail -5 $O/r03_ovl$M.log; exit 1; }
  tail -1 $O/r03_ovl$M.log | cut -c1-100
  python3 $R/scripts/exp/timeline_last_settle.py $O/r03_ovl$M > $O/r03_ovl${M}_timeline.txt
  rm -rf $O/r03_ovl$M
done
cat $O/r03_ovl1_timeline.txt; echo ====; cat $O/r03_ovl0_timeline.txt
