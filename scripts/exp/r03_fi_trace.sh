#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r03_fi_tr
rocprofv3 --kernel-trace --output-format csv -d $O/r03_fi_tr -- python3 $R/scripts/exp/trace_window.py ${1:-1} > $O/r03_fi_tr.log 2>&1 || { tail -5 $O/r03_fi_tr.log; exit 1; }
python3 $R/scripts/exp/timeline_last_settle.py $O/r03_fi_tr | cut -c1-110 > $O/r03_fi_timeline.txt
rm -rf $O/r03_fi_tr
head -12 $O/r03_fi_timeline.txt
