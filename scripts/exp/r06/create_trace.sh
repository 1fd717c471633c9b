#!/bin/bash
# round 6: device timeline (kernels + copies) of the LAST of three streamed creates of one shape (config 4 by default)
# usage (on the GPU box): bash scripts/exp/r06/create_trace.sh [c4|c5|x3] [OSC_CREATE_STREAM value]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=${1:-c4}
OUT=$ROOT/gpurun_out/r06_create_trace_$SHAPE${2:+_s$2}
mkdir -p $OUT
export OSC_CREATE_STREAM=${2:-1}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 $ROOT/scripts/exp/r06/create_n.py $SHAPE 3 > $OUT/run.log 2>&1
tail -4 $OUT/run.log
python3 $ROOT/scripts/exp/r05/create_timeline.py $OUT/t 3 > $OUT/timeline.txt 2>&1
head -120 $OUT/timeline.txt
