#!/bin/bash
# profiles/r04_allreduce_wait.txt: the stop test's all-reduce beside the solve (one rank, RCCL-shaped stand-in kernel).
# run on the GPU box from the repo root:  bash scripts/allreduce_wait.sh
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_arw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export OSC_SHARD_TIMES_RCCL=1 OSC_RCCL_PROXY=1
{
echo "# settle of rank 0's window (one-rank RCCL communicator + RCCL-shaped stand-in kernel per all-reduce), no profiler"
for W in 8 1; do for M in 1 0; do
  echo "## window 1/$W of config 3, OSC_COMM_OVERLAP=$M"; OSC_COMM_OVERLAP=$M timeout -k 10 300 python3 $R/scripts/shard_local_times.py c3 $W 2>&1 | grep "^c3" | cut -c1-140
done; done
echo "## window 1/8, no communicator at all"; OSC_SHARD_TIMES_RCCL= OSC_RCCL_PROXY= timeout -k 10 300 python3 $R/scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-140
} > $O/times.txt 2>&1
ARGS=""
for W in 8 1; do
  rm -rf $O/t$W
  OSC_COMM_OVERLAP=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/t$W -- python3 $R/scripts/exp/trace_window.py $W > $O/t$W.log 2>&1 || { tail -5 $O/t$W.log; exit 1; }
  ARGS="$ARGS window_1/$W $O/t$W"
done
python3 $R/scripts/allreduce_wait.py $ARGS > $O/wait.txt
python3 $R/scripts/exp/timeline_last_settle.py $O/t8 > $O/timeline8.txt
rm -rf $O/t8 $O/t1
cat $O/times.txt $O/wait.txt
