#!/bin/bash
# round 3: deferred x update (8 instead of 9 array passes per CG iteration) and the stop test's all-reduce beside the
# solve: the whole GPU suite, then A/B timings (OSC_X_DEFER=0 is the previous loop)
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r03_xdefer_tests.txt 2>&1 || { tail -40 $O/r03_xdefer_tests.txt; exit 1; }
tail -2 $O/r03_xdefer_tests.txt
{
for v in 1 0; do
  echo "## OSC_X_DEFER=$v"
  OSC_X_DEFER=$v timeout -k 10 300 python scripts/shard_local_times.py c3 1 8 2>&1 | grep "^c3" | cut -c1-125
  OSC_X_DEFER=$v timeout -k 10 300 python scripts/mid_size_probe.py 2>&1 | grep "^N=" | cut -c1-60
done
echo "## one-rank RCCL communicator, 8-rank window: stop test on the second stream / inside the solve's stream"
OSC_SHARD_TIMES_RCCL=1 timeout -k 10 300 python scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-125
OSC_SHARD_TIMES_RCCL=1 OSC_COMM_OVERLAP=0 timeout -k 10 300 python scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-125
} > $O/r03_xdefer_times.txt 2>&1
cat $O/r03_xdefer_times.txt
