#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q > $O/r03_overlap_tests.txt 2>&1 || { tail -30 $O/r03_overlap_tests.txt; exit 1; }
tail -1 $O/r03_overlap_tests.txt
{
echo "# 8-rank window of config 3, no communicator"; timeout -k 10 300 python scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-125
echo "# one-rank RCCL communicator, stop test on the second stream"; OSC_SHARD_TIMES_RCCL=1 timeout -k 10 300 python scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-125
echo "# one-rank RCCL communicator, stop test inside the solve's stream"; OSC_SHARD_TIMES_RCCL=1 OSC_COMM_OVERLAP=0 timeout -k 10 300 python scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-125
} > $O/r03_overlap_times.txt 2>&1
cat $O/r03_overlap_times.txt
