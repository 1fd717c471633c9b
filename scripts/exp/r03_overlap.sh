#!/bin/bash
# round 3: the stop test's all-reduce beside the next iteration (run_cg): tests, then the 8-rank window of config 3 with a
# one-rank RCCL communicator, overlapped and inline, and without a communicator
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q > gpurun_out/r03_overlap_tests.txt 2>&1 || { tail -30 gpurun_out/r03_overlap_tests.txt; exit 1; }
tail -3 gpurun_out/r03_overlap_tests.txt
{
echo "# no communicator"; timeout -k 10 300 python scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-120
echo "# one-rank RCCL communicator, all-reduce on the second stream"; OSC_SHARD_TIMES_RCCL=1 timeout -k 10 300 python scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-120
echo "# one-rank RCCL communicator, all-reduce inside the solve's stream"; OSC_SHARD_TIMES_RCCL=1 OSC_COMM_OVERLAP=0 timeout -k 10 300 python scripts/shard_local_times.py c3 8 2>&1 | grep "^c3" | cut -c1-120
} > gpurun_out/r03_overlap_times.txt 2>&1
cat gpurun_out/r03_overlap_times.txt
