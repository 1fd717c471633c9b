#!/bin/bash
# round 6: hand-placed cores, one vs two row groups, at sizes without a tail on 256 workgroups
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_core
mkdir -p $OUT
B=$ROOT/oscillink_amd/build
{
for N in 65536 131072; do
for rep in 1 2; do
  timeout -k 10 120 $B/knn_core32 $N 4 256
  timeout -k 10 120 $B/knn_core32s_1_stamp $N 256 5
  timeout -k 10 120 $B/knn_core2_3_2 $N 256 5
  for v in 3_2 4_2 2_3 2_5; do timeout -k 10 120 $B/knn_core2s_$v $N 256 5; done
done
done
} 2>&1 | tee $OUT/core_ab3.txt
