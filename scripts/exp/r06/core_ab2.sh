#!/bin/bash
# round 6: the hand-placed K loop (knn_core32s) against the fenced one (knn_core32) and the two-row-group core
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_core
mkdir -p $OUT
B=$ROOT/oscillink_amd/build
{
for N in 98304 100000; do
for rep in 1 2; do
  timeout -k 10 120 $B/knn_core32 $N 4 256
  for v in 0 1 2; do timeout -k 10 120 $B/knn_core32s_$v $N 256 5; done
  timeout -k 10 120 $B/knn_core2_3_2 $N 256 5
done
done
} 2>&1 | tee $OUT/core_ab2.txt
