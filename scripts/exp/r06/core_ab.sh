#!/bin/bash
# round 6: stand-alone panel-GEMM cores side by side (run through gpurun from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_core
mkdir -p $OUT
B=$ROOT/oscillink_amd/build
N=${1:-100000}
{
for rep in 1 2; do
  timeout -k 10 120 $B/knn_core32 $N 4 256
  timeout -k 10 120 $B/knn_core16 $N 4 256
  for v in 4_2 4_1 2_5 2_3 2_1 6_1 3_3 3_2 1_8 1_11; do
    timeout -k 10 120 $B/knn_core2_$v $N 256 5
  done
done
} 2>&1 | tee $OUT/core_ab.txt
