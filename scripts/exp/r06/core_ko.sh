#!/bin/bash
# round 6: measurement-only knockouts of the hand-placed core (what does each part of the K loop cost in wall time?)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_core
mkdir -p $OUT
B=$ROOT/oscillink_amd/build
{
for rep in 1 2; do
  for b in knn_core32s_1_stamp $(cd $B && ls | grep knn_core32s_ko); do echo "## $b"; timeout -k 10 120 $B/$b ${N:-98304} 256 5; done
done
} 2>&1 | tee $OUT/core_ko.txt
