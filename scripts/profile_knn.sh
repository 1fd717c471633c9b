#!/bin/bash
# kNN-build profiling passes (run through gpurun from the repo root)
set -u
TAG=${1:-r01_knn}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/scripts/knn_only.py > $OUT/trace.log 2>&1
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/scripts/knn_only.py > $OUT/pmc_$N.log 2>&1
done
tail -2 $OUT/*.log
