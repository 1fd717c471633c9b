#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: kernel-trace stats + PMC passes of the bench workload and of
# the lattice build.  Outputs land in gpurun_out/<tag>/ ; scripts/summarize_profile.py turns them into
# profiles/<tag>_*.  Counters are collected in their own passes (never combined with sys/hip/hsa tracing).
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 10 --warmup 2 --no-cpu-baseline"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$N.log 2>&1
done
# exact-fp32 lattice build for comparison (kernel stats only)
OSC_KNN_MODE=exact rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_knn_exact -- python3 $ROOT/scripts/knn_only.py > $OUT/trace_knn_exact.log 2>&1
find $OUT -name "*.csv" | wc -l
