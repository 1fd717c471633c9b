#!/bin/bash
# PMC passes (separate runs, kernel trace only beside them) of one settle loop at config 4's and config 5's shapes on one
# GPU, and of rank 0's column window of an 8-rank config-3 solve: HBM/fabric bytes and L2 hit rate of the operator apply there.  Usage (on the GPU box): bash scripts/pmc_configs.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG0=${1:-r03}
OUT=$ROOT/gpurun_out/${TAG0}_cfg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in "1000000 384 16" "200000 1536 64 chain"; do
  TAG=$(echo $CFG | cut -d' ' -f1-3 | tr ' ' 'x')
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_$N -- python3 $ROOT/scripts/exp/settle_loop.py $CFG > $OUT/${TAG}_$N.log 2>&1
  done
done
# rank 0's 96-column window of an 8-rank config-3 solve (what its matvec moves)
TAG=100000x768x32w8
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  OSC_FAKE_COL_SHARD=0/8 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_$N -- python3 $ROOT/scripts/exp/settle_loop.py 100000 768 32 12 > $OUT/${TAG}_$N.log 2>&1
done
python3 $ROOT/scripts/exp/pmc_configs_summary.py $OUT
