#!/bin/bash
# Round-4 bounded experiment on the blocked matvec (VERDICT r03 item 6 ii): non-temporal hint on the slot rows' LDS-DMA.
# PMC passes (separate runs) of a config-3 settle loop with OSC_BLK_SLOT_NT=0 / 1.  usage (GPU box): bash scripts/exp/r04_slot_nt_pmc.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_slot_nt
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  TAG=100000x768x32nt$V
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | tr ' ' '_')
    OSC_BLK_SLOT_NT=$V rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_$N -- python3 $ROOT/scripts/exp/settle_loop.py 100000 768 32 12 > $OUT/${TAG}_$N.log 2>&1
  done
done
python3 $ROOT/scripts/exp/pmc_configs_summary.py $OUT | grep -v "k_update\|k_rows"
