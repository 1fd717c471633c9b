#!/bin/bash
# Round-5 bounded experiment on k_apply_blocked: gather rounds in flight per wave.
# OSC_BLK_VARIANT: 0 = one round in flight (rounds 2-4), 1 = <14,7> two rounds at 4 waves/SIMD, 2 = <18,11> three rounds at
# 3 waves/SIMD, 3 = <28,7> four rounds at 2 waves/SIMD.  Usage (GPU box, repo root): bash scripts/exp/r05/blk_inflight.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05a}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
AB=scripts/exp/blocked_apply/ab.py
V="OSC_BLK_VARIANT=0 OSC_BLK_VARIANT=1 OSC_BLK_VARIANT=2 OSC_BLK_VARIANT=3"
{
  timeout -k 10 200 python $AB 100000 768 32 $V
  timeout -k 10 200 python $AB 100000 384 16 $V
  timeout -k 10 200 python $AB 200000 768 32 $V
  timeout -k 10 200 python $AB 50000 512 32 $V
  timeout -k 10 300 python $AB 200000 1536 64 $V
} > $OUT/ab.txt 2>&1
echo "ab done"; grep -c settle $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
WANT="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM"
HAVE=""
for c in $WANT; do if grep -qw "$c" $OUT/counters.txt; then HAVE="$HAVE $c"; fi; done
echo "counters present:$HAVE" | tee $OUT/have.txt
set -- $HAVE
P1="${1:-} ${2:-} ${3:-} ${4:-} ${5:-} ${6:-} ${7:-} ${8:-}"
shift 8 2>/dev/null
P2="$*"
for VAR in 0 1 2 3; do
  for PASS in "$P1" "$P2" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    [ -z "$(echo $PASS | tr -d ' ')" ] && continue
    N=$(echo $PASS | tr ' ' '_' | cut -c1-60)
    OSC_BLK_VARIANT=$VAR timeout -k 10 240 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_v${VAR}_$N -- python3 $ROOT/scripts/exp/settle_loop.py 100000 768 32 8 > $OUT/pmc_v${VAR}_$N.log 2>&1 || echo "pass failed: v$VAR $PASS"
  done
  echo "pmc v$VAR done"
done
python3 $ROOT/scripts/exp/r05/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
tail -40 $OUT/pmc_summary.txt
