#!/bin/bash
# Where the panel prefilter's wave cycles go (k_panel<NKT, 1, NRG>, the main sweep): PMC passes over lattice builds at a
# config-4-like shape (K depth 6, two row groups per wave) and at config 3 (K depth 12, one).  Run on the GPU box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/panel_stall
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SHAPE in "200000 384 16" "100000 768 32"; do
  TAG=$(echo $SHAPE | tr ' ' 'x')
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_p$i -- python3 $ROOT/scripts/knn_only.py $SHAPE > $OUT/${TAG}_p$i.log 2>&1
  done
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections, os, re
MAIN = re.compile(r"k_panel<\d+, 1, \d+>")
for tag in ("200000x384x16", "100000x768x32"):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(f"{sys.argv[1]}/{tag}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if MAIN.search(r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{sys.argv[1]}/{tag}_p*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if MAIN.search(r["Kernel_Name"]):
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    print(f"== {tag}: main sweep {sum(dur) / max(1, len(dur)):.2f} ms (n={len(dur)})")
    for k in sorted(acc): print(f"  {k:32s} {sum(acc[k]) / len(acc[k]):.4e}  (n={len(acc[k])})")
PY
rm -rf $OUT/*_p*/
