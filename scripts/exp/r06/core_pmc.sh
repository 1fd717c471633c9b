#!/bin/bash
# round 6: in-kernel clock + wave-cycle counters of the stand-alone panel-GEMM cores (run through gpurun from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_core_pmc
mkdir -p $OUT
B=$ROOT/oscillink_amd/build
N=${N:-98304}
{
  for v in 0 1; do timeout -k 10 120 $B/knn_core32s_${v}_stamp $N 256 5; done
} 2>&1 | tee $OUT/stamps.txt
cd /tmp && export TMPDIR=/tmp
for BIN in "$@"; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"; do
    i=$((i+1))
    ARGS="$N 256 3"; case $BIN in knn_core32) ARGS="$N 4 256";; esac
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${BIN}_p$i -- $B/$BIN $ARGS > $OUT/${BIN}_p$i.log 2>&1
  done
  python3 - $OUT $BIN <<'PY' | tee $OUT/${BIN}_summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
dur = []
for f in glob.glob(sys.argv[1] + "/" + sys.argv[2] + "_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_core" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/" + sys.argv[2] + "_p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_core" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("==", sys.argv[2], "kernel ms (profiled passes): mean %.3f over %d launches" % (sum(dur) / max(len(dur), 1), len(dur)))
for k in sorted(acc): print(f"{k:32s} {sum(acc[k]) / len(acc[k]):.4e}  (n={len(acc[k])})")
PY
done
