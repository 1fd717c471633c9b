#!/bin/bash
# wave-cycle breakdown of scripts/exp/knn_core.hip (run on the GPU box via gpurun from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/knn_core_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- $ROOT/oscillink_amd/build/knn_core 100000 ${1:-4} 256 > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_core" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:32s} {sum(acc[k]) / len(acc[k]):.4e}  (n={len(acc[k])})")
PY
