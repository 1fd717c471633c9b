#!/bin/bash
# Round 6: counters of the tile core's main sweep (k_tile_thr2) and sample sweep (k_tile_thr<0>) at config 5's shape
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_tile_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OSC_CREATE_STREAM=0
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $ROOT/scripts/knn_only.py 200000 1536 64 > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for name in ("k_tile_thr2", "k_tile_thr<0>"):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(f"{sys.argv[1]}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{sys.argv[1]}/p*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]: dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    s = lambda k: sum(acc[k])
    n = len(dur)
    tot_ms = sum(dur) / 3.0  # three passes traced
    print(f"== {name}: {n} launches over 3 passes, {tot_ms / 2:.2f} ms per build (2 builds per pass)")
    cyc = s("GRBM_GUI_ACTIVE") / 8.0
    print(f"  clock {cyc / (sum(dur) / 3.0) / 1e6:.3f} GHz; MFMA-busy {100 * s('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / cyc:.1f} %; waiting {100 * s('SQ_WAIT_ANY') / s('SQ_WAVE_CYCLES'):.1f} %, issue stalls {100 * s('SQ_WAIT_INST_ANY') / s('SQ_WAVE_CYCLES'):.1f} % (LDS {100 * s('SQ_WAIT_INST_LDS') / s('SQ_WAVE_CYCLES'):.1f} %), issuing {100 * s('SQ_ACTIVE_INST_ANY') / s('SQ_WAVE_CYCLES'):.1f} %; L2 hit {100 * s('TCC_HIT_sum') / (s('TCC_HIT_sum') + s('TCC_MISS_sum')):.1f} %")
PY
rm -rf $OUT/p*/
