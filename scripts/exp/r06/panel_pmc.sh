#!/bin/bash
# Round 6: wave-cycle / LDS / MFMA counters of the main sweep at config 3 with one and two row groups per wave
# (k_panel<12,1,1,true> vs k_panel<12,1,2,true>): whole-array launches only (OSC_CREATE_STREAM=0).  Run on the GPU box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_panel_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OSC_CREATE_STREAM=0
SHAPE="${1:-100000} ${2:-768} ${3:-32}"
for nrg in 1 2; do
  export OSC_KNN_PANEL_NRG=$nrg
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/nrg${nrg}_p$i -- python3 $ROOT/scripts/knn_only.py $SHAPE > $OUT/nrg${nrg}_p$i.log 2>&1
  done
done
python3 - $OUT "$SHAPE" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections, re
print("# scripts/exp/r06/panel_pmc.sh", sys.argv[2], ": per-launch means over the whole-array launches of the main sweep (2 per pass: create + rebuild)")
for nrg in ("1", "2"):
    MAIN = re.compile(r"k_panel<12, 1, %s, true>" % nrg)
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(f"{sys.argv[1]}/nrg{nrg}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if MAIN.search(r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{sys.argv[1]}/nrg{nrg}_p*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if MAIN.search(r["Kernel_Name"]):
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    m = lambda k: sum(acc[k]) / max(1, len(acc[k]))
    print(f"== k_panel<12, 1, {nrg}, true>: {sum(dur) / max(1, len(dur)):.3f} ms per launch under the profiler (n={len(dur)})")
    for k in sorted(acc): print(f"  {k:32s} {m(k):.4e}  (n={len(acc[k])})")
    if acc["GRBM_GUI_ACTIVE"]:
        cyc = m("GRBM_GUI_ACTIVE") / 8.0
        print(f"  -> shader clock {cyc / (sum(dur) / len(dur)) / 1e6:.3f} GHz (GRBM_GUI_ACTIVE / 8 / launch time); MFMA-busy "
              f"{100.0 * m('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0 / cyc:.1f} % of the SIMD cycles; wave cycles: waiting {100 * m('SQ_WAIT_ANY') / m('SQ_WAVE_CYCLES'):.1f} %, "
              f"issue stalls {100 * m('SQ_WAIT_INST_ANY') / m('SQ_WAVE_CYCLES'):.1f} % (LDS {100 * m('SQ_WAIT_INST_LDS') / m('SQ_WAVE_CYCLES'):.1f} %), issuing {100 * m('SQ_ACTIVE_INST_ANY') / m('SQ_WAVE_CYCLES'):.1f} %; "
              f"LDS instructions {m('SQ_INSTS_LDS'):.3e}, LDS-array cycles {m('SQ_LDS_IDX_ACTIVE'):.3e}")
PY
rm -rf $OUT/nrg*_p*/
