set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for NRG in 1 2; do
  OSC_KNN_PANEL_NRG=$NRG rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_knn_c4_nrg$NRG -- python3 $R/scripts/knn_only.py 1000000 384 16 > $O/r03_knn_c4_nrg$NRG.log 2>&1
  OSC_KNN_PANEL_NRG=$NRG rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r03_knn_c4_nrg${NRG}_pmc -- python3 $R/scripts/knn_only.py 1000000 384 16 > $O/r03_knn_c4_nrg${NRG}_pmc.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
for nrg in (1, 2):
    for f in glob.glob(f"{o}/r03_knn_c4_nrg{nrg}/**/*kernel_stats.csv", recursive=True):
        for r in list(csv.DictReader(open(f)))[:8]:
            n = r["Name"].replace("void osc::(anonymous namespace)::", "").replace("osc::(anonymous namespace)::", "").split("(")[0]
            print(f"nrg{nrg} {n:30s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:9.3f} ms total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
    acc = {}
    for f in glob.glob(f"{o}/r03_knn_c4_nrg{nrg}_pmc/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_panel<6, 1" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(f"nrg{nrg} k_panel<6,1,*> counters:", {k: sum(v) / len(v) for k, v in acc.items()})
PY
