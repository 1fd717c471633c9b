set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_nbpmc
mkdir -p $O
export OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip_nb48.so
cd /tmp && export TMPDIR=/tmp
for CFG in "200000 1536 64 chain" "100000 768 32 nochain"; do
  T=$(echo $CFG | cut -d' ' -f1-3 | tr ' ' 'x')
  if [ "$T" = "200000x1536x64" ]; then NBS="16 20 24 28"; else NBS="9 12"; fi
  for B in $NBS; do
    for C in "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE; do
      N=$(echo $C | tr ' ' '_')
      OSC_SPMM_BLOCKED=$B rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${T}nb${B}_$N -- python3 $R/scripts/exp/settle_loop.py $CFG 3 > $O/${T}nb${B}_$N.log 2>&1
    done
  done
done
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
tags = sorted({os.path.basename(d).rsplit("_", 1)[0].replace("_TCC_HIT_sum_TCC_MISS", "").replace("_FETCH", "") for d in glob.glob(out + "/*nb*") if os.path.isdir(d)})
for tag in tags:
    vals = defaultdict(list); dur = []
    for d in glob.glob(f"{out}/{tag}_*"):
        if not os.path.isdir(d): continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_apply_blocked" in r["Kernel_Name"]: vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_apply_blocked" in r["Kernel_Name"]: dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    live = lambda v: [x for x in v if x >= 0.05 * max(v)] or v
    e = {c: sum(live(v)) / len(live(v)) for c, v in vals.items()}
    d = live(dur); ms = sum(d) / len(d) / 1e6
    print(f"{tag:28s} apply {ms:7.3f} ms  hits {e.get('TCC_HIT_sum',0)/1e6:7.1f} M  misses {e.get('TCC_MISS_sum',0)/1e6:7.1f} M  fetched {2*1024*e.get('FETCH_SIZE',0)/1e9:6.2f} GB (x2)")
PY
