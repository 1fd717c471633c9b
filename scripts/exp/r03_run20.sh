set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_drop
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in "" _drop _prev; do
  export OSC_LIB_PATH=$R/oscillink_amd/liboscillink_hip$L.so
  python3 $R/scripts/exp/settle_loop.py 100000 768 32 nochain 6 | tail -1 | cut -c1-150
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/hm$L -- python3 $R/scripts/exp/settle_loop.py 100000 768 32 nochain 3 > $O/hm$L.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fs$L -- python3 $R/scripts/exp/settle_loop.py 100000 768 32 nochain 3 > $O/fs$L.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
for L in ("", "_drop", "_prev"):
    acc = {}; dur = []
    for d in (f"{o}/hm{L}", f"{o}/fs{L}"):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_apply_blocked" in r["Kernel_Name"]: acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_apply_blocked" in r["Kernel_Name"]: dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    live = lambda v: [x for x in v if x >= 0.05 * max(v)] or v
    e = {k: sum(live(v)) / len(live(v)) for k, v in acc.items()}
    d = live(dur)
    print(f"lib{L or '(product)':10s} apply {sum(d)/len(d)/1e6:.3f} ms  hits {e.get('TCC_HIT_sum',0)/1e6:.1f} M  misses {e.get('TCC_MISS_sum',0)/1e6:.1f} M  fetched {2*1024*e.get('FETCH_SIZE',0)/1e9:.2f} GB")
PY
