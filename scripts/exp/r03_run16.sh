set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for D in 768 1536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_knn_200k_$D -- python3 $R/scripts/knn_only.py 200000 $D 64 > $O/r03_knn_200k_$D.log 2>&1
  tail -1 $O/r03_knn_200k_$D.log
  f=$(find $O/r03_knn_200k_$D -name "*kernel_stats.csv" | head -1)
  python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    n = r["Name"].replace("void osc::(anonymous namespace)::", "").replace("osc::(anonymous namespace)::", "").split("(")[0]
    print(f"   {n:34s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:9.3f} ms")
PY
done
