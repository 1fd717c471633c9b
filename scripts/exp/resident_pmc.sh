#!/bin/bash
# L2 hit rate of the resident_spmm variants (run on the GPU box through gpurun from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/resident_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ARGS in "8 1 128 0 1" "8 4 128 1 1" "24 4 64 1 1" "24 3 64 1 1" "16 4 96 1 1"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/run$i -- $ROOT/oscillink_amd/build/resident_spmm 100000 768 $ARGS > $OUT/run$i.log 2>&1
  echo "== $ARGS" >> $OUT/summary.txt
  python3 - "$OUT/run$i" >> $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_res" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in acc.items(): print(k, s / max(n, 1), n)
if "TCC_HIT_sum" in acc and "TCC_MISS_sum" in acc:
    h = acc["TCC_HIT_sum"][0]; m = acc["TCC_MISS_sum"][0]; print("L2 hit rate", h / (h + m))
PY
done
cat $OUT/summary.txt
