#!/bin/bash
# kernel timeline of the LAST lattice build of scripts/knn_only.py: bash scripts/exp/r04_build_timeline.sh N D k
# (start offset, duration, gap to the previous kernel's end; all in us)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/scripts/knn_only.py $1 $2 $3 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - $O/tr <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the last build starts at the last k_normalize_rows
last = max(i for i, n in enumerate(names) if 'k_normalize_rows' in n)
t0 = int(rows[last]['Start_Timestamp']); prev = t0
for r in rows[last:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('osc::', '').split('(')[0][:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:9.1f} gap {(s - prev) / 1e3:8.1f}  {n}")
    prev = e
print(f"total {(prev - t0) / 1e3:.1f} us")
PY
rm -rf $O/tr
