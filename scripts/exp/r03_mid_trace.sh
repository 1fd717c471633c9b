#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r03_mid_tr
rocprofv3 --kernel-trace --output-format csv -d $O/r03_mid_tr -- python3 $R/scripts/exp/trace_window.py 1 ${1:-20000} ${2:-128} ${3:-16} > $O/r03_mid_tr.log 2>&1 || { tail -5 $O/r03_mid_tr.log; exit 1; }
tail -1 $O/r03_mid_tr.log | cut -c1-120
python3 - <<P
import csv, glob, os
f = sorted(glob.glob("$O/r03_mid_tr/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last settle: from the last control fill that precedes an INIT-type kernel
idx = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("osc::(anonymous namespace)::", "").replace("void ", "")[:56]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:6.1f}  {n}")
    prev = max(prev, e)
P
rm -rf $O/r03_mid_tr
