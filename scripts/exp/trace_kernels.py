"""Per-kernel mean duration of the second half of a rocprofv3 kernel trace."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
rows = rows[len(rows) // 2:]
dur = defaultdict(list)
for s, e, n in rows:
    n = n.replace("void ", "").replace("osc::(anonymous namespace)::", "")
    dur[n.split("(")[0][:60]].append(e - s)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:62s} n={len(v):4d} mean={sum(v)/len(v)/1e3:9.2f} us  total={sum(v)/1e6:8.3f} ms")
