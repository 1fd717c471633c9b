"""Summarise a rocprofv3 kernel trace of repeated settles: per-kernel mean duration, and busy vs gap time of the last
solves (how much of a settle's wall time the device idles between launches)."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
rows = rows[len(rows) // 2:]  # steady state
dur = defaultdict(list)
for s, e, n in rows:
    dur[n.split("(")[0].replace("void osc::(anonymous namespace)::", "")[:60]].append(e - s)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:62s} n={len(v):5d} mean={sum(v)/len(v)/1e3:8.2f} us  total={sum(v)/1e3:9.1f} us")
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 20000]
print(f"launches {len(rows)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us ({100*busy/span:.0f} %)  "
      f"median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us  mean gap<20us {sum(small)/max(1,len(small))/1e3:.2f} us")
