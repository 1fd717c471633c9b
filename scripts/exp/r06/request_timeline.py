#!/usr/bin/env python3
"""Device timeline of the LAST request of scripts/request_latency.py in a rocprofv3 kernel + memory-copy trace: every operation
of >= 15 us (and every idle gap of >= 20 us in front of one).  usage: request_timeline.py <trace dir>"""
import csv
import glob
import os
import re
import sys

ops = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?")))
ops.sort()
groups, cur = [], []
for o in ops:  # requests are separated by the host's work between them (> 2 ms without a device operation)
    if cur and o[0] - max(x[1] for x in cur) > 2_000_000:
        groups.append(cur)
        cur = []
    cur.append(o)
groups.append(cur)
g = [x for x in groups if any("k_panel<" in o[2] for o in x)][-1]
t0, prev_end, busy = g[0][0], g[0][0], 0
for s, e, n in g:
    gap = max(0, s - prev_end)
    if e - s >= 15_000 or gap >= 20_000:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  idle before {gap / 1e3:7.1f}  {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"# {(prev_end - t0) / 1e3:.1f} us first to last operation, device busy {busy / 1e3:.1f} us (streams overlap in the create), {len(g)} operations")
