#!/usr/bin/env python3
"""Compact table of a rocprofv3 *kernel_stats.csv: calls, average ms, share.  usage: kernel_stats_table.py <dir or csv> [rows]"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    name = r["Name"].replace("void ", "").replace("osc::(anonymous namespace)::", "").split("(")[0][:48]
    print(f"{name:48s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs']) / 1e6:9.4f} ms  total {float(r['TotalDurationNs']) / 1e6:9.3f} ms  {float(r['Percentage']):5.2f} %")
