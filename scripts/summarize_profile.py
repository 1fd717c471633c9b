#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (rocprofv3 csv) into profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json.

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide (16 B/lane) coalesced read stream, so the read side is doubled.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    n = name.replace("void ", "").replace("osc::(anonymous namespace)::", "")
    return n.split("(")[0]


def newest(pattern_dir, pattern):
    """gpurun merges every call's output into the same local directory: only the newest run of a pass counts."""
    files = glob.glob(os.path.join(pattern_dir, "**", pattern), recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


for sub, name in (("trace", "kernel_stats"), ("trace_knn_exact", "knn_exact_kernel_stats")):
    stats = newest(os.path.join(src, sub), "*kernel_stats.csv")
    if stats:
        with open(stats[0]) as f, open(os.path.join(dst, f"{tag}_{name}.csv"), "w") as g:
            g.write(f.read())

pmc = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values
for d in glob.glob(os.path.join(src, "pmc_*")):
    for f in newest(d, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            pmc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))

dur = defaultdict(list)
for f in newest(os.path.join(src, "trace"), "*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        dur[short(row["Kernel_Name"])].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))

def live(v):
    """Drop the speculative CG launches that were gated off (they return at once: tiny counters / durations)."""
    if not v:
        return v
    cut = 0.05 * max(v)
    kept = [x for x in v if x >= cut]
    return kept or v


out = {}
for k, cs in pmc.items():
    e = {"launches_profiled": max(len(live(v)) for v in cs.values())}
    for c, v in cs.items():
        v = live(v)
        e[c + "_mean"] = sum(v) / len(v)
    if k in dur:
        d = live(dur[k])
        e["mean_ns_unprofiled_trace"] = sum(d) / len(d)
        e["launches_traced"] = len(d)
    if "FETCH_SIZE_mean" in e:
        e["hbm_read_bytes_per_launch"] = 2.0 * 1024.0 * e["FETCH_SIZE_mean"]  # gfx950: x2 for wide coalesced reads
    if "WRITE_SIZE_mean" in e:
        e["hbm_write_bytes_per_launch"] = 1024.0 * e["WRITE_SIZE_mean"]
    if "TCC_HIT_sum_mean" in e and "TCC_MISS_sum_mean" in e:
        e["l2_hit_rate"] = e["TCC_HIT_sum_mean"] / max(1.0, e["TCC_HIT_sum_mean"] + e["TCC_MISS_sum_mean"])
    out[k] = e
# the build the counters belong to: bench.py only quotes `traffic` from a profile whose lib_hash equals the running
# library's source hash (oscillink_amd/liboscillink_hip.so.stamp)
try:
    stamp = open(os.path.join(root, "oscillink_amd", "liboscillink_hip.so.stamp")).read().strip()
except OSError:
    stamp = None
meta = {"lib_hash": stamp, "tag": tag, "workload": "python3 bench.py (config 3, one GPU)"}
json.dump({"_meta": meta, **out}, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
for k, e in sorted(out.items()):
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items()})
