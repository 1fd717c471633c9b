#!/usr/bin/env python3
"""Device timeline of the LAST bench step (reset + settle) in a rocprofv3 kernel + memory-copy trace of `bench.py --steps K`:
every operation with its start offset, duration and the idle time in front of it.  usage: step_timeline.py <trace dir>"""
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
# a step starts at reset_U's device-to-device copy of N x D floats (a blit kernel in the kernel trace, or a traced copy)
starts = [i for i, o in enumerate(ops) if ("copyBuffer" in o[2] or "DEVICE_TO_DEVICE" in o[2]) and o[1] - o[0] > 60_000]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # which step, counted from the end
i0, i1 = starts[-1 - K], starts[-K]
t0, prev_end = ops[i0][0], ops[i0][0]
busy = 0
for s, e, n in ops[i0:i1]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  idle before {max(0, s - prev_end) / 1e3:6.1f}  {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"# step: {(ops[i1][0] - t0) / 1e3:.1f} us from reset to reset, device busy {busy / 1e3:.1f} us, {i1 - i0} operations")
