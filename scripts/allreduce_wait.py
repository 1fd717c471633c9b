#!/usr/bin/env python3
"""How long does the stop test's all-reduce KERNEL wait behind the solve's own kernels when it runs on the second
stream?  Reads rocprofv3 kernel traces of scripts/exp/trace_window.py taken with a one-rank RCCL
communicator and OSC_RCCL_PROXY=1 (csrc/comm.hip: at world 1 ncclAllReduce launches nothing, so a kernel with the launch
shape of RCCL's ncclDevKernel_Generic -- 256 threads, 248 VGPRs, 37.6 KB LDS -- stands in for it) and reports, per
iteration of every settle after the warm-ups: the time from the end of the beta reduction (whose event the second
stream waits for) to the start of the stand-in kernel, its duration, and the time until the publish kernel has written
the word the host polls; and which kernels of the solve's stream were running when the stand-in started.
usage: allreduce_wait.py <label> <trace dir> [<label> <trace dir> ...]"""
import csv
import glob
import os
import sys

import numpy as np


def load(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("osc::(anonymous namespace)::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0], r.get("Queue_Id", "?")))
    rows.sort()
    return rows


def analyse(label, rows):
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_rows_to_slab")]
    if len(starts) > 6:
        rows = rows[starts[-6]:]  # the last six settles
    wait, dur, total, behind = [], [], [], {}
    for i, (s, e, name, q) in enumerate(rows):
        if not name.startswith("k_rccl_shape_proxy"):
            continue
        beta = max((r for r in rows[:i] if r[2].startswith("k_reduce_beta")), key=lambda r: r[1], default=None)
        pub = next((r for r in rows[i:] if r[2].startswith("k_publish_word")), None)
        if beta is None or pub is None:
            continue
        wait.append((s - beta[1]) / 1e3)
        dur.append((e - s) / 1e3)
        total.append((pub[1] - beta[1]) / 1e3)
        running = [r[2] for r in rows if r[3] != q and r[0] <= s < r[1]]
        key = running[0].split("<")[0] if running else "(nothing)"
        behind[key] = behind.get(key, 0) + 1
    if not wait:
        print(f"{label}: no stand-in kernel in the trace")
        return
    w, d, t = np.array(wait), np.array(dur), np.array(total)
    print(f"{label}: {len(w)} hand-overs | beta reduction end -> all-reduce kernel start: median {np.median(w):.1f} us, "
          f"p90 {np.percentile(w, 90):.1f}, max {w.max():.1f} | kernel {np.median(d):.1f} us | beta end -> word published: "
          f"median {np.median(t):.1f} us, p90 {np.percentile(t, 90):.1f}, max {t.max():.1f} | solve-stream kernel running at "
          f"its start: {behind}")


if __name__ == "__main__":
    a = sys.argv[1:]
    for label, d in zip(a[0::2], a[1::2]):
        analyse(label, load(d))
