"""Timeline of one lattice create from a rocprofv3 trace (kernel + memory-copy): start offset, duration, stream of every
device operation of the LAST create the traced run made.
usage: create_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [creates in the run]"""
import csv, glob, os, re, sys

d = sys.argv[1]
ncreate = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ops = []
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"]) or re.search(r"(\w+)$", r["Kernel_Name"].split("(")[0])).group(1), r.get("Stream_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?") + " " + r.get("Size", "?") if "Size" in r else "COPY " + r.get("Direction", "?"), r.get("Stream_Id", "?")))
ops.sort()
# the creates: k_normalize_w is the last big kernel of a build; split the run at gaps > 3 ms
groups, cur = [], []
for o in ops:
    if cur and o[0] - max(x[1] for x in cur) > 3_000_000:
        groups.append(cur); cur = []
    cur.append(o)
if cur:
    groups.append(cur)
groups = [g for g in groups if any("k_panel" in o[2] for o in g)]
g = groups[-1]
t0 = g[0][0]
print(f"# {len(groups)} builds in the trace; the last one: {len(g)} operations, {(max(o[1] for o in g) - t0) / 1e6:.3f} ms")
for s, e, name, st in g:
    if e - s < 20_000 and "COPY" not in name:
        continue
    print(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e6:7.3f} ms  stream {st:>3}  {name}")
