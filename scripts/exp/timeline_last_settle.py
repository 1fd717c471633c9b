"""Timeline of the LAST settle in a rocprofv3 kernel trace (trace_window.py): start offset, duration, queue, kernel.
usage: timeline_last_settle.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_rows_to_slab" in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("osc::(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    prev_end = max(prev_end, e)
