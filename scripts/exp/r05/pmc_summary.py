"""Per-variant counter means of k_apply_blocked from the passes blk_inflight.sh wrote (gpurun_out/<tag>/pmc_v<variant>_*)."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]
for var in sorted({os.path.basename(d).split("_")[1] for d in glob.glob(out + "/pmc_v*_*") if os.path.isdir(d)}):
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in glob.glob(f"{out}/pmc_{var}_*"):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"].replace("void ", "").replace("osc::(anonymous namespace)::", "").split("(")[0]
                vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"].replace("void ", "").replace("osc::(anonymous namespace)::", "").split("(")[0]
                dur[n].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    print(f"== variant {var}")
    for n, cs in sorted(vals.items()):
        if not n.startswith("k_apply_blocked"):
            continue

        def live(v):
            cut = 0.05 * max(v)
            return [x for x in v if x >= cut] or v

        e = {c: sum(live(v)) / len(live(v)) for c, v in cs.items()}
        d = live(dur[n])
        ms = sum(d) / len(d) / 1e6
        print(f"  {n}: launches {len(d)} mean {ms:.4f} ms (under the profiler)")
        for c in sorted(e):
            print(f"      {c:24s} {e[c]:16.0f}")
        wc = e.get("SQ_WAVE_CYCLES", 0)
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM"):
                if c in e:
                    print(f"      {c} / SQ_WAVE_CYCLES = {e[c] / wc:.3f}")
        if "FETCH_SIZE" in e:
            print(f"      read {2 * 1024 * e['FETCH_SIZE'] / 1e9:.2f} GB (x2-corrected) written {1024 * e.get('WRITE_SIZE', 0) / 1e9:.2f} GB")
