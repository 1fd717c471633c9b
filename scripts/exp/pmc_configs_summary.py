import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for tag in sorted({os.path.basename(d).split("_")[0] for d in glob.glob(out + "/*x*_*") if os.path.isdir(d)}):
    vals = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
    for d in glob.glob(f"{out}/{tag}_*"):
        if not os.path.isdir(d): continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"].replace("void ", "").replace("osc::(anonymous namespace)::", "").split("(")[0]
                vals[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"].replace("void ", "").replace("osc::(anonymous namespace)::", "").split("(")[0]
                dur[n].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    print(f"== {tag} (N x D x k)")
    for n, cs in sorted(vals.items()):
        if not n.startswith(("k_spmm", "k_update", "k_apply_blocked", "k_chain_fix", "k_init_finish", "k_rows_to_slab")): continue
        def live(v):
            cut = 0.05 * max(v); return [x for x in v if x >= cut] or v
        e = {c: sum(live(v)) / len(live(v)) for c, v in cs.items()}
        d = live(dur[n]); ms = sum(d) / len(d) / 1e6
        rd = 2 * 1024 * e.get("FETCH_SIZE", 0) / 1e9; wr = 1024 * e.get("WRITE_SIZE", 0) / 1e9
        hit = e.get("TCC_HIT_sum", 0) / max(1.0, e.get("TCC_HIT_sum", 0) + e.get("TCC_MISS_sum", 0))
        print(f"  {n:28s} launches {len(d):4d}  mean {ms:7.3f} ms  read {rd:7.2f} GB (x2-corrected)  write {wr:6.2f} GB  L2 hits {e.get('TCC_HIT_sum', 0)/1e6:7.1f} M / misses {e.get('TCC_MISS_sum', 0)/1e6:7.1f} M ({100*hit:5.1f} %)  -> {(rd+wr)/ms:6.2f} TB/s of traffic")
