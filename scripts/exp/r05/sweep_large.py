import sys, os
sys.path.insert(0, "scripts")
import importlib.util
spec = importlib.util.spec_from_file_location("shape_sweep", "scripts/shape_sweep.py"); ss = importlib.util.module_from_spec(spec); spec.loader.exec_module(ss)
for N, D, k, kind in [(500000, 384, 16, "iid"), (600000, 768, 32, "iid"), (700000, 384, 16, "iid"), (1000000, 384, 16, "iid"), (1000000, 64, 8, "iid"), (1000000, 128, 16, "iid"), (400000, 256, 16, "clustered"), (1000000, 128, 16, "clustered"), (600000, 128, 16, "clustered")]:
    r = ss.sweep_shape(N, D, k, kind, reps=8)
    print(f"{N} {D} {k} {kind} | {r['iters']} | {r['default_ms']:.3f} {r['default_plan']} | {r['best_forced']}: {r['best_forced_ms']:.3f} ({r['best_forced_plan']}) | {r['ratio']:.3f} | {r['all']}", flush=True)
