"""Round 5: what a rank of a sharded lattice build spends, measured on ONE GPU with OSC_KNN_FAKE_SHARDS=G (the G per-rank
passes run one after another into one handle: total device build time / G = a rank's share, collectives excluded), with
the half sweep shared by the ranks (default) and with a full sweep per rank (OSC_KNN_PANEL_SYM=0; D > 768: the tile
prefilter, as before round 5).  The lattices must be the single-pass build's.  usage: sharded_build_times.py N D k G"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k, G = (int(t) for t in sys.argv[1:5])
Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
base = Oscillink(Y, kneighbors=k)
base.rebuild_graph()
t1 = base.graph_stats()[2]
want = base.graph_csr()
print(f"N={N} D={D} k={k}: single pass build {t1:.1f} ms, route {base.build_info()['prefilter']}, fallback {base.build_info()['fallback_rows']}", flush=True)
os.environ["OSC_KNN_FAKE_SHARDS"] = str(G)
for sym in ("1", "0"):
    os.environ["OSC_KNN_PANEL_SYM"] = sym
    lat = Oscillink(Y, kneighbors=k)
    ts = []
    for _ in range(3):
        lat.rebuild_graph()
        ts.append(lat.graph_stats()[2])
    got = lat.graph_csr()
    same = all(np.array_equal(a, b) for a, b in zip(want[:3], got[:3]))
    info = lat.build_info()
    print(f"  {G} passes, {'shared half sweep' if sym == '1' else 'full sweep per rank'}: build {min(ts):.1f} ms = {min(ts) / G:.1f} ms per rank; "
          f"route {info['prefilter']}, fallback rows {info['fallback_rows']}, lattice {'identical' if same else 'DIFFERENT'}", flush=True)
    lat.close()
