#!/bin/bash
# sample density of the panel prefilter's thresholds (one sample column in rho): build time and fallback rows
for cfg in "1000000 384 16" "100000 768 32"; do
  for rho in 0 8 12 16 24 32; do
    echo "== $cfg rho=$rho"
    OSC_KNN_PANEL_RHO=$rho timeout -k 10 300 python - $cfg <<'P' 2>&1 | tail -2
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from oscillink_amd import Oscillink
N, D, k = (int(x) for x in sys.argv[1:4])
Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k)
ts = []
for _ in range(3):
    lat.rebuild_graph(); ts.append(lat.graph_stats()[2])
print("build_ms", [round(t, 1) for t in ts], "nnz", lat.graph_stats()[0], {k: v for k, v in lat.build_info().items() if k in ("prefilter", "fallback_rows")})
P
  done
done
