#!/usr/bin/env python3
"""Host transfers around a solve at config 3 (the boundary hands over host buffers): `lat.U` after a settle (307 MB device
-> a fresh NumPy array), `refresh_Ustar()` (solve + the same download), with the pinned chunked download (default) and
the plain copy into pageable memory (OSC_PINNED_DL=0), each in its own process.  usage: transfer_times.py [N D k]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(N, D, k):
    from oscillink_amd import Oscillink

    Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    tu, tr, ts = [], [], []
    for _ in range(6):
        lat.reset_U()
        t0 = time.perf_counter()
        lat.settle(max_iters=12, tol=1e-3)
        t1 = time.perf_counter()
        U = lat.U
        t2 = time.perf_counter()
        Us = lat.refresh_Ustar()
        t3 = time.perf_counter()
        ts.append(t1 - t0), tu.append(t2 - t1), tr.append(t3 - t2)
    ok = bool(np.isfinite(U).all() and np.isfinite(Us).all() and abs(float(U[N // 2, 3])) > 0)
    print(f"N={N} D={D} pinned_results={os.environ.get('OSC_PINNED_RESULTS', '1')} pinned_dl={os.environ.get('OSC_PINNED_DL', '1')} threads={os.environ.get('OSC_COPY_THREADS', 'auto')}: "
          f"settle_ms={1e3 * np.median(ts):.2f} U_read_ms={1e3 * np.median(tu[1:]):.2f} (first {1e3 * tu[0]:.1f}) "
          f"refresh_Ustar_ms={1e3 * np.median(tr[1:]):.2f} (solve {lat.last_ustar['solve_ms']:.2f}) finite={ok} "
          f"checksum={float(U.astype(np.float64).sum()):.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(*(int(x) for x in sys.argv[2:5]))
    else:
        cfg = sys.argv[1:4] if len(sys.argv) > 3 else ["100000", "768", "32"]
        for env in ({"OSC_PINNED_RESULTS": "0", "OSC_PINNED_DL": "0"}, {"OSC_PINNED_RESULTS": "0"},
                    {"OSC_PINNED_RESULTS": "0", "OSC_COPY_THREADS": "4"}, {}, {"OSC_PINNED_RESULTS": "2"}):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", *cfg], env={**os.environ, **env}, check=False)
