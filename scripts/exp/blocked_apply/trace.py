"""Per-wave entry times of the blocked apply's sub-phases on XCD 0 (OSC_BLK_TRACE).  Usage: blk_trace.py nb slices [env...]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
nb, nsl = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k_, _, v_ = kv.partition("="); os.environ[k_] = v_
os.environ["OSC_BLK_TRACE"] = "/tmp/blk_trace.bin"
from oscillink_amd import Oscillink
N, D, k = 100000, 768, 32
Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k); lat.set_query(psi)
for _ in range(2):
    lat.reset_U(); lat.settle(max_iters=12, tol=1e-3)
t = np.fromfile("/tmp/blk_trace.bin", dtype=np.uint64)
W = 512; nphase = 3 * nsl * nb
t = t[:W * nphase].reshape(W, nphase).astype(np.float64)
t0 = t[t > 0].min()
t = (t - t0) / 100.0  # wall_clock64 ticks at 100 MHz -> us
print("sub-phase: min / median / max entry time over the 512 waves of XCD 0 (us), spread")
for ph in range(nphase):
    c = t[:, ph]
    print(f"  ph {ph:3d} (slab {ph // (nsl * nb)} slice {(ph % (nsl * nb)) // nb} blk {ph % nb}): {c.min():8.1f} {np.median(c):8.1f} {c.max():8.1f}   spread {c.max() - c.min():7.1f}")
# who is late?  lateness of wave w at sub-phase ph = its entry time - the median entry time
late = t - np.median(t, axis=0, keepdims=True)
m = late[:, 1:].mean(axis=1)
order = np.argsort(-m)
print("latest waves (index within the XCD = workgroup * 4 + wave): mean lateness us, and at sub-phases 1..8")
for w in order[:12]:
    print(f"  wave {w:3d} (wg {w // 4:3d}): {m[w]:6.1f}   " + " ".join(f"{late[w, p]:6.1f}" for p in range(1, 9)))
print("earliest:")
for w in order[-6:]:
    print(f"  wave {w:3d} (wg {w // 4:3d}): {m[w]:6.1f}   " + " ".join(f"{late[w, p]:6.1f}" for p in range(1, 9)))
wg = late[:, 1:].mean(axis=1).reshape(-1, 4).mean(axis=1)
print("per workgroup mean lateness (us), workgroups 0..127:")
print(" ".join(f"{x:5.1f}" for x in wg))
