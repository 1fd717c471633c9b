# time the prefilter kernel under experiment builds (results of those builds are wrong by construction)
for v in "" noepi noepi_nogl noepi_nogl_nolds noepi_nolds; do
  if [ -n "$v" ]; then export OSC_LIB_PATH=$PWD/oscillink_amd/liboscillink_hip_$v.so; else unset OSC_LIB_PATH; fi
  python - <<'PY'
import os, ctypes as C, numpy as np, sys
sys.path.insert(0, os.getcwd())
from oscillink_amd import Oscillink
Y = np.random.default_rng(0).standard_normal((100000, 768)).astype(np.float32)
lat = Oscillink(Y, kneighbors=32)
lat._call("osc_profile_enable", 1); lat._call("osc_profile_reset")
try:
    lat.rebuild_graph()
except Exception as e:
    print("rebuild raised", type(e).__name__)
n, ms = C.c_int64(0), C.c_double(0.0)
lat._call("osc_profile_get", 3, C.byref(n), C.byref(ms))
print(os.environ.get("OSC_LIB_PATH", "default").split("hip_")[-1], "topk kernel ms", round(ms.value / max(1, n.value), 2), "launches", n.value)
PY
done
