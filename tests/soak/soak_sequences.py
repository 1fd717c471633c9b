#!/usr/bin/env python3
"""Extended run of the randomised operation-sequence parity test (tests/test_gpu_parity.py) over many seeds; prints
the failing seeds.  Not collected by pytest (the suite runs seeds 0-3); run by hand on a GPU box to look for rare divergences:
    python tests/soak/soak_sequences.py 4 60"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oscillink_amd as amd  # noqa: E402
from oracle import oscillink_oracle as orc  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 40)
bad = []
for seed in range(lo, hi):
    try:
        T.test_random_operation_sequences_track_the_oracle(amd, orc, seed)
    except Exception:  # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED")
        traceback.print_exc(limit=3)
print(f"seeds {lo}..{hi - 1}: {len(bad)} failed {bad}  (OSC_SMALL_PATH={os.environ.get('OSC_SMALL_PATH', 'default')})")
sys.exit(1 if bad else 0)
