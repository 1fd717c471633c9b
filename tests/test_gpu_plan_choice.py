"""The automatic choice of the operator-apply plan (osc_api.hip: xs_plan / blocked_plan / auto_slab / maybe_reorder, the
measured thresholds VERDICT r03 called a thicket) against every plan that can be forced through the OSC_* switches, on a
6-shape subset of scripts/shape_sweep.py (the full 41-shape table is profiles/r04_shape_sweep.txt): the default must be
within 10 % of the best forced plan.  Medians of 12 settles; one retry absorbs a noisy neighbour.
A wall-clock comparison, so it carries the `perf` marker and runs only with `pytest -m perf`: on a shared GPU
it can fail with no code defect, and it rebuilds six large lattices 5-8 times each."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.perf

SHAPES = [(20000, 128, 16, "iid"), (40000, 256, 32, "iid"), (100000, 768, 32, "iid"), (500000, 384, 16, "iid"),
          (150000, 640, 20, "clustered"), (200000, 1536, 64, "iid")]


def _sweep():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "shape_sweep.py")
    spec = importlib.util.spec_from_file_location("shape_sweep", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("N,D,k,kind", SHAPES)
def test_default_plan_is_within_ten_percent_of_the_best_forced_plan(N, D, k, kind):
    mod = _sweep()
    r = mod.sweep_shape(N, D, k, kind)
    if r["ratio"] > 1.10:
        r = mod.sweep_shape(N, D, k, kind, reps=20)
    assert r["ratio"] <= 1.10, r
