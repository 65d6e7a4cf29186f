"""Extended seeded sweep of tests/test_gpu_fuzz.py beyond the seeds in the suite (python tools/fuzz_sweep.py [last_seed [first_seed]]); prints failures.
SCRI_AMD_FUZZ_AXIS=1: every boost along the polar axis of the rotated grid, all shapes on the separable route
(tests/test_gpu_axis_boost_separable.py::_axis_kwargs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scri_amd
from oracle import quat
quat.ROBUST_POLES = True
import tests.test_gpu_fuzz as F
if os.environ.get("SCRI_AMD_FUZZ_AXIS"):
    from tests.test_gpu_axis_boost_separable import _axis_kwargs
    os.environ["SCRI_AMD_AXIS_BOOST_MIN_WORK"] = "0"
    F._random_kwargs = _axis_kwargs
ctx = scri_amd.Context(0)
bad = 0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 36
for seed in range(first, int(sys.argv[1]) if len(sys.argv) > 1 else 240):
    try:
        F.test_random_waveform_transform.__wrapped__(ctx, seed) if hasattr(F.test_random_waveform_transform, "__wrapped__") else F.test_random_waveform_transform(ctx, seed)
    except AssertionError as e:
        bad += 1
        print("WM seed", seed, "FAILED", str(e)[:300])
for seed in range(10 + first - 36, 60 + first - 36):
    try:
        F.test_random_abd_transform(ctx, seed)
    except AssertionError as e:
        bad += 1
        print("ABD seed", seed, "FAILED", str(e)[:300])
print("done, failures:", bad)
