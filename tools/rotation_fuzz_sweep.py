"""Extended seeded sweep of tests/test_gpu_fuzz.py::test_random_rotation beyond the seeds in the suite
(python tools/rotation_fuzz_sweep.py [last_seed]); prints failures."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scri_amd
import tests.test_gpu_fuzz as F

ctx = scri_amd.Context(0)
bad = 0
last = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for seed in range(40, last):
    try:
        F.test_random_rotation(ctx, seed)
    except AssertionError as e:
        bad += 1
        print("rotation seed", seed, "FAILED", str(e)[:300])
print("done,", last - 40, "cases, failures:", bad)
