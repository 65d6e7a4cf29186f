import time, numpy as np
from scri_amd import _lib
_lib.Context(0)
prev = None
for i in range(6):
    t0 = time.perf_counter()
    a = _lib.pinned_empty((99961, 285), np.complex128)
    t1 = time.perf_counter()
    a[0, 0] = 1  # touch
    prev = a
    print(i, f"alloc {1e3 * (t1 - t0):.2f} ms", "pool", dict(_lib._PinnedBlock._pool), _lib._PinnedBlock._pooled_bytes, flush=True)
