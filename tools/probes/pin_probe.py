import time, sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from scri_amd import _lib
shape=(99961, 285)
prev=None
for i in range(6):
    t0=time.perf_counter(); a=_lib.pinned_empty(shape, np.complex128); t1=time.perf_counter()
    print(i, f"{(t1-t0)*1e3:.2f} ms", "pool:", {k: len(v) for k,v in _lib._PinnedBlock._pool.items()}, "recent", list(_lib._PinnedBlock._recent)[-2:])
    prev=a
