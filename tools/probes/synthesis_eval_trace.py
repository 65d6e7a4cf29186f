"""Timeline of synthesis_eval_kernel's phases (probe build, SCRI_AMD_SE_TRACE=<file>): workgroup 0, steps 8..23, every wave, clock
stamps at: 0 start of the step, 1 end of the role's phase-A work (theta: F of the row; phi: MFMAs + ring writes), 2 behind barrier 1,
3 end of the evaluation, 4 behind the write of the next row's modes, 5 behind barrier 2.
Usage (after make -C scri_amd/csrc PROBES=1): SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so SCRI_AMD_SYNTHESIS_EVAL=1 python tools/probes/synthesis_eval_trace.py"""
import os, subprocess, sys
import numpy as np
path = "/tmp/se_trace.bin"
env = dict(os.environ, SCRI_AMD_SE_TRACE=path, SCRI_AMD_SYNTHESIS_EVAL="1")
subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "..", "boost_free_rate.py"), "20000"], env=env, check=True, stdout=subprocess.DEVNULL)
t = np.fromfile(path, dtype=np.uint64).reshape(16, 8, 6).astype(np.int64)
d = np.diff(t, axis=2)  # [step][wave][phase]
names = ["phase A work", "wait at barrier 1", "evaluation", "modes write", "wait at barrier 2"]
print("cycles per step (s_memtime ticks; median over 16 steps), waves 0..2 = phi, 3..7 = theta")
for w in range(8):
    print(f"wave {w}: " + ", ".join(f"{n} {int(np.median(d[:, w, i]))}" for i, n in enumerate(names)) +
          f";  step {int(np.median(t[1:, w, 0] - t[:-1, w, 0]))}")
