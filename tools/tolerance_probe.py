# Needs the probe build of the library (its switch does not exist in the default one; scri_amd/csrc/env.h):
#   make -C scri_amd/csrc PROBES=1 && export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
"""Observed maxima of the reference's analytic translation sweeps (tests/test_waveform_grid.py:41-158) on the GPU, without
assertions: run once per setting of SCRI_AMD_ZGEMM_4M (0 = three real products per complex one, 1 = four) to see which
digits the synthesis costs.  Prints one JSON line.   python tools/tolerance_probe.py [label]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import scri_amd  # noqa: E402
from oracle import quat, spinsfast_ref, wigner  # noqa: E402
from oracle import sample_waveforms_ref as samples  # noqa: E402
from tests.test_gpu_reference_suite import _translated_error, _zero_aux, to_gpu  # noqa: E402


def main():
    quat.ROBUST_POLES = True
    ctx = scri_amd.Context(0)
    out = {"label": sys.argv[1] if len(sys.argv) > 1 else "", "SCRI_AMD_ZGEMM_4M": os.environ.get("SCRI_AMD_ZGEMM_4M")}
    for s in range(-2, 3):
        aux = _zero_aux(s, ctx)
        worst = 0.0
        for ell in range(abs(s), 9):
            for m in range(-ell, ell + 1):
                for st in ([1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]):
                    w1 = to_gpu(samples.single_mode_proportional_to_time(s=s, ell=ell, m=m), ctx).transform(space_translation=st, **aux)
                    w2 = samples.single_mode_proportional_to_time_supertranslated(s=s, ell=ell, m=m, space_translation=np.array(st))
                    worst = max(worst, _translated_error(w1, w2, 1.0))
        out[f"space_s{s}"] = worst
        worst = 0.0
        for ellpp, mpp in wigner.LM_range(2, 4):
            ellpp, mpp = int(ellpp), int(mpp)
            st = np.zeros(25, dtype=complex)
            if mpp == 0:
                st[wigner.LM_index(ellpp, mpp, 0)] = 1.0
            elif mpp < 0:
                st[wigner.LM_index(ellpp, mpp, 0)] = 1.0
                st[wigner.LM_index(ellpp, -mpp, 0)] = (-1.0) ** mpp
            else:
                st[wigner.LM_index(ellpp, mpp, 0)] = 1.0j
                st[wigner.LM_index(ellpp, -mpp, 0)] = (-1.0) ** mpp * -1.0j
            disp = abs(spinsfast_ref.salm2map(st, 0, 4, 17, 17)).max()
            for ell in range(abs(s), 5):
                for m in range(-ell, ell + 1):
                    w1 = to_gpu(samples.single_mode_proportional_to_time(s=s, ell=ell, m=m), ctx).transform(supertranslation=st, **aux)
                    w2 = samples.single_mode_proportional_to_time_supertranslated(s=s, ell=ell, m=m, supertranslation=st)
                    worst = max(worst, _translated_error(w1, w2, disp))
        out[f"hyper_s{s}"] = worst
    w1 = to_gpu(samples.constant_waveform(), ctx)
    out["time_translation"] = float(np.abs(w1.data - w1.transform(time_translation=1.469).data).max())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
