#!/bin/bash
O=gpurun_out/r05n; mkdir -p $O
python tools/fuzz_sweep.py 1500 400 > $O/fuzz_default.txt 2>&1; tail -2 $O/fuzz_default.txt
SCRI_AMD_SYNTHESIS_EVAL=1 SCRI_AMD_NO_SMALL_DENSE=1 python tools/fuzz_sweep.py 1200 36 > $O/fuzz_synthesis_eval.txt 2>&1; tail -2 $O/fuzz_synthesis_eval.txt
python tools/rotation_fuzz_sweep.py > $O/fuzz_rotation.txt 2>&1; tail -2 $O/fuzz_rotation.txt
