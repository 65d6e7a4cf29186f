#!/bin/bash
# after the narrow-panel change: the whole suite and a share of every sweep once more on the final build
O=gpurun_out/r05m; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; echo "suite: $(tail -1 $O/pytest_default.log)" | tee $O/summary.txt
SCRI_AMD_GEMM_EVAL_STEP=61 python -m pytest tests/test_gpu_transform_modes.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -q 2>&1 | tail -1 | sed 's/^/overlapping tiles (SCRI_AMD_GEMM_EVAL_STEP=61): /' | tee -a $O/summary.txt
timeout 900 python tools/fuzz_sweep.py 1500 36 2>&1 | grep -E "FAILED|done, failures" | sed 's/^/fuzz 36..1500: /' | tee -a $O/summary.txt
timeout 900 python tools/consistency_sweep.py 400 0 2>&1 | grep -E "FAILED|checked|done, failures" | sed 's/^/consistency 0..400: /' | tee -a $O/summary.txt
SWEEP_ABD=1 timeout 900 python tools/consistency_sweep.py 300 0 2>&1 | grep -E "FAILED|checked|done, failures" | sed 's/^/consistency ABD 0..300: /' | tee -a $O/summary.txt
