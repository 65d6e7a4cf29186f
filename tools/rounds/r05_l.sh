#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edge_cases.py -k "large_grid_analysis" -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_transform_abd.py tests/test_gpu_separable_large.py -x -q 2>&1 | tail -3
for sw in 0 1; do
  if [ $sw = 1 ]; then export SCRI_AMD_NO_FUSED_LARGE_ANALYSIS=1; fi
  python tools/separable_probe_abd.py 25000 24 3 2>&1 | tail -4 | tee $O/separable_probe_abd_nofused$sw.txt
done
