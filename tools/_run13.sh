#!/bin/bash
export PYTHONPATH=.
python tools/analysis_knock_probe.py 0 2>&1 | tail -3
SCRI_AMD_NO_SPLIT_ANALYSIS=1 python tools/analysis_knock_probe.py 0 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q -k "map2salm or analysis or golden or reference_suite or kernels" 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python tools/kernel_line.py split
