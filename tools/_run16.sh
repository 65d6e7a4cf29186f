#!/bin/bash
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_edge_cases.py -x -q -k "separable" 2>&1 | tail -2
python tools/separable_probe.py 2>&1 | grep -A2 "100000 supertranslation only h"
python bench.py --workload cfg2 --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python tools/kernel_line.py cfg2
