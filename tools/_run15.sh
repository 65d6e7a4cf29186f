#!/bin/bash
export PYTHONPATH=.
python bench.py --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python tools/kernel_line.py cfg3
python bench.py --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python tools/kernel_line.py cfg3
python bench.py --workload cfg2 --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python tools/kernel_line.py cfg2
python bench.py --workload cfg5 --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | python tools/kernel_line.py cfg5
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
