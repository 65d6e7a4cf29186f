#!/bin/bash
export PYTHONPATH=.
python bench.py --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python tools/kernel_line.py cfg3
python bench.py --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python tools/kernel_line.py cfg3
