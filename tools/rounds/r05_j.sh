#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -k "rotat" -x -q 2>&1 | tail -3
for a in "16 100000" "16 1000000" "8 100000" "24 100000"; do CPU_BASELINE=0 python3 tools/bench_rotation.py $a 2>/dev/null | tail -1; done | tee $O/rotation_lines_pairs.jsonl
