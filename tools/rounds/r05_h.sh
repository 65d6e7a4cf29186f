#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_edge_cases.py -k "reserved_slab" -x -q 2>&1 | tail -3
python tools/superrest_timing.py 100000 12 250 > $O/superrest_plain.txt 2>&1; tail -3 $O/superrest_plain.txt
SCRI_AMD_TRACE=1 python tools/superrest_timing.py 100000 12 250 --reserve > $O/superrest_reserve_trace.txt 2>&1; grep -v " us$" $O/superrest_reserve_trace.txt | tail -8
python tools/superrest_timing.py 100000 12 250 --reserve > $O/superrest_reserve.txt 2>&1; tail -4 $O/superrest_reserve.txt
