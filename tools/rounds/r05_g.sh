#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O
export SCRI_AMD_SYNTHESIS_EVAL=1
timeout 900 python -m pytest tests/test_gpu_edge_cases.py -k "evaluation_in_it" -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_guard_regions.py -k "boost-free" -x -q 2>&1 | tail -2
for ax in uniform jitter sxs; do python tools/boost_free_rate.py 100000 16 $ax 2>/dev/null | tail -1; done | tee $O/se_axes.txt
for l in 10 12 14; do python tools/boost_free_rate.py 100000 $l 2>/dev/null | tail -1; done | tee -a $O/se_axes.txt
unset SCRI_AMD_SYNTHESIS_EVAL
for ax in uniform jitter sxs; do python tools/boost_free_rate.py 100000 16 $ax 2>/dev/null | tail -1; done | tee $O/two_pass_axes.txt
for l in 10 12 14; do python tools/boost_free_rate.py 100000 $l 2>/dev/null | tail -1; done | tee -a $O/two_pass_axes.txt
