#!/bin/bash
# round 5, the round's set of measurements on the final build (one gpurun call; copy what is to be judged into profiles/ afterwards)
set -u
O=gpurun_out/r05z; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_default.log 2>&1; echo "default: $(tail -1 $O/pytest_default.log)" | tee $O/suite_under_switches.txt
for sw in SCRI_AMD_SYNTHESIS_EVAL SCRI_AMD_NO_ABD_SIGMA_EVAL; do
  env $sw=1 python -m pytest tests -m gpu -q > $O/pytest_$sw.log 2>&1; echo "== $sw=1: $(tail -1 $O/pytest_$sw.log)" | tee -a $O/suite_under_switches.txt
done
for i in 1 2 3 4 5 6 7 8 9 10; do python -m pytest tests/test_gpu_threading.py -x -q 2>&1 | tail -1; done > $O/threading_x10.txt; sort $O/threading_x10.txt | uniq -c
bash tools/collect_round_profiles.sh > $O/collect.log 2>&1
for ax in jitter sxs; do python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-live-pmc --time-axis $ax 2>/dev/null | tail -1 > $O/bench_cfg3_axis_$ax.json; done
SCRI_AMD_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 3 --warmup 1 --cpu-sample 0 2>$O/gloo8_cfg4.err | tail -1 > $O/bench_cfg4_8ranks_1gpu_gloo.json
SCRI_AMD_BENCH_BACKEND=gloo python bench.py --workload cfg5 --gpus 8 --n-times 16000 --steps 3 --warmup 1 --cpu-sample 0 2>$O/gloo8_cfg5.err | tail -1 > $O/bench_cfg5_strong_8ranks_1gpu_gloo_16000.json
python tools/boost_free_rate.py 100000 16 > $O/boost_free_two_pass.txt 2>/dev/null
SCRI_AMD_SYNTHESIS_EVAL=1 python tools/boost_free_rate.py 100000 16 > $O/boost_free_fused.txt 2>/dev/null
python tools/fuzz_sweep.py 400 240 > $O/fuzz_sweep.txt 2>&1; tail -3 $O/fuzz_sweep.txt
python tools/superrest_timing.py 100000 12 250 --reserve > $O/superrest_timing_reserve.txt 2>&1
ls $O | head -40
