#!/bin/bash
# round 6, the round's set of measurements on the final build (one gpurun call; copy what is to be judged into profiles/ afterwards)
set -u
O=gpurun_out/r06z; mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_default.log 2>&1; echo "default: $(tail -1 $O/pytest_default.log)" | tee $O/suite_under_switches.txt
# (contexts take their route options from the environment when they are created: the whole suite on the other side of two shape rules)
for sw in SCRI_AMD_NO_SYNTHESIS_EVAL SCRI_AMD_NO_ABD_SIGMA_EVAL; do
  env $sw=1 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_$sw.log 2>&1; echo "== $sw=1: $(tail -1 $O/pytest_$sw.log)" | tee -a $O/suite_under_switches.txt
done
for i in 1 2 3 4 5; do python -m pytest tests/test_gpu_threading.py -x -q -p no:cacheprovider 2>&1 | tail -1; done > $O/threading_x5.txt; sort $O/threading_x5.txt | uniq -c
bash tools/collect_round_profiles.sh > $O/collect.log 2>&1
for ax in jitter sxs; do python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-live-pmc --time-axis $ax 2>/dev/null | tail -1 > $O/bench_cfg3_axis_$ax.json; done
# the multi-rank lines through the library functions (scri_amd.sharding.ShardedTransform): gloo dry runs on the box's one GPU
SCRI_AMD_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 3 --warmup 1 --cpu-sample 0 2>$O/gloo8_cfg4.err | tail -1 > $O/bench_cfg4_8ranks_1gpu_gloo.json
SCRI_AMD_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 3 --warmup 1 --cpu-sample 0 --overlap-halo 2>$O/gloo8_cfg4_overlap.err | tail -1 > $O/bench_cfg4_8ranks_1gpu_gloo_overlap.json
SCRI_AMD_BENCH_BACKEND=gloo python bench.py --workload cfg5 --gpus 8 --n-times 16000 --steps 3 --warmup 1 --cpu-sample 0 2>$O/gloo8_cfg5.err | tail -1 > $O/bench_cfg5_strong_8ranks_1gpu_gloo_16000.json
# one process over eight contexts (devices=[0]*8 on this box): cfg4 and a cfg5 slice, host arrays in and out
python bench.py --inprocess 8 --steps 3 --warmup 2 2>$O/inprocess8_cfg4.err | tail -1 > $O/bench_inprocess8_cfg4.json
python bench.py --inprocess 8 --workload cfg5 --n-times 16000 --steps 2 --warmup 2 2>$O/inprocess8_cfg5.err | tail -1 > $O/bench_inprocess8_cfg5_16000.json
python bench.py --inprocess 8 --workload cfg3 --time-axis sxs --steps 3 --warmup 2 2>$O/inprocess8_cfg3_sxs.err | tail -1 > $O/bench_inprocess8_cfg3_sxs.json
python tools/fuzz_sweep.py 400 240 > $O/fuzz_sweep.txt 2>&1; tail -3 $O/fuzz_sweep.txt
python tools/superrest_timing.py 100000 12 250 --reserve > $O/superrest_timing_reserve.txt 2>&1
ls $O | head -60
