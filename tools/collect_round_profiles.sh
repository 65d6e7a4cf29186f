#!/bin/bash
# The round's set of measurements in one gpurun call (run from the repo root on the MI355X box); everything lands in gpurun_out/.
# Copy what is to be judged into profiles/ afterwards (tools/pmc_summary.py gpurun_out > profiles/rNN_x_pmc_cfg3.json).
set -u
OUT=gpurun_out
mkdir -p $OUT
bash tools/run_profiles.sh > $OUT/run_profiles.log 2>&1
python3 bench.py --workload cfg2 --cpu-sample 0 2>/dev/null | tail -1 > $OUT/bench_cfg2_line.json
python3 bench.py --workload cfg4 --gpus 1 --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $OUT/bench_cfg4_1gpu_line.json
python3 bench.py --workload cfg5 --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $OUT/bench_cfg5_shard_line.json
python3 bench.py --boost-scale 26.7 --cpu-sample 0 --no-live-pmc 2>/dev/null | tail -1 > $OUT/bench_cfg3_beta1e-2.json
for a in "16 100000" "16 1000000" "8 100000" "24 100000"; do python3 tools/bench_rotation.py $a 2>/dev/null | tail -1; done > $OUT/rotation_lines.jsonl
for i in 1 2 3; do python3 tools/host_mode_rate.py 2>/dev/null | tail -1; done > $OUT/host_mode_rate.txt
python3 tools/superrest_timing.py 100000 12 250 > $OUT/superrest_timing.txt 2>&1
python3 tools/separable_probe_abd.py 25000 24 3 > $OUT/separable_probe_abd.txt 2>&1
ls -la $OUT | tail -30
