#!/bin/bash
export PYTHONPATH=.
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/run_profiles.sh
python tools/superrest_timing.py 100000 12 250 > gpurun_out/superrest_timing.txt 2>&1; tail -2 gpurun_out/superrest_timing.txt
python bench.py --workload cfg2 --cpu-sample 0 > gpurun_out/bench_cfg2.log 2>&1
python bench.py --workload cfg5 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/bench_cfg5.log 2>&1
python bench.py --workload cfg4 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/bench_cfg4.log 2>&1
bash tools/pmc_sq_probe.sh > gpurun_out/sq_probe.log 2>&1
