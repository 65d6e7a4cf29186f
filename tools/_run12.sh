mkdir -p gpurun_out/r02z
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r02z/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02z/pytest.log
tail -9 gpurun_out/r02z/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/run_profiles.sh > gpurun_out/r02z/run_profiles.log 2>&1
python tools/pmc_summary.py gpurun_out > gpurun_out/r02z/pmc_cfg3.json
cp gpurun_out/stats/runc/*_kernel_stats.csv gpurun_out/r02z/kernel_stats.csv
tail -1 gpurun_out/bench_default.log > gpurun_out/r02z/bench_line.json
grep '^{"metric' gpurun_out/bench_under_rocprof.log | tail -1 > gpurun_out/r02z/bench_line_under_rocprof.json
python bench.py --workload cfg2 --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/r02z/bench_cfg2.json
python bench.py --workload cfg5 --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/r02z/bench_cfg5.json
python bench.py --workload cfg4 --steps 5 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/r02z/bench_cfg4_1gpu.json
python tools/superrest_timing.py 100000 12 250 2>&1 | tail -2 > gpurun_out/r02z/superrest_timing.txt
python tools/pcie_probe.py 2>&1 | tail -5 > gpurun_out/r02z/pcie_probe.txt
cut -c1-300 gpurun_out/r02z/bench_line.json; cut -c1-200 gpurun_out/r02z/bench_cfg2.json; cut -c1-200 gpurun_out/r02z/bench_cfg5.json; cut -c1-200 gpurun_out/r02z/bench_cfg4_1gpu.json; cat gpurun_out/r02z/superrest_timing.txt
