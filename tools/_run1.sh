mkdir -p gpurun_out/r02a
free -g | head -2 > gpurun_out/r02a/mem.txt; nproc >> gpurun_out/r02a/mem.txt
SCRI_AMD_ZGEMM_4M=0 timeout 900 python tools/tolerance_probe.py 3M > gpurun_out/r02a/probe_3m.json 2> gpurun_out/r02a/probe_3m.err
SCRI_AMD_ZGEMM_4M=1 timeout 900 python tools/tolerance_probe.py 4M > gpurun_out/r02a/probe_4m.json 2> gpurun_out/r02a/probe_4m.err
timeout 900 python tools/tolerance_probe.py auto > gpurun_out/r02a/probe_auto.json 2> gpurun_out/r02a/probe_auto.err
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r02a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
tail -5 gpurun_out/r02a/pytest.log; cat gpurun_out/r02a/probe_*.json; cat gpurun_out/r02a/bench.json
