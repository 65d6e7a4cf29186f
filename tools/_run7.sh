mkdir -p gpurun_out/r02g
timeout 1800 python -m pytest tests -m gpu -q -x -k "edge_cases or sharding or transform_modes or transform_abd or superrest" > gpurun_out/r02g/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02g/pytest.log
tail -6 gpurun_out/r02g/pytest.log
python tools/host_mode_rate.py > gpurun_out/r02g/host_mode_rate.txt 2>&1; tail -5 gpurun_out/r02g/host_mode_rate.txt
