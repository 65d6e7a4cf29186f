mkdir -p gpurun_out/r02b
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02b/pytest.log
tail -15 gpurun_out/r02b/pytest.log
