mkdir -p gpurun_out/r02h
timeout 1800 python -m pytest tests/test_gpu_device_resident.py tests/test_gpu_superrest.py tests/test_gpu_charges.py -m gpu -q -x > gpurun_out/r02h/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02h/pytest.log
tail -8 gpurun_out/r02h/pytest.log
timeout 600 python -m cProfile -s cumtime tools/superrest_timing.py 100000 12 250 > gpurun_out/r02h/superrest_profile.txt 2>&1; grep -n "N = " gpurun_out/r02h/superrest_profile.txt; sed -n '/cumulative/,+45p' gpurun_out/r02h/superrest_profile.txt | cut -c1-160
