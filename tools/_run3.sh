mkdir -p gpurun_out/r02c
timeout 1200 python -m pytest tests -m gpu -q -x -k "rotat or g10 or cfg1 or cfg2 or patched or D_match" > gpurun_out/r02c/pytest_rot.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02c/pytest_rot.log
tail -5 gpurun_out/r02c/pytest_rot.log
for L in 16 8; do
CPU_BASELINE=0 timeout 300 python tools/bench_rotation.py $L 100000 > gpurun_out/r02c/rot_l${L}_1e5.json 2>gpurun_out/r02c/rot_l${L}.err
CPU_BASELINE=0 timeout 300 python tools/bench_rotation.py $L 1000000 > gpurun_out/r02c/rot_l${L}_1e6.json 2>>gpurun_out/r02c/rot_l${L}.err
CPU_BASELINE=0 SCRI_AMD_ROTATE_STAGED=1 timeout 300 python tools/bench_rotation.py $L 100000 > gpurun_out/r02c/rot_l${L}_1e5_staged.json 2>>gpurun_out/r02c/rot_l${L}.err
done
cat gpurun_out/r02c/rot_*.json
