mkdir -p gpurun_out/r02e
for G in 1 2 3 4; do
for L in 16 8; do
echo "G=$G L=$L" >> gpurun_out/r02e/groups.txt
SCRI_AMD_ROTATE_GROUPS=$G CPU_BASELINE=0 timeout 300 python tools/bench_rotation.py $L 100000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_times'], d['kernel_ms'], d['roofline']['frac'])" >> gpurun_out/r02e/groups.txt
SCRI_AMD_ROTATE_GROUPS=$G CPU_BASELINE=0 timeout 300 python tools/bench_rotation.py $L 1000000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_times'], d['kernel_ms'], d['roofline']['frac'])" >> gpurun_out/r02e/groups.txt
done; done
cat gpurun_out/r02e/groups.txt
