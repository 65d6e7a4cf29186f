mkdir -p gpurun_out/r02f
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r02f/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02f/pytest.log
tail -12 gpurun_out/r02f/pytest.log
timeout 600 python bench.py > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err
cut -c1-600 gpurun_out/r02f/bench.json
export SCRI_AMD_BENCH_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r02f/bench_gloo2_cfg4.json 2> gpurun_out/r02f/bench_gloo2.err
tail -1 gpurun_out/r02f/bench_gloo2_cfg4.json | cut -c1-1500
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --overlap-halo --no-n1-reference > gpurun_out/r02f/bench_gloo2_cfg4_overlap.json 2>> gpurun_out/r02f/bench_gloo2.err
tail -1 gpurun_out/r02f/bench_gloo2_cfg4_overlap.json | cut -c1-700
tail -5 gpurun_out/r02f/bench_gloo2.err
