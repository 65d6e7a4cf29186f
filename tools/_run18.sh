#!/bin/bash
export PYTHONPATH=.
SCRI_AMD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | tail -2 | cut -c1-1500
