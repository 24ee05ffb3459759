#!/bin/bash
for st in 20 20; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps $st --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"value_api": [0-9.]*'
python bench.py --steps $st --warmup 3 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"value_api": [0-9.]*'
done
