#!/bin/bash
# session script: the slab pass with 3 gloo ranks on one GPU
cd "$(dirname "$0")/.."
for envs in "FTKX_DIST_CELLS=2" "FTKX_DIST_CELLS=2 FTKX_BENCH_X=--no-pipeline"; do
  echo "=== env: $envs" 
  env $envs timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 3 --config small3 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-streaming-tracker --no-other-scaling --backend gloo --single-device 2>&1 | grep -v "^W\|^\[W\|amdgpu.ids" | grep -v "rank[12]\]" | tail -60
done
