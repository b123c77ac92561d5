#!/bin/bash
# functional check of the multi-rank benchmark path on a 1-GPU box: two ranks share the GPU and talk over gloo
export LNR_DIST_BACKEND=gloo TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 10 --warmup 3 2>gpurun_out/bench_gloo2.err | tail -1 | cut -c1-600
echo "exit $?"
