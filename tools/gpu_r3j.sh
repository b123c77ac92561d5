#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mapping.py tests/test_gpu_sharded.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for kf in 8 1; do
  echo "== keyframes $kf"
  timeout 300 python bench.py --keyframes $kf --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/j.err | python tools/bench_kernels.py | head -2
done
