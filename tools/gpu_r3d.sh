#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mapping.py -m gpu -q -x -p no:cacheprovider -k "partition or binned or full_size or reproducible or frozen" 2>&1 | tail -4
for v in "A=1" "LNR_BINS_W8=1" "LNR_NO_BINS=1"; do
  echo "== $v"
  env $v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>gpurun_out/d.err | python tools/bench_kernels.py | head -3
done
echo "== 1 keyframe"; timeout 300 python bench.py --keyframes 1 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels.py
