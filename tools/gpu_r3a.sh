#!/bin/bash
# round 3, A/B of the binned partition: parity tests first, then the bench with and without bins
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mapping.py tests/test_known_answer.py -m gpu -q -x -p no:cacheprovider -k "partition or binned or guard or failure or full_size or reproducible or known or frozen" 2>&1 | tail -15
for v in "" 1; do
  echo "== LNR_NO_BINS=$v"
  LNR_NO_BINS=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>gpurun_out/ab_$v.err | python tools/bench_kernels.py
done
