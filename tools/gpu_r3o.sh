#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "partition or full_size or binned" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels.py | grep -E "ms_per_step|table_grad"; done
