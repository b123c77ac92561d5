#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mapping.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "pipelined or reproducible or full_size or fused_loss or frozen or failure or deferred or optimizer_loop or fp16_mode_config5" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels.py | grep -E "ms_per_step"; done
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --dtype f16 2>/dev/null | python tools/bench_kernels.py | grep -E "ms_per_step"
