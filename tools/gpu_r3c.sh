#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_mapping.py -m gpu -q -x -p no:cacheprovider -s -k "partition or binned or full_size or reproducible or frozen or l1_depth_curve or fused_loss" 2>&1 | grep -v Warning | grep "phase\|passed\|failed\|Error\|assert" | tail -15
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>gpurun_out/c.err | python tools/bench_kernels.py --all | head -24
