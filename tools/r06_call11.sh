#!/bin/bash
# round 6, call 11: 256 x n fp16 backward in its kept form (f16 weight gradient + first-layer back-propagation, fp32 hidden back-propagation); RCCL world-size-1 tests
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "wide or fp16_mode_general or rccl" > gpurun_out/pytest_gpu_subset.log 2>&1; tail -4 gpurun_out/pytest_gpu_subset.log
timeout 900 python tools/probe_wide_nets.py --only 256 > gpurun_out/r06_wide_networks_kept.txt 2>&1; tail -8 gpurun_out/r06_wide_networks_kept.txt
