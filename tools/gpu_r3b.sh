#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mapping.py -m gpu -q -x -p no:cacheprovider -s -k "l1_depth_curve or failure" 2>&1 | grep -v Warning | tail -15
bash tools/exp_ablate_binned.sh 2>&1 | tee gpurun_out/ablate_binned.txt
