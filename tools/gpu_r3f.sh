#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_known_answer.py -m gpu -q -x -p no:cacheprovider -k "fp16 or known" 2>&1 | tail -4
bash tools/exp_gen_bwd.sh 2>&1 | tail -16
bash tools/exp_gen_bwd.sh siren 2>&1 | tail -8
