#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
export LNR_LIB_PATH=$PWD/loner_amd/_lib/libloner_hip_ablate.so
echo "== transposes in LDS"; bash tools/exp_gen_bwd.sh 2>&1 | tail -8 | head -3
echo "== gathers"; LNR_X_NO_WT=1 bash tools/exp_gen_bwd.sh 2>&1 | tail -8 | head -3
