#!/bin/bash
# SQ-side counters (issue/wait breakdown) per kernel; kernel-trace only, separate from any other tracing
export TMPDIR=/tmp
SET=${1:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"}
NAME=${2:-sq}
rm -rf gpurun_out/pmc_$NAME
timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmc_$NAME -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$NAME.log 2>&1
echo "exit $?"; find gpurun_out/pmc_$NAME -name "*.csv" | head -5
