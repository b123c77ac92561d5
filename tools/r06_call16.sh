#!/bin/bash
# round 6, call 16: per-phase cycles of the fused frequency kernels after the restructuring
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
LNR_PHASE_TIMING=1 LNR_LIB_PATH=$PWD/loner_amd/_lib/libloner_hip_phase.so timeout 300 python tools/probe_ns_trace.py 4096 > gpurun_out/r06_ns_phases.txt 2>&1 < /dev/null
grep "lnr phases" gpurun_out/r06_ns_phases.txt | tail -16; grep "mlp_forward" gpurun_out/r06_ns_phases.txt | tail -6
