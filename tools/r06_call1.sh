#!/bin/bash
# round 6, call 1: VALU issue-rate micro-benchmark, per-phase cycles of encode_backward (LNR_PHASE_TIMING build), baseline bench at HEAD
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/valu_rate.bin > gpurun_out/r06_valu_rate.txt 2>&1; tail -45 gpurun_out/r06_valu_rate.txt
LNR_PHASE_TIMING=1 LNR_LIB_PATH=$PWD/loner_amd/_lib/libloner_hip_phase.so timeout 600 python bench.py --quick --steps 6 --warmup 2 > gpurun_out/r06_phases.out 2> gpurun_out/r06_phases.err
grep "lnr phases" gpurun_out/r06_phases.err | tail -40
timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/r06_bench_head.log 2> gpurun_out/r06_bench_head.err
tail -1 gpurun_out/r06_bench_head.log | python tools/bench_kernels.py --all | head -40
timeout 600 python tools/probe_sharded_overhead.py > gpurun_out/r06_sharded_head.txt 2>&1; tail -6 gpurun_out/r06_sharded_head.txt
