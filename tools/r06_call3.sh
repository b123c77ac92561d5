#!/bin/bash
# round 6, call 3: fused frequency kernels (lane = frequency mod 4, hardware sine) - parity + north-star leg; lnr_comm at RCCL world size 1; v_cndmask after one compare
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/valu_rate.bin > gpurun_out/r06_valu_rate.txt 2>&1; grep -E " 8 wave.*(cndmask|v_cmp|s_mov)" gpurun_out/r06_valu_rate.txt
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "density or fp16 or freq or partition or known or reproducible or frozen or smoke or sharded or rccl" > gpurun_out/pytest_gpu_subset.log 2>&1; tail -8 gpurun_out/pytest_gpu_subset.log
timeout 600 python tools/probe_ns.py > gpurun_out/r06_probe_ns.txt 2>&1; tail -2 gpurun_out/r06_probe_ns.txt
timeout 900 python tools/probe_sharded_overhead.py > gpurun_out/r06_sharded_overhead_world1.txt 2>&1; grep "per iteration" gpurun_out/r06_sharded_overhead_world1.txt || tail -20 gpurun_out/r06_sharded_overhead_world1.txt
