#!/bin/bash
# round 6, call 27: final validation with encode_backward's d/dx-before-copy-out as the default
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_run.sh "stats::--quick --steps 30 --warmup 10" keep:r06 "pmc:FETCH_SIZE" "pmc:WRITE_SIZE" > gpurun_out/r06_final_pmc.log 2>&1; tail -4 gpurun_out/r06_final_pmc.log
LNR_COMMIT=$1 python tools/traffic_from_pmc.py gpurun_out/r06_traffic.json | tail -3
cp gpurun_out/r06_traffic.json profiles/traffic.json
bash tools/gpu_run.sh tests smoke 2>&1 | tail -6
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_command.log 2> gpurun_out/r06_bench_driver_command.err; tail -1 gpurun_out/r06_bench_driver_command.log | cut -c1-160
