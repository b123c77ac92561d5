#!/bin/bash
# round 6, call 23: final validation at HEAD - PMC traffic restamp, full GPU suite, smoke, default bench
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_run.sh "stats::--quick --steps 30 --warmup 10" keep:r06 "pmc:FETCH_SIZE" "pmc:WRITE_SIZE" > gpurun_out/r06_final_pmc.log 2>&1; tail -4 gpurun_out/r06_final_pmc.log
LNR_COMMIT=$1 python tools/traffic_from_pmc.py gpurun_out/r06_traffic.json | tail -3
cp gpurun_out/r06_traffic.json profiles/traffic.json
bash tools/gpu_run.sh tests smoke "bench::" 2>&1 | tail -34
