#!/bin/bash
# round 6, call 8: 256 x n backward on the f16 pipe (parity + timing), kernel stats / PMC traffic of the quick bench command, 8-rank test
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "wide or fp16_mode_general or eight_ranks_default or front_to_back or density_backward_matches" > gpurun_out/pytest_gpu_subset.log 2>&1; tail -6 gpurun_out/pytest_gpu_subset.log
timeout 900 python tools/probe_wide_nets.py --only 256 > gpurun_out/r06_wide_networks.txt 2>&1; cat gpurun_out/r06_wide_networks.txt | tail -8
bash tools/gpu_run.sh "stats::--quick --steps 30 --warmup 10" keep:r06 "pmc:FETCH_SIZE" "pmc:WRITE_SIZE" 2>&1 | tail -30
python tools/traffic_from_pmc.py gpurun_out/r06_traffic.json | tail -12
