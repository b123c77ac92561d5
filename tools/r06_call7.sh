#!/bin/bash
# round 6, call 7: whole GPU suite; kernel stats (rocprofv3 --kernel-trace --stats) and PMC traffic passes of the bench command; driver command
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rf > gpurun_out/pytest_gpu.log 2>&1; tail -6 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -rf gpurun_out/prof_product
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_product -o run -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/prof_product.log 2>&1
f=$(find gpurun_out/prof_product -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06_kernel_stats_timed_region.csv; head -14 gpurun_out/r06_kernel_stats_timed_region.csv | cut -c1-160
bash tools/pmc.sh
python tools/traffic_from_pmc.py gpurun_out/r06_traffic.json "" | tail -12
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_command.log 2> gpurun_out/r06_bench_driver_command.err; tail -1 gpurun_out/r06_bench_driver_command.log | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('driver command: ms_per_step', l['ms_per_step'], 'value', l['value'], 'roofline frac', l['roofline']['frac'], 'stale', l['roofline'].get('traffic_stale'))"
