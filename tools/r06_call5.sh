#!/bin/bash
# round 6, call 5: whole GPU suite, default bench line (timing diagnostics), north-star leg, sharded overhead
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rf -x > gpurun_out/pytest_gpu.log 2>&1; tail -8 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r06_bench_default.log 2> gpurun_out/r06_bench_default.err; echo "bench exit $?"; grep "bench.py:" gpurun_out/r06_bench_default.err | cut -c1-300
tail -1 gpurun_out/r06_bench_default.log | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('ms_per_step', l['ms_per_step'], 'value', l['value'])
print('cpu_baseline', {k: l['cpu_baseline'].get(k) for k in ('value','cores','available_cores','threads_probe_ms_per_iter')})
mq=l.get('matched_quality',{})
print('matched', mq.get('matched'), mq.get('comparison_hip_vs_torch_rocm'))
print('headline', mq.get('headline_workload'))
r=l.get('render',{})
print('render', {k: r.get(k) for k in ('ms_per_scan','l1_depth_m_of_this_scan','front_to_back')})
print('north_star', {k: l.get('north_star_network',{}).get(k) for k in ('forward_ms','backward_ms')})
print('kernels', l.get('kernels_ms'))
"
timeout 600 python tools/probe_sharded_overhead.py > gpurun_out/r06_sharded_overhead_world1.txt 2>&1; grep "per iteration" gpurun_out/r06_sharded_overhead_world1.txt
