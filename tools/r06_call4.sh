#!/bin/bash
# round 6, call 4: whole GPU suite at the current head, default bench line (thread probe, headline-workload quality pair), early-termination potential
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rf > gpurun_out/pytest_gpu.log 2>&1; tail -12 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06_bench_default.log 2> gpurun_out/r06_bench_default.err; tail -3 gpurun_out/r06_bench_default.err
tail -1 gpurun_out/r06_bench_default.log | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('ms_per_step', l['ms_per_step'], 'value', l['value'])
print('cpu_baseline', {k: l['cpu_baseline'].get(k) for k in ('value','cores','threads_probe_ms_per_iter')})
mq=l.get('matched_quality',{})
print('matched', mq.get('matched'), mq.get('comparison_hip_vs_torch_rocm'))
print('headline', mq.get('headline_workload'))
print('render', {k: l.get('render',{}).get(k) for k in ('ms_per_scan','l1_depth_m_of_this_scan')})
print('north_star', l.get('north_star_network'))
print('roofline', {k: l['roofline'].get(k) for k in ('kernel','frac','avg_launch_ms','traffic','traffic_stale')})
"
timeout 900 python tools/probe_render_dead.py > gpurun_out/r06_render_dead.txt 2>&1; tail -5 gpurun_out/r06_render_dead.txt
