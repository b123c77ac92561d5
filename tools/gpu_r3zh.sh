#!/bin/bash
# sampler change: parity (bit-identical fixtures) + bench kernel times + render leg
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "sampl or occ or ray or render or mapping or l1" > gpurun_out/r3zh_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3zh_tests.log
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k:d['kernels_ms'].get(k) for k in ('sample_rays_occ','compact_rays','encode_forward')}, d['render']['ms_per_scan'], d['render']['kernels_ms_per_scan'].get('sample_rays_occ'))"; done
