#!/bin/bash
# round 6, call 6: 16-byte x-pair gathers in the forward (A/B), 256-thread partition workgroups (A/B), parity subset, render leg
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "density or front_to_back or partition or reproducible or known or fp16 or golden or checkpoint or l1_depth or compact" > gpurun_out/pytest_gpu_subset.log 2>&1; tail -5 gpurun_out/pytest_gpu_subset.log
for tag in "" x16off b256 "" x16off b256; do
  echo "===== quick bench, library tag '$tag'"
  LNR_LIB_PATH=$([ -n "$tag" ] && echo $PWD/loner_amd/_lib/libloner_hip_$tag.so) timeout 300 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python tools/bench_kernels.py --all | grep -E "ms_per_step|kernel (encode|table)" | head -6
done
for tag in "" x16off; do
  echo "===== render leg, library tag '$tag'"
  LNR_LIB_PATH=$([ -n "$tag" ] && echo $PWD/loner_amd/_lib/libloner_hip_$tag.so) timeout 300 python bench.py --mode render --steps 3 --warmup 110 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: l.get(k) for k in ('ms_per_scan','l1_depth_m_of_this_scan','kernels_ms_per_scan')}); print('ftb', {k: l.get('front_to_back',{}).get(k) for k in ('ms_per_scan','l1_depth_m_of_this_scan')})"
done
