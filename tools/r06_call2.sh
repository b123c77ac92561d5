#!/bin/bash
# round 6, call 2: fused frequency kernels (parity + north-star leg), d/dx-behind-barrier-1 A/B, v_cndmask variants
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/valu_rate.bin > gpurun_out/r06_valu_rate.txt 2>&1; grep -E " 8 wave|v_sin_f32 over" gpurun_out/r06_valu_rate.txt
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "density or fp16 or freq or partition or known or reproducible or frozen or smoke" > gpurun_out/pytest_gpu_subset.log 2>&1; tail -15 gpurun_out/pytest_gpu_subset.log
for tag in "" dxlate ""; do
  echo "===== quick bench, library tag '$tag'"
  LNR_LIB_PATH=$([ -n "$tag" ] && echo $PWD/loner_amd/_lib/libloner_hip_$tag.so) timeout 300 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python tools/bench_kernels.py --all | grep -E "ms_per_step|kernel " | head -8
done
timeout 600 python tools/probe_ns.py > gpurun_out/r06_probe_ns.txt 2>&1; tail -3 gpurun_out/r06_probe_ns.txt
