#!/bin/bash
# round 6, call 29: A/B on one box - encode_backward's batch loop specialised by point source (product) against the previous commit's library (prev)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
  for tag in prev ""; do
    lib=""; [ -n "$tag" ] && lib=$PWD/loner_amd/_lib/libloner_hip_$tag.so
    echo "== ${tag:-product}"
    LNR_LIB_PATH=$lib timeout 300 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | python tools/bench_kernels.py --all | grep -E "ms_per_step|kernel (encode_backward)"
  done
done
