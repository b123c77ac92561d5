#!/bin/bash
# round 6, call 19: encode_backward at two waves per SIMD instead of four (LDS padding: one workgroup per CU)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for pad in 0 24576; do
  echo "LNR_ENC_BWD_LDS_PAD=$pad"
  LNR_ENC_BWD_LDS_PAD=$pad timeout 300 python bench.py --quick --steps 20 --warmup 5 2>/dev/null | python tools/bench_kernels.py --all | grep -E "ms_per_step|kernel (encode|table)"
done
