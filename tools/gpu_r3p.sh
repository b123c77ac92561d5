#!/bin/bash
export TMPDIR=/tmp LNR_LIB_PATH=$PWD/loner_amd/_lib/libloner_hip_ablate.so
for lv in 0xFFFF 0xFF00 0x00F8 0x0007; do
  printf "%-10s" $lv; LNR_X_LEVELS=$lv timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels.py --all | grep -E "kernel (table_grad_reduce|encode_backward) " | tr '\n' ' '; echo
done
