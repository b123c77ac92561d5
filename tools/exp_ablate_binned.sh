#!/bin/bash
# Development: where the time of the binned encode-backward kernels goes.  Library built with
#   LNR_BUILD_TAG=ablate LNR_EXTRA_HIPCC_FLAGS=-DLNR_ABLATE python -m loner_amd.build
# (results are WRONG by construction, only the kernel times mean something).
#   LNR_X_LEVELS: bit mask of the levels that run (0xFF00: the 8 x-pair levels, 0x00F8: the 5 binned pair-record levels, 0x7: the dense ones)
#   LNR_X_DBG bits: 1 no per-ray accumulation, 2 no d/dx arithmetic, 4 no table gathers, 8 no line stores, 16 no partition at all,
#                   32 no copy-out phase, 64 no rank atomics (fake offsets), 128 no staging writes, 256 no first barrier
export TMPDIR=/tmp LNR_LIB_PATH=$PWD/loner_amd/_lib/libloner_hip_ablate.so
run() { printf "%-40s" "$*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels.py --all | grep -E "kernel encode_backward " ; }
for lv in 0xFF00 0x00F8 0x0007; do
  for d in 0 1 3 7 8 16 32 40 64 192 23; do run LNR_X_LEVELS=$lv LNR_X_DBG=$d; done
done
