#!/bin/bash
# Development: ablation timings of encode_backward_kernel.  Needs a library built with LNR_EXTRA_HIPCC_FLAGS=-DLNR_ABLATE
# (python -m loner_amd.build); results are WRONG by construction, only the kernel times mean something.
#   LNR_X_DBG bits: 1 no per-ray accumulation, 2 no d/dx arithmetic, 4 no table gathers, 8 no record stores, 16 no partition at all,
#                   32 plain instead of streaming loads of the d_feature planes
#   LNR_X_LEVELS:   bit mask of the record levels to run
export TMPDIR=/tmp LNR_EXTRA_HIPCC_FLAGS=-DLNR_ABLATE
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels.py | grep -E "encode_backward|table_grad"; }
for d in 0 1 2 4 8 16; do run LNR_X_DBG=$d; done
run LNR_X_LEVELS=0x00fe
run LNR_X_LEVELS=0xff00
