#!/bin/bash
# Development: ablation timings of encode_backward_kernel.  Needs a library built with LNR_EXTRA_HIPCC_FLAGS=-DLNR_ABLATE
# (python -m loner_amd.build); results are WRONG by construction, only the kernel times mean something.
#   LNR_X_DBG bits: 1 no per-ray accumulation, 2 no d/dx arithmetic, 4 no table gathers, 8 no record stores, 16 no partition at all,
#                   32 plain instead of streaming loads of the d_feature planes, 128 no samples at all (workgroup overhead only)
#   LNR_X_LEVELS:   bit mask of the record levels to run
export TMPDIR=/tmp LNR_EXTRA_HIPCC_FLAGS=-DLNR_ABLATE
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python tools/bench_kernels.py --all | grep -E "kernel encode_backward "; }
for d in 0 256 512 1024 2048; do run LNR_X_DBG=$d; done
