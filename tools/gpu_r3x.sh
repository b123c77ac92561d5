#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python tools/bench_kernels.py | head -2
