#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_known_answer.py -m gpu -q -x -p no:cacheprovider -k "fp16 or known or density_forward or density_backward" 2>&1 | tail -4
python tools/probe_ns.py 2>/dev/null | tail -1
timeout 600 python -m pytest tests/test_gpu_mapping.py -m gpu -q -x -p no:cacheprovider -s -k "l1_depth_curve" 2>&1 | grep -E "^phase|passed|failed|Assertion" | head
timeout 300 python bench.py --mode render --steps 3 --warmup 30 2>gpurun_out/render.err | tee gpurun_out/render_f32.json | cut -c1-1800
timeout 300 python bench.py --mode render --steps 3 --warmup 30 --dtype f16 2>>gpurun_out/render.err | tee gpurun_out/render_f16.json | cut -c1-900
tail -3 gpurun_out/render.err
