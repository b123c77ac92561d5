#!/bin/bash
# round 6, call 10: 256 x n f16 backward after pipelining (parity + per-kernel times + probe)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "wide or fp16_mode_general" > gpurun_out/pytest_gpu_subset.log 2>&1; tail -4 gpurun_out/pytest_gpu_subset.log
rm -rf gpurun_out/prof_wide
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide -o run -- python tools/probe_wide_nets.py --only "freq12 -> 256 x 2" --prec fp16 > gpurun_out/prof_wide.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_wide/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:11]:
    print(f"{r['Name'][:48]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
timeout 900 python tools/probe_wide_nets.py --only 256 > gpurun_out/r06_wide_networks.txt 2>&1; tail -8 gpurun_out/r06_wide_networks.txt
