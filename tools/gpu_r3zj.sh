#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/rend
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rend -o p -- python bench.py --mode render --steps 3 --warmup 5 > gpurun_out/rend.log 2>&1
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/rend/*kernel_stats.csv")[0])))
for r in rows[:10]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}%")
PY
