#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/gen_bwd
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gen_bwd -o t -- python tools/probe_gen_bwd.py $1 2>/dev/null | grep "backward ms"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gen_bwd/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-14:]
for r in last:
    print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:8.3f} ms  {r["Kernel_Name"][:80]}')
PY
