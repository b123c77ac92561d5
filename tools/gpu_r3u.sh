#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof_k1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_k1 -o t -- python bench.py --keyframes 1 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/k1.log 2>&1
python tools/bench_kernels.py < gpurun_out/k1.log | head -2
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_k1/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void encode_forward_kernel<2, false>")]
sel = starts[30:60]
span = busy = 0
for a, b in zip(sel[:-1], sel[1:]):
    seg = rows[a:b]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    cs, ce = iv[0]; bz = 0
    for s_, e_ in iv[1:]:
        if s_ > ce: bz += ce - cs; cs, ce = s_, e_
        else: ce = max(ce, e_)
    bz += ce - cs
    span += int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"]); busy += bz
n = len(sel) - 1
print("1 keyframe: span %.3f ms, GPU busy %.3f ms, idle %.3f ms per iteration, %d kernels" % (span / n / 1e6, busy / n / 1e6, (span - busy) / n / 1e6, sel[1] - sel[0]))
PY
python tools/trace_iteration.py $(find gpurun_out/prof_k1 -name "*kernel_trace.csv") 41
