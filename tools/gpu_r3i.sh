#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/bench_default.log") if x.startswith("{")][-1]
j=json.loads(l)
print("ms_per_step", j["ms_per_step"], "value", j["value"])
print(json.dumps(j.get("matched_quality"), indent=1)[:2500])
print("cpu", {k:j["cpu_baseline"].get(k) for k in ("value","ms_per_iter","iterations","cores","cpu_model","l1_depth_m_before","l1_depth_m_after")})
print("render", {k:j["render"].get(k) for k in ("value","ms_per_scan")} if j.get("render") else None)
print("speedups", j.get("speedup_vs_cpu_oracle_same_workload"), j.get("speedup_vs_torch_rocm_oracle_same_workload"))
PY
