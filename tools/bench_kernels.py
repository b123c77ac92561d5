"""Print the per-kernel / per-op tables of a bench.py JSON line (stdin)."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    j = json.loads(line)
    r = j["roofline"]
    print(f"ms_per_step {j['ms_per_step']:.3f}  value {j['value']:.0f} {j['unit']}  dtype {j['dtype']}")
    print(f"  {r['kernel']:28s} {r['avg_launch_ms']:.4f} ms")
    for k, v in r.get("secondary", {}).items():
        print(f"  {k:28s} {v['avg_ms']:.4f} ms  x{v['calls']}")
    if "--all" in sys.argv:
        for k, v in sorted(j.get("kernels_ms", {}).items(), key=lambda kv: -kv[1]):
            print(f"    kernel {k:28s} {v:.4f}")
        for k, v in sorted(j.get("ops_ms", {}).items(), key=lambda kv: -kv[1]):
            print(f"    op     {k:28s} {v:.4f}")
