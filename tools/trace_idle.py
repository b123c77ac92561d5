"""Period, kernel time and GPU idle time per iteration from a rocprofv3 kernel trace of ONE loop (tools/probe_sharded_overhead.py --only i):
iterations are delimited by consecutive los_loss_fused_kernel launches, the last `--last` of them are averaged; idle = period minus the
union of the kernels' busy intervals (kernels on different queues may overlap)."""
import argparse, collections, csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--last", type=int, default=30)
ap.add_argument("--title", default="")
a = ap.parse_args()
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.trace))), key=lambda r: r[0])
marks = [i for i, r in enumerate(rows) if "los_loss_fused_kernel" in r[2]]
marks = marks[-(a.last + 1):]
n_it = len(marks) - 1
span = rows[marks[0]:marks[-1]]
period = (rows[marks[-1]][0] - rows[marks[0]][0]) / n_it / 1e3
busy, cur_s, cur_e = 0, None, None
for s, e, _ in span:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in span:
    name = k.split("(")[0].split("<")[0].replace("void ", "")
    per[name][0] += 1; per[name][1] += e - s
ksum = sum(v[1] for v in per.values()) / n_it / 1e3
print(f"--- {a.title}: period {period:.1f} us, kernel time {ksum:.1f} us, GPU busy {busy / n_it / 1e3:.1f} us, GPU idle {period - busy / n_it / 1e3:.1f} us, "
      f"launches per iteration {len(span) / n_it:.1f}")
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"    {k:40s} {v[0] / n_it:5.1f} launches  {v[1] / n_it / 1e3:8.2f} us")
