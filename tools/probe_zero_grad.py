"""Development probe: what fraction of the samples of a bench iteration carry an exactly-zero d_sigma, and how are they laid out
along the rays?  (Decides whether compacting the samples before encode_backward would pay.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from loner_amd import ops

args = bench.parse() if hasattr(bench, "parse") else None
seen = {}
orig = ops.density_backward
def hook(spec, p, d_sigma, *a, **k):
    d = d_sigma.detach()
    nz = d != 0
    seen["frac_nonzero"] = float(nz.float().mean())
    n, s = d.shape
    w = nz.reshape(n, s // 64, 64)
    seen["frac_waves_any"] = float(w.any(-1).float().mean())
    seen["frac_lanes_in_live_waves"] = float(w.float().sum() / max(float(w.any(-1).float().sum()) * 64, 1))
    first = torch.where(nz.any(1), nz.float().argmax(1), torch.full((n,), -1, device=d.device))
    last = torch.where(nz.any(1), s - 1 - nz.flip(1).float().argmax(1), torch.full((n,), -1, device=d.device))
    seen["mean_first"] = float(first.float().mean()); seen["mean_last"] = float(last.float().mean())
    seen["frac_nonzero_inside_span"] = float(nz.float().sum() / max(float((last - first + 1).clamp(min=0).sum()), 1))
    return orig(spec, p, d_sigma, *a, **k)
ops.density_backward = hook
import loner_amd.mapping.optimizer as O
O.ops.density_backward = hook
sys.argv = ["bench.py", "--steps", "12", "--warmup", "3", "--no-cpu-baseline"]
try:
    bench.main()
except SystemExit:
    pass
print("PROBE", seen)
