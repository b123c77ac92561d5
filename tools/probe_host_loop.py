"""Development probe: where does the host spend its time in one mapping iteration?  cProfile over the timed optimisation loop of
bench.py with a GPU workload small enough that the GPU is idle (the wall time is the host's)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import loner_amd.mapping.optimizer as O
kf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
STEPS = 300
orig = O.Optimizer._do_iterate_optimizer
def wrapped(self, *a, **k):
    n = k["optimizer_settings"].num_iterations if "optimizer_settings" in k else 0
    if n != STEPS:
        return orig(self, *a, **k)
    pr = cProfile.Profile(); pr.enable()
    r = orig(self, *a, **k)
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
    print("HOSTPROF", s.getvalue()[:7000]); wrapped.done = True
    return r
O.Optimizer._do_iterate_optimizer = wrapped
sys.argv = ["bench.py", "--keyframes", str(kf), "--rays", "16", "--samples", "64", "--steps", str(STEPS), "--warmup", "20", "--no-cpu-baseline"]
try:
    bench.main()
except SystemExit:
    pass
