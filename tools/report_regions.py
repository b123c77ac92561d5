"""How full the table-gradient record regions run on the bench workload (flag LNR_BWD_REPORT_REGIONS of lnr_density_backward):
one mapping iteration with the report switched on for its density backward.  python tools/report_regions.py [--rays N --samples S]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline",)]
import bench                                                   # noqa: E402
from loner_amd import ops                                      # noqa: E402

if __name__ == "__main__":
    args = bench.parse()
    orig = ops.density_backward
    state = {"on": False}
    ops.density_backward = lambda *a, **k: orig(*a, **dict(k, report_regions=state["on"]))
    from loner_amd.common.pose_utils import WorldCube
    from loner_amd.common.settings import default_optimizer_settings
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.utils import synthetic as SY
    scale, shift = SY.world_cube()
    s = default_optimizer_settings(log_directory="/tmp/loner_amd_report")
    s["num_samples"]["lidar"], s["num_samples"]["sky"] = args.rays, 0
    s["model_config"]["model"]["render"]["N_samples_train"] = args.samples
    torch.manual_seed(0)
    opt = Optimizer(s, None, WorldCube(torch.tensor(scale), torch.from_numpy(shift)), 0, False, True, False)
    window = bench.build_window(args.keyframes)
    if args.warmup > 0:            # train silently first: the report is wanted for a map that has formed, not for iteration 0
        opt._do_iterate_optimizer(window, [None], optimizer_settings=OptimizationSettings(args.warmup, False, False, False, True))
    state["on"] = True
    opt._do_iterate_optimizer(window, [None], optimizer_settings=OptimizationSettings(max(args.steps, 1), False, False, False, True))
    torch.cuda.synchronize()
