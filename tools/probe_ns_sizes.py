"""Development probe: the north-star network leg at several sample counts (fixed cost of a launch = weights into LDS, vs the per-step cost)."""
import sys, json
sys.path.insert(0, '.')
import torch
import bench
for rays in (128, 1024, 4096, 4096, 4096, 16384):
    r = bench.north_star_network_leg(rays, 512, "cuda")
    print(rays * 512, r["forward_ms"], r["backward_ms"], flush=True)
