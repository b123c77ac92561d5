"""Development probe: one north-star network leg at a given ray count (for rocprofv3 --kernel-trace --stats: which kernels a call is made of)."""
import sys
sys.path.insert(0, '.')
import bench
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
r = bench.north_star_network_leg(rays, 512, "cuda")
print(rays * 512, r["forward_ms"], r["backward_ms"], flush=True)
