"""Development probe: forward / backward of the north-star network class (bench.north_star_network_leg) and of the other general fp16 shapes."""
import sys, json
sys.path.insert(0, '.')
import torch
import bench
print(json.dumps(bench.north_star_network_leg(4096, 512, "cuda")))
