"""bench.py's host_boundary leg alone: 256 frames in pinned host memory -> saddle lists in host memory, the next batch's upload in flight."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aprilgrid_rs_amd as A
import bench
dev = torch.device("cuda", 0)
for _ in range(int(os.environ.get("REPEAT", "3"))):
    print(json.dumps(bench.host_boundary_leg(torch, A, dev, int(os.environ.get("FRAMES", "256")), 1280, 800, int(os.environ.get("STEPS", "20")))))
