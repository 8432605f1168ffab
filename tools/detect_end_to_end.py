"""bench.py's detect_end_to_end leg alone (agx_detect_batch over configs[1]'s frames in pageable host memory, by
thread count): python tools/detect_end_to_end.py [repeat]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aprilgrid_rs_amd as A
import bench
dev = torch.device("cuda", 0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    print(json.dumps(bench.detect_end_to_end_leg(torch, A, dev, 256, 1280, 800)), flush=True)
