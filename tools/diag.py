import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import aprilgrid_rs_amd as A
det = A.TagDetector("t36h11")
img = np.random.default_rng(1).integers(0, 256, (2, 2), dtype=np.uint8)
print(det.refined_saddle_points(img, as_array=True))
print(det.debug_fetch(0, "counters"))
print(det.debug_fetch(0, "resp", (2, 2)))
