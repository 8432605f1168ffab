import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
if order == "torch_first":
    import torch; print("torch avail", torch.cuda.is_available())
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import _ffi
_ffi.lib()
if order == "lib_first":
    import torch; print("torch avail", torch.cuda.is_available())
try:
    d = A.TagDetector("t36h11"); print(order, "create OK"); d.close()
except Exception as e:
    print(order, "FAILED:", e)
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
