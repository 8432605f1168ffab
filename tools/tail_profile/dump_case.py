"""Oracle saddles + u8 luma of one frame for tools/tail_profile (synthetic frame index or fixture image name)."""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
out, what = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "0")
if what.endswith(".png"):
    from tests.util import load_image
    img = load_image(what)
else:
    import aprilgrid_rs_amd  # noqa: F401
    from aprilgrid_rs_amd import synth
    img = np.asarray(synth.render_frame(int(what), 1280, 800)[0])
s, g = O.refined_saddle_points(img), O.luma_u8(img)
with open(out, "wb") as f:
    f.write(struct.pack("iii", g.shape[1], g.shape[0], len(s)))
    f.write(s.tobytes())
    f.write(g.tobytes())
print(what, g.shape, len(s), "saddles")
