"""CPU-side checks of bench.py: the cpu_baseline leg (the oracle timed on host cores) and the
command-line contract.  The GPU legs are exercised by the driver's bench run."""
import os
import subprocess
import sys

import numpy as np

from tests.util import synth_module

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_leg_reports_the_contract_fields():
    sys.path.insert(0, ROOT)
    import bench
    synth = synth_module()
    frames = np.stack([np.asarray(synth.render_frame(i, 320, 240)[0]) for i in range(2)])
    r = bench.cpu_baseline(frames, "L8", 0.5)
    assert r["unit"] == "Mpix/s" and r["kind"] == "port" and r["cores"] == 1 and r["value"] > 0
    assert "oracle/agx_oracle.c" in r["sample"]
    assert r["all_cores"]["cores"] >= 1 and r["all_cores"]["value"] > 0


def test_bench_refuses_multi_gpu_without_the_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env={**os.environ, "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)
