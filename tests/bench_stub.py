"""CPU stand-in for the device side of bench.py (AGX_BENCH_STUB=1): the gloo backend, CPU tensors, wall-clock
"events", and a detector whose enqueue call fills the caller's result buffers from the oracle.  What is under test
is bench.main()'s N > 1 control flow -- every rank must issue the same sequence of collectives (settle-round
broadcast, per-step gathers, barriers, the all_reduce of the step time) and rank 0 must find every rank's frames in
the gathered tables -- not the chain."""
import time

import numpy as np


class _Event:
    def __init__(self):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _Family:
    T36H11 = 3


class StubDetector:
    """TagDetector's batch interface on CPU tensors: saddles_batch_enqueue_to packs the oracle's saddle lists of the
    frames back to back and writes the frame table (count, offset, status, clusters), as the chain does."""
    calls = 0

    def __init__(self, tag_family, params=None, device=0):
        self._opts = {}
        self._ms = 0.0
        self._n = 0

    def saddles_batch_enqueue_to(self, frames, out_saddles, frame_table):
        import torch
        from oracle import oracle as O
        t0 = time.perf_counter()
        host = frames.numpy()
        off = 0
        for i in range(host.shape[0]):
            s = O.refined_saddle_points(host[i])
            a = np.stack([s["x"], s["y"], s["k"], s["theta"], s["phi"]], axis=1) if len(s) else np.zeros((0, 5), np.float32)
            out_saddles[off:off + len(a)] = torch.from_numpy(a.astype(np.float32))
            frame_table[i] = torch.tensor([len(a), off, 0, 0], dtype=torch.int32)
            off += len(a)
        self._ms += (time.perf_counter() - t0) * 1e3
        self._n += 1
        StubDetector.calls += 1

    def set_option(self, name, value):
        self._opts[name] = value

    def get_option(self, name):
        return self._opts.get(name, 0)

    def profile_enable(self, on):
        pass

    def profile_reset(self):
        self._ms, self._n = 0.0, 0

    def profile_read(self):
        return {"k_blur_hessian": (self._ms, max(self._n, 1)), "k_sparse_stub": (0.0, 1)}

    def close(self):
        pass


class _Module:
    TagDetector = StubDetector
    TagFamily = _Family


class StubRuntime:
    backend = "gloo"

    def __init__(self, torch):
        self.torch = torch
        self.A = _Module
        self.detector_cls = StubDetector

    def device(self, local_rank):
        return self.torch.device("cpu")

    def init_process_group(self, dist, dev):
        dist.init_process_group(self.backend)

    def synchronize(self, dev):
        pass

    def event(self):
        return _Event()
