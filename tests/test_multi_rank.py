"""The N > 1 path on CPU: two gloo ranks shard a batch by frame, each produces its frames'
saddle lists (here with the oracle standing in for the GPU chain -- the sharding and the
gather are what is under test), and rank 0 must end up with every frame's list, in frame
order, exactly as a single process computes them."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import ROOT

FRAMES_PER_RANK = 3
W, H = 192, 128


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame_saddles(idx):
    import sys
    sys.path.insert(0, ROOT)
    import aprilgrid_rs_amd  # noqa: F401
    from aprilgrid_rs_amd import synth
    from oracle import oracle as O
    f, _ = synth.render_frame(idx, W, H)
    s = O.refined_saddle_points(f.numpy())
    return np.stack([s["x"], s["y"], s["k"], s["theta"], s["phi"]], axis=1) if len(s) else np.zeros((0, 5), np.float32)


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import aprilgrid_rs_amd  # noqa: F401
    from aprilgrid_rs_amd import sharding
    lo, hi = sharding.shard_range(rank, world, FRAMES_PER_RANK)
    saddles, table = sharding.alloc_result_buffers(FRAMES_PER_RANK, "cpu")
    off = 0
    for i, g in enumerate(range(lo, hi)):
        s = _frame_saddles(g)
        saddles[off:off + len(s)] = torch.from_numpy(s)
        table[i] = torch.tensor([len(s), off, 0, 0], dtype=torch.int32)
        off += len(s)
    gs, gt = sharding.gather_results(saddles, table, dst=0)
    # the same through the double-buffered asynchronous pipeline bench.py uses: 3 "steps", the
    # buffers of the last one must arrive intact
    # one slab per gather; several steps per gather with the last group full (k = 3, 6 steps), not full (k = 4: 5 and 9 steps:
    # the second pass over the buffers), a single step; every n-th step
    for kw, steps in (({}, 3), ({"steps_per_gather": 3}, 6), ({"steps_per_gather": 4}, 5), ({"steps_per_gather": 4}, 9),
                      ({"steps_per_gather": 4}, 1), ({"every": 2}, 5)):
        pipe = sharding.GatherPipeline(FRAMES_PER_RANK, "cpu", dst=0, **kw)
        for step in range(steps):
            ps, pt = pipe.acquire()
            ps.zero_(); pt.zero_()
            ps.copy_(saddles * (1.0 if step == steps - 1 else 0.5))  # only the LAST step carries the real records
            pt.copy_(table)
            pipe.submit()
        ps_all, pt_all = pipe.finish()
        if rank == 0:
            for r in range(world):
                assert torch.equal(pt_all[r], gt[r]) and torch.equal(ps_all[r], gs[r]), "pipeline gather differs (%s, %d steps)" % (kw, steps)
        else:
            assert ps_all is None and pt_all is None
    if rank == 0:
        frames = []
        for r in range(world):
            frames += sharding.unpack_frames(gs[r], gt[r])
        q.put([f.tobytes() for f in frames])
    else:
        assert gs is None and gt is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == world * FRAMES_PER_RANK
    for g in range(world * FRAMES_PER_RANK):
        assert got[g] == _frame_saddles(g).astype(np.float32).tobytes(), "frame %d" % g


def test_shard_ranges_cover_the_batch_without_overlap():
    import aprilgrid_rs_amd  # noqa: F401
    from aprilgrid_rs_amd import sharding
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            lo, hi = sharding.shard_range(r, world, 256)
            seen += list(range(lo, hi))
        assert seen == list(range(256 * world))


def test_gather_is_a_noop_for_one_rank():
    import aprilgrid_rs_amd  # noqa: F401
    from aprilgrid_rs_amd import sharding
    s, t = sharding.alloc_result_buffers(2, "cpu")
    gs, gt = sharding.gather_results(s, t)
    assert gs[0] is s and gt[0] is t
    t[0] = torch.tensor([0, 0, 4, 0], dtype=torch.int32)  # overflow status -> None
    assert sharding.unpack_frames(s, t)[0] is None and len(sharding.unpack_frames(s, t)[1]) == 0
