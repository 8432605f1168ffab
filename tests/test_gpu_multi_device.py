"""One process, several detectors / devices: per-handle and per-device state of the
C ABI, handles driven from threads, and the C-level detector groups with their result gather.
Every test is parametrised by what the box has and passes with a single GPU (two ranks may share
device 0 with the peer-copy transport; the RCCL transport needs distinct devices and runs where
there are at least two)."""
import os
import threading

import numpy as np
import pytest

from tests.util import check_saddles, oracle_saddles_parallel, synth_module

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


def _frames(first, n, device, w=640, h=400):
    synth = synth_module()
    fr, _ = synth.render_batch(first, n, w, h, device=device)
    return fr


def test_one_handle_per_device_from_threads(oracle):
    """One TagDetector per visible device, each created and driven by its own thread of ONE process
    (plus two more handles on device 0): every result equals the oracle's.  Exercises the per-device
    kernel attributes (a process-wide flag used to configure device 0 only) and the per-thread
    current-device / error state."""
    import torch
    import aprilgrid_rs_amd as A
    n_dev = torch.cuda.device_count()
    jobs = [(d, 100 + 8 * d) for d in range(n_dev)] + [(0, 300), (0, 400)]
    results, errors = {}, []

    def work(slot, dev, first):
        try:
            det = A.TagDetector("t36h11", None, device=dev)
            fr = _frames(first, 3, "cuda:%d" % dev)
            torch.cuda.synchronize(dev)
            for _ in range(3):  # several batches per handle while the other threads run theirs
                det.saddles_batch_enqueue(fr)
                res, status = det.saddles_batch_fetch()
            assert (status == 0).all()
            results[slot] = (fr.cpu().numpy(), res)
            det.close()
        except Exception as e:  # noqa: BLE001
            errors.append((slot, repr(e)))

    threads = [threading.Thread(target=work, args=(i, d, f)) for i, (d, f) in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for slot, (host, res) in results.items():
        refs = oracle_saddles_parallel(oracle, host, threads=3)
        for i in range(len(res)):
            check_saddles(res[i], refs[i], "job %d frame %d" % (slot, i))


@pytest.mark.parametrize("ranks", [1, 2, 3])
def test_group_peer_transport(oracle, ranks):
    """agx_group_* with the peer-copy gather: `ranks` ranks over the visible devices (round-robin, so
    a one-GPU box runs them all on device 0).  The gathered, rank-major frame list must equal the
    oracle frame by frame."""
    import torch
    import aprilgrid_rs_amd as A
    n_dev = torch.cuda.device_count()
    devices = [r % n_dev for r in range(ranks)]
    grp = A.DetectorGroup("t36h11", devices, transport="peer")
    assert len(grp) == ranks
    frames = [_frames(50 + 4 * r, 4, "cuda:%d" % devices[r]) for r in range(ranks)]
    for d in set(devices):
        torch.cuda.synchronize(d)
    for rep in range(2):  # twice: slab reuse
        grp.saddles_enqueue(frames)
        res, status = grp.saddles_fetch()
    assert (status == 0).all() and len(res) == 4 * ranks
    host = np.concatenate([f.cpu().numpy() for f in frames])
    refs = oracle_saddles_parallel(oracle, host, threads=4)
    for i in range(len(res)):
        check_saddles(res[i], refs[i], "global frame %d" % i)
    grp.close()


def test_group_reports_overflow_per_frame(oracle):
    import torch
    import aprilgrid_rs_amd as A
    grp = A.DetectorGroup("t36h11", [0, 0], transport="peer")
    frames = [_frames(9, 2, "cuda:0"), _frames(11, 2, "cuda:0")]
    torch.cuda.synchronize(0)
    grp.saddles_enqueue(frames, records_per_frame=16)  # slabs of 32 records: nothing fits
    with pytest.raises(A.AgxError) as e:
        grp.saddles_fetch()
    assert e.value.status == -3
    grp.saddles_enqueue(frames)
    res, status = grp.saddles_fetch()
    assert (status == 0).all() and all(len(r) > 50 for r in res)
    # a list that is merely longer than the caller's room reports its length, as the single-detector fetch
    # does, so that the caller can size a retry (a frame whose device-side lists overflowed reports 0)
    short, st_short = grp.saddles_fetch(cap_per_frame=8, raise_on_overflow=False)
    assert (st_short == -3).all() and all(len(r) == 0 for r in short)
    assert [int(c) for c in grp.last_counts] == [len(r) for r in res]
    again, st_again = grp.saddles_fetch(cap_per_frame=int(grp.last_counts.max()))
    assert (st_again == 0).all() and all(a.tobytes() == b.tobytes() for a, b in zip(again, res))
    grp.close()


def test_group_rccl_transport(oracle):
    """The RCCL gather (ncclSend / ncclRecv over xGMI) through the REAL librccl on every visible device.  With one GPU the
    group has a single rank; the test switch AGX_GROUP_RCCL_SELF=1 makes its slabs take the library's path -- a send to
    itself and the matching receive in one ncclGroup on the rank's non-blocking stream --: the dlopen / dlsym binding of
    group.cpp, ncclCommInitAll, the datatype constant and the ordering behind the chain run against librccl.so itself, not
    the test suite's stand-in.  (Without the switch a group of one does not touch the library: next test.)"""
    import torch
    import aprilgrid_rs_amd as A
    assert "AGX_RCCL_LIBRARY" not in os.environ
    n_dev = torch.cuda.device_count()
    os.environ["AGX_GROUP_RCCL_SELF"] = "1"
    try:
        grp = A.DetectorGroup("t36h11", list(range(n_dev)), transport="rccl")
    finally:
        del os.environ["AGX_GROUP_RCCL_SELF"]
    maps = open("/proc/self/maps").read()
    assert "librccl" in maps, "the RCCL transport did not load librccl"
    assert "librccl_stub" not in maps
    frames = [_frames(70 + 3 * r, 3, "cuda:%d" % r) for r in range(n_dev)]
    for d in range(n_dev):
        torch.cuda.synchronize(d)
    for rep in range(3):  # slab reuse; one ncclGroup per batch
        grp.saddles_enqueue(frames)
        res, status = grp.saddles_fetch()
    assert (status == 0).all()
    host = np.concatenate([f.cpu().numpy() for f in frames])
    refs = oracle_saddles_parallel(oracle, host, threads=4)
    for i in range(len(res)):
        check_saddles(res[i], refs[i], "global frame %d" % i)
    grp.close()
    with pytest.raises(A.AgxError):
        A.DetectorGroup("t36h11", [0, 0], transport="rccl")  # duplicate devices: refused, not hung


def test_group_of_one_rank_does_not_need_librccl(oracle):
    """a single rank must not need librccl at all, whatever the transport asked for: with an unloadable
    library named in AGX_RCCL_LIBRARY a one-rank "rccl" group is created and gathers (two device-to-device copies);
    a two-rank group with the same environment fails with the loader's message."""
    import aprilgrid_rs_amd as A
    os.environ["AGX_RCCL_LIBRARY"] = "/nonexistent/librccl_not_here.so"
    try:
        grp = A.DetectorGroup("t36h11", [0], transport="rccl")
        frames = [_frames(75, 3, "cuda:0")]
        import torch
        torch.cuda.synchronize(0)
        grp.saddles_enqueue(frames)
        res, status = grp.saddles_fetch()
        assert (status == 0).all()
        refs = oracle_saddles_parallel(oracle, frames[0].cpu().numpy(), threads=3)
        for i in range(len(res)):
            check_saddles(res[i], refs[i], "frame %d" % i)
        grp.close()
        with pytest.raises(A.AgxError, match="dlopen librccl"):
            A.DetectorGroup("t36h11", [0, 0], transport="rccl")
    finally:
        del os.environ["AGX_RCCL_LIBRARY"]


def test_group_rccl_branch_with_stand_in_library(oracle, tmp_path):
    """group.cpp's RCCL transport -- the dlsym'd prototypes, one ncclGroupStart / ncclGroupEnd bracket per batch with a
    send pair per non-root rank and the matching receives on the root's communicator, everything enqueued on the
    ranks' own streams behind their chains -- with THREE ranks on one device: AGX_RCCL_LIBRARY points the loader at
    tests/stub_rccl (peer copies + events behind librccl's entry points, with the argument checks of the real
    library).  The gathered lists must equal the oracle's; the stand-in's counters must show exactly the calls the
    transport is supposed to make.  (No multi-GPU hardware: the real library's first run is the driver's.)"""
    import ctypes as C
    import subprocess
    import torch
    import aprilgrid_rs_amd as A
    from tests.util import ROOT
    so = str(tmp_path / "librccl_stub.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    os.environ["AGX_RCCL_LIBRARY"] = so
    try:
        ranks, fpr = 3, 3
        grp = A.DetectorGroup("t36h11", [0] * ranks, transport="rccl")
        frames = [_frames(90 + fpr * r, fpr, "cuda:0") for r in range(ranks)]
        torch.cuda.synchronize(0)
        for rep in range(3):  # slab reuse, one group per batch
            grp.saddles_enqueue(frames)
            res, status = grp.saddles_fetch()
        assert (status == 0).all() and len(res) == ranks * fpr
        host = np.concatenate([f.cpu().numpy() for f in frames])
        refs = oracle_saddles_parallel(oracle, host, threads=4)
        for i in range(len(res)):
            check_saddles(res[i], refs[i], "global frame %d" % i)
        stub = C.CDLL(so)
        st = (C.c_int * 8)()
        stub.stub_rccl_stats(st)
        groups, sends, recvs, kib, errors, created, destroyed, max_ops = list(st)
        assert errors == 0 and groups == 3 and created == ranks and destroyed == 0
        assert sends == recvs == 3 * 2 * (ranks - 1) and max_ops == 4 * (ranks - 1) and kib > 0
        grp.close()
        stub.stub_rccl_stats(st)
        assert st[6] == ranks and st[4] == 0
    finally:
        del os.environ["AGX_RCCL_LIBRARY"]


def test_group_of_eight_ranks_through_the_rccl_branch_one_rank_overflowing(oracle, tmp_path):
    """The shape of the 8-GPU node, rehearsed on whatever is visible: EIGHT ranks through group.cpp's RCCL transport
    (tests/stub_rccl behind AGX_RCCL_LIBRARY, the ranks round-robin over the visible devices), one ncclGroup with 14
    sends and 14 receives per batch, rank-major frame order -- and rank 5's detector given room for 40 saddles per frame,
    so that exactly ITS frames report AGX_ERR_CAPACITY while the 14 frames of the other seven ranks come back complete and
    equal to the oracle's."""
    import ctypes as C
    import subprocess
    import torch
    import aprilgrid_rs_amd as A
    from tests.util import ROOT
    so = str(tmp_path / "librccl_stub.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    os.environ["AGX_RCCL_LIBRARY"] = so
    try:
        ranks, fpr, bad = 8, 2, 5
        n_dev = torch.cuda.device_count()
        devices = [q % n_dev for q in range(ranks)]
        grp = A.DetectorGroup("t36h11", devices, transport="rccl")
        assert len(grp) == ranks
        lib = grp._lib
        lib.agx_group_detector.restype = C.c_void_p
        lib.agx_group_detector.argtypes = [C.c_void_p, C.c_int]
        h = lib.agx_group_detector(grp._g, bad)
        assert h and lib.agx_detector_set_limits(C.c_void_p(h), 0, 0, 40) == 0
        frames = [_frames(300 + fpr * q, fpr, "cuda:%d" % devices[q]) for q in range(ranks)]
        for d in set(devices):
            torch.cuda.synchronize(d)
        for rep in range(2):
            grp.saddles_enqueue(frames)
            res, status = grp.saddles_fetch(raise_on_overflow=False)
        assert len(res) == ranks * fpr
        want_bad = [bad * fpr + f for f in range(fpr)]
        assert [i for i in range(len(res)) if status[i] != 0] == want_bad and all(status[i] == -3 for i in want_bad), status
        host = np.concatenate([f.cpu().numpy() for f in frames])
        refs = oracle_saddles_parallel(oracle, host, threads=4)
        for i in range(len(res)):
            if i in want_bad:
                assert len(res[i]) == 0 and len(refs[i]) > 40
            else:
                check_saddles(res[i], refs[i], "global frame %d (rank %d)" % (i, i // fpr))
        stub = C.CDLL(so)
        st = (C.c_int * 8)()
        stub.stub_rccl_stats(st)
        groups, sends, recvs, kib, errors, created, destroyed, max_ops = list(st)
        assert errors == 0 and groups == 2 and created == ranks and sends == recvs == 2 * 2 * (ranks - 1) and max_ops == 4 * (ranks - 1)
        grp.close()
    finally:
        del os.environ["AGX_RCCL_LIBRARY"]


def test_plain_c_group_client(oracle, tmp_path):
    """examples/c_group_client.c -- no Python / torch in that process: 3 ranks round-robin over the
    visible devices with the peer-copy gather, and one rank per device with the RCCL gather; the
    per-frame saddle counts it prints equal the oracle's."""
    import subprocess
    import torch
    from tests.test_abi_cpu import _build_c_group_client
    synth = synth_module()
    exe = _build_c_group_client(tmp_path)
    n_dev = torch.cuda.device_count()
    for ranks, transport in ((3, "peer"), (n_dev, "rccl")):
        fpr = 2
        frames = np.stack([np.asarray(synth.render_frame(600 + i, 480, 320)[0]) for i in range(ranks * fpr)])
        raw = tmp_path / ("frames_%s.raw" % transport)
        frames.tofile(raw)
        r = subprocess.run([exe, str(raw), "480", "320", str(fpr), str(ranks), transport], capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout, r.stderr)
        refs = oracle_saddles_parallel(oracle, frames, threads=4)
        for i, ref in enumerate(refs):
            assert "frame %d: %d\n" % (i, len(ref)) in r.stdout, (i, len(ref), r.stdout)
        assert "%d saddles in %d frames" % (sum(len(x) for x in refs), ranks * fpr) in r.stdout


def test_batches_in_flight_are_bitwise_reproducible():
    """tools/stress_concurrency.py, short: three detectors on three streams take batches of different
    size / format in turn; every result (records, counts, status flags, cluster counts) must equal the
    result of the same batch computed alone.  Found in round 2: k_refine's workgroups took the cluster
    count as their loop bound while others were appending second-tier clusters to it."""
    import subprocess
    import sys
    from tests.util import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_concurrency.py"), "700", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_chain_can_be_captured_into_a_hip_graph():
    """tools/graph_capture_probe.py: two consecutive batches captured with torch.cuda.CUDAGraph on a side
    stream and replayed give the eager results, and so does a single captured batch replayed three times
    (in its own process: a failed capture would leave the stream in an error state)."""
    import subprocess
    import sys
    from tests.util import ROOT
    env = dict(os.environ, FRAMES="12")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "graph_capture_probe.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.count("graph replay equals eager: True") == 2, r.stdout[-2000:] + r.stderr[-2000:]
    # ONE captured batch replayed three times (every captured batch clears its own counter set), and an
    # eager batch behind the graphs
    assert r.stdout.count("equals eager: True") == 6, r.stdout[-2000:]


def test_bench_sends_its_slabs_through_nccl_with_a_world_of_one(tmp_path):
    """bench.py --collective-world-1: the nccl (= RCCL) process group is initialised with one rank and every step's result
    slabs go through its gather (sharding.GatherPipeline), the gathered slab is compared with the rank's own."""
    import json
    import subprocess
    import sys
    from tests.util import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--collective-world-1", "--steps", "4", "--warmup", "1",
                        "--frames", "64", "--settle-ms", "0", "--no-extra", "--no-cpu-baseline", "--extra-pipeline", "0"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    gc = out["gather_check"]
    assert gc["through_collective"] is True and gc["backend"] == "nccl" and gc["ranks"] == 1 and gc["saddles"] > 0
    assert out["verified_frames"] == 64 and out["backend"] == "hip-gfx950"
