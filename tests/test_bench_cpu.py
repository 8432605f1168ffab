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


def test_plain_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the form the driver uses for N = 1, and the likeliest form of its
    N = 8 command): bench.py starts the ranks itself as a child under torch.distributed.run -- before it has imported torch or
    touched a device -- hands rank 0's one JSON line through and exits with the child's code."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"AGX_BENCH_STUB": "1", "OMP_NUM_THREADS": "1"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "2",
           "--width", "192", "--height", "128", "--settle-ms", "30", "--no-extra", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd="/tmp")  # (from any directory)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]  # stdout carries the one line and nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gather_check"]["ranks"] == 2 and out["gather_check"]["through_collective"] is True
    assert out["config"]["steps_per_gather"] == 1 and out["grouped_gather"]["steps_per_collective"] == 4
    assert "torch.distributed.run" in r.stderr  # the launch is announced on stderr


def test_a_failing_rank_fails_the_plain_form():
    """The child's exit code is the parent's: a launcher-less run whose ranks die reports non-zero and prints no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"AGX_BENCH_STUB": "1", "OMP_NUM_THREADS": "1"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "0", "--warmup", "0", "--frames", "2",
           "--width", "192", "--height", "128", "--settle-ms", "0", "--no-extra"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)  # (zero timed steps: rank 0 divides by it)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_a_launcher_that_disagrees_with_gpus_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env={**os.environ, "WORLD_SIZE": "1", "AGX_BENCH_STUB": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_more_gpus_than_the_node_shows_is_refused_early():
    """No stub: the parent counts the GPUs a rank would see in a short-lived child (never through HIP in its own process) and
    refuses before starting anything -- here, without a GPU, any N > 1 (on a box with n GPUs: n + 1)."""
    sys.path.insert(0, ROOT)
    import bench
    n = bench.visible_gpus()
    assert n is not None and n >= 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "AGX_BENCH_STUB")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(2, n + 1))], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 2 and "visible" in r.stderr and not r.stdout.strip()


def test_bench_two_rank_control_flow_under_gloo():
    """bench.main() with WORLD_SIZE = 2, launched exactly as the driver launches it (torch.distributed.run), on the
    CPU: gloo backend and tests/bench_stub.py behind the detector.  Every rank walks the real control flow -- settle
    rounds broadcast from rank 0, the double-buffered per-step gathers, the barriers, gather_check with the oracle
    check of a remote frame, the all_reduce of the step time -- so a collective-sequence mismatch between the ranks
    hangs or fails here, not on the first 8-GPU run.  Rank 0 prints the one JSON line."""
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {**os.environ, "AGX_BENCH_STUB": "1", "PYTHONPATH": ROOT, "OMP_NUM_THREADS": "1"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frames", "2", "--width", "192", "--height", "128", "--settle-ms", "30", "--no-extra", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["backend"] == "cpu-stub" and out["unit"] == "Mpix/s (cpu stub)" and out["metric"].startswith("STUB")  # a stub line cannot pass for a measurement
    gc = out["gather_check"]
    assert gc["ranks"] == 2 and gc["frames"] == 4 and gc["saddles"] > 0 and gc["oracle_checked_remote_frames"] == 1
    # whole-job value: both ranks' pixels over the slowest rank's time
    assert abs(out["value"] - 2 * 2 * 192 * 128 * 3 / (out["ms_per_step"] * 3 * 1e-3) / 1e6) < 0.02 * out["value"]
    assert "configs[2]" in out["config"]["workload"] and "chain_frac" in out["roofline"] and "a_min_frac" in out["roofline"]


def _launch(world, extra, timeout=900):
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {**os.environ, "AGX_BENCH_STUB": "1", "PYTHONPATH": ROOT, "OMP_NUM_THREADS": "1", "MKL_NUM_THREADS": "1"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_eight_rank_control_flow_under_gloo():
    """The run nobody has seen (SCALE has been skipped every round): bench.main() exactly as the driver launches it for
    N = 8 -- torch.distributed.run, 8 processes, `--gpus 8` -- with configs[2]'s sharding: 256 frames per rank, 2 048 in
    all, rank r owns frames [256 r, 256 r + 256) of the generator.  gloo + tests/bench_stub.py stand in for the devices
    (small frames: the stub's chain is the oracle).  Rank 0 must find all eight ranks' tables in the gather, its own slab
    unchanged, and the FIRST FRAME OF EVERY OTHER RANK -- re-rendered from its global index -- equal to the oracle's list:
    rank-major order, nothing dropped, no collective out of step on any of the 8 ranks (a mismatch hangs or fails here)."""
    out = _launch(8, ["--steps", "2", "--warmup", "1", "--frames", "256", "--width", "192", "--height", "128", "--settle-ms", "0",
                      "--no-extra", "--no-cpu-baseline"])
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["scaling"] == "weak" and out["backend"] == "cpu-stub"
    gc = out["gather_check"]
    assert gc["ranks"] == 8 and gc["frames"] == 2048 and gc["oracle_checked_remote_frames"] == 7 and gc["through_collective"] is True
    assert out["config"]["frames_per_gpu"] == 256 and "x8" in out["config"]["parallelism"] and "1 steps per collective" in out["config"]["parallelism"]
    assert out["grouped_gather"]["steps_per_collective"] == 4 and out["grouped_gather"]["value"] > 0
    # whole-job value: all eight ranks' pixels over the slowest rank's time
    assert abs(out["value"] - 8 * 256 * 192 * 128 * 2 / (out["ms_per_step"] * 2 * 1e-3) / 1e6) < 0.02 * out["value"]


def test_bench_gathers_every_nth_step_and_always_the_last():
    """--gather-every 3 over 4 timed steps on 3 ranks (a world that is neither 1, 2 nor a power of two): the last step's
    results still arrive complete on rank 0 and are checked like any other run's."""
    out = _launch(3, ["--steps", "4", "--warmup", "1", "--frames", "3", "--width", "192", "--height", "128", "--settle-ms", "0",
                      "--no-extra", "--no-cpu-baseline", "--gather-every", "3"])
    gc = out["gather_check"]
    assert out["n_gpus"] == 3 and gc["ranks"] == 3 and gc["frames"] == 9 and gc["oracle_checked_remote_frames"] == 2


def test_bench_one_collective_per_step_or_per_three():
    """The default is a collective per step, with a second reported pass at four steps per collective (the 2- and 8-rank tests above
    run both: the group that is not full at a fence is sent by the fence).  Here: --gather-steps 1 and 3, over 7 timed steps on 3 ranks;
    rank 0 still finds every rank's LAST step complete and checks it like any other run."""
    for k in ("1", "3"):
        out = _launch(3, ["--steps", "7", "--warmup", "1", "--frames", "3", "--width", "192", "--height", "128", "--settle-ms", "0",
                          "--no-extra", "--no-cpu-baseline", "--gather-steps", k])
        gc = out["gather_check"]
        assert out["n_gpus"] == 3 and gc["ranks"] == 3 and gc["frames"] == 9 and gc["oracle_checked_remote_frames"] == 2
        assert ("%s steps per collective" % k) in out["config"]["parallelism"]


def test_soft_legs_record_errors_but_never_hide_a_parity_failure():
    """bench.soft_leg: a reported-beside leg that fails for a reason other than parity (out of memory, a HIP error, a refused
    option) becomes an "error" field; an AssertionError -- results that differ from the oracle -- stays fatal."""
    sys.path.insert(0, ROOT)
    import bench
    import pytest
    assert bench.soft_leg(lambda a, b=1: a + b, 2, b=3) == 5

    def oom():
        raise MemoryError("no room for the stream")

    def hip():
        raise RuntimeError("HIP error: out of memory")

    def parity():
        assert False, "saddle lists differ from the oracle"

    assert bench.soft_leg(oom)["error"].startswith("MemoryError")
    assert "HIP error" in bench.soft_leg(hip)["error"]
    with pytest.raises(AssertionError):
        bench.soft_leg(parity)


def test_the_stream_of_the_end_to_end_leg_is_sized_from_the_memory_the_process_may_take():
    """bench.host_memory_available(): MemAvailable narrowed by the cgroup's limit (a cgroup limit kills, it does not raise)."""
    sys.path.insert(0, ROOT)
    import bench
    avail = bench.host_memory_available()
    total = next(int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemTotal:"))
    assert avail is not None and 0 < avail <= total
