#!/usr/bin/env python3
"""bench.py -- throughput of the AprilGrid saddle chain on MI355X.

A "step" is one pass of the hot path (luma -> blur -> Hessian response -> min/threshold ->
clustering -> rochade_refine -> k/phi filter, kernels K1..K4) over one batch of synthetic
1280x800 u8 frames that is already resident in HBM.  N = 1 is BASELINE.json configs[1]
(256 frames on one MI355X); N > 1 is configs[2]: every rank owns 256 frames (weak scaling,
2048 frames on 8 GPUs) and the only collective is the RCCL gather of the result slabs to rank 0
(one collective per step; a second pass with --grouped-gather steps per collective is reported beside it).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  value = pixels of all ranks through the chain per second
(Mpix/s), inputs resident in HBM, over the K timed steps (wall clock between two fences);
ms_per_step_median = median of the K per-step device times (events between the steps).
roofline = K1 (the dominant kernel): algorithmic bytes per launch / average launch duration from
hipEvents recorded on the launch stream inside the timed region (every 5th step: the events
themselves cost the stream ~10 us per timed launch).  The timed region is strictly
serial (one batch at a time); at N = 1 a second pass with --extra-pipeline batches in flight
(sharding.ChainPipeline) is reported as "pipelined".

N = 1 additionally reports, in the same line:
  cpu_baseline   the C oracle (port of the reference CPU path, 1 thread; all_cores = the same
                 frame-parallel on the host cores) on a bounded sample of the same frames;
  verified_frames  the bench's own frames whose GPU saddle lists were compared with the oracle
                 (outside the timed region; the run aborts on a mismatch);
  extra_configs  BASELINE.json configs[3] (3840x2160 x32), configs[4] (RGB8 x256, the kornia
                 front-end layout), L16 x256 and the pure-noise sensitivity row, each with its own
                 K1 roofline and oracle check;
  host_boundary  the PCIe-inclusive rate: the same batch handed over in pinned HOST memory, saddle
                 lists back in host memory, batch after batch -- never `value`;
  detect_end_to_end  TagDetector::detect over the batch (agx_detect_batch: pageable host frames in, tag ids +
                 corners out; upload, chain, board search and decode inside) -- never `value`.  Two tails, the same tags:
                 `device_tail` (the default: board search + decode as a kernel behind the chain, set by PCIe and the
                 kernel's slowest frame) and `host_tail` (a pool of host threads: frames/s by thread count up to the
                 CPUs the process is granted, cgroup quota stated);
  extra_configs["configs[0]_single_frame"], reference_bench_detection
                 BASELINE.json configs[0] (one frame through detect: the reference's
                 data/1520525725372653511.png and one synthetic 1280x800 frame) and the reference's own
                 bench shape (benches/bench_detection.rs:24-36) on its 7 images: latency of the GPU
                 chain, of the GPU path's detect and of the oracle's detect, side by side.
roofline additionally carries chain_frac (the bytes this design moves per step / ms_per_step / 8 TB/s)
and a_min_frac (SURVEY.md 8(d)'s fused lower bound over the same time).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec)
HBM_MEASURED_GBPS = 6290.0    # float4-copy ceiling measured on MI355X (same guide)
IN_BYTES = {"L8": 1, "L16": 2, "RGB8": 3}
ANGLE_TOL_DEG = 1e-3          # theta / phi: device acosf / atan2f vs glibc (tests/util.py)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--format", default="L8", choices=["L8", "L16", "RGB8"])
    ap.add_argument("--unique", type=int, default=0,
                    help="distinct rendered frames per GPU, tiled to --frames (0 = every frame distinct: frame "
                         "indices rank*F .. rank*F+F-1 of the seeded generator; rendering is not timed)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the bench's own frames")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra configurations (4K, RGB8, L16, noise)")
    ap.add_argument("--noise", action="store_true", help="pure-noise frames (sensitivity row) as the main workload")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="batches in flight per GPU in the timed region (detectors on separate HIP streams; "
                         "1 = strictly serial: value, ms_per_step and roofline then describe the same launches)")
    ap.add_argument("--extra-pipeline", type=int, default=3,
                    help="N = 1 only: a second, separately reported timed pass with this many batches in flight (0 = skip)")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed: keep the chain running about this long before the W warm-up steps so that workspace "
                         "allocation is done and the GPU clocks have ramped (0 = off)")
    ap.add_argument("--images", action="store_true",
                    help="instead of the batch benchmark: the reference's own bench shape (benches/bench_detection.rs, "
                         "benches/bench_blur.rs) on its 7 / 3 fixture images, GPU path next to the CPU oracle "
                         "(tools/bench_images.py)")
    ap.add_argument("--collective-world-1", action="store_true",
                    help="N = 1 only: initialise the nccl (= RCCL) process group with a world of one rank anyway and send every "
                         "step's result slabs through its gather -- what a one-GPU box can exercise of the multi-GPU path")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1: gather the result slabs to rank 0 every n-th step (and the last one) instead of every step; "
                         "the default -- and what the headline is quoted on -- is every step")
    ap.add_argument("--gather-steps", type=int, default=1,
                    help="N > 1: the result slabs of this many consecutive steps travel to rank 0 in ONE collective (every step's results "
                         "are delivered, the fence sends a group that is not full).  Default 1 = a collective per step: BASELINE.json "
                         "configs[2] as written, and what `value` is quoted on")
    ap.add_argument("--grouped-gather", type=int, default=4,
                    help="N > 1: after the timed region, a second, separately reported timed pass with this many steps per collective "
                         "(fewer, larger messages: one rank through nccl pays +18 us per step with 1, +11 with 4, +9 with 8 -- "
                         "profiles/r5_gather_steps.txt); reported as `grouped_gather`, never `value`; 0 = skip")
    ap.add_argument("--stream-frames", type=int, default=8192,
                    help="N = 1: frames of the long stream of the detect_end_to_end leg (8192 = 8.4 GB of host memory and as much on the "
                         "device; halved until it fits a quarter of the host memory this process may still take)")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per K1 launch from a separate rocprofv3 --pmc run (profiles/)")
    return ap.parse_args(argv)


def make_workload(first_frame, n_frames, width, height, fmt, unique, noise, device):
    """The bench's frames: `unique` distinct frames [first_frame, first_frame + unique) of the seeded
    renderer (per-frame seed = splitmix64(0xA9121D ^ frame_index)), rendered
    straight into HBM and tiled to n_frames.  Returns (frames tensor, number of distinct frames)."""
    from aprilgrid_rs_amd import synth
    uniq = max(1, min(unique if unique > 0 else n_frames, n_frames))
    base, _ = synth.render_batch(first_frame, uniq, width, height, device=device, fmt=fmt, pure_noise=noise)
    reps = (n_frames + uniq - 1) // uniq
    frames = base.repeat((reps,) + (1,) * (base.dim() - 1))[:n_frames].contiguous()
    return frames, uniq


def host_view(frames, fmt):
    import numpy as np
    h = frames.cpu().numpy()
    return h.view(np.uint16) if fmt == "L16" else h


def verify_against_oracle(results, frames_host, what, threads=8):
    """GPU saddle lists (arrays [n, 5]: x, y, k, theta, phi) against the oracle on the same frames:
    count and order identical, x / y / k bit-exact, theta / phi within ANGLE_TOL_DEG.  Outside the
    timed region.  Returns the number of frames compared; raises on the first mismatch."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    with ThreadPoolExecutor(max(1, threads)) as ex:
        refs = list(ex.map(O.refined_saddle_points, frames_host))
    for i, (got, ref) in enumerate(zip(results, refs)):
        if got is None or len(got) != len(ref):
            raise SystemExit("VERIFY FAILED (%s frame %d): %s saddles vs oracle %d"
                             % (what, i, "no" if got is None else len(got), len(ref)))
        for j, f in enumerate(("x", "y", "k")):
            if not np.array_equal(np.ascontiguousarray(got[:, j]).view(np.uint32), ref[f].view(np.uint32)):
                raise SystemExit("VERIFY FAILED (%s frame %d): field %s is not bit-exact" % (what, i, f))
        for j, f in ((3, "theta"), (4, "phi")):
            if len(ref) and float(np.max(np.abs(got[:, j] - ref[f]))) > ANGLE_TOL_DEG:
                raise SystemExit("VERIFY FAILED (%s frame %d): field %s beyond %g deg" % (what, i, f, ANGLE_TOL_DEG))
    return len(refs)


def host_parallelism():
    """CPUs this process may keep busy = min(affinity mask, cgroup CPU quota): the library's agx_host_parallelism() when it
    is loadable, else the affinity mask."""
    try:
        from aprilgrid_rs_amd import _ffi
        return max(1, int(_ffi.lib().agx_host_parallelism()))
    except Exception:
        return max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))


def native_oracle_library():
    """oracle/agx_oracle.c built -O3 -march=native ON THIS MACHINE (the prebuilt oracle/liborc_native.so travels with the
    snapshot and was compiled for the build container's CPU: -march=native of another machine is neither safe nor the best
    this host can do).  Falls back to the prebuilt file where there is no compiler."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle", "agx_oracle.c")
    prebuilt = os.path.join(ROOT, "oracle", "liborc_native.so")
    out = os.path.join(tempfile.gettempdir(), "liborc_native_%d.so" % os.getuid())
    try:
        if not (os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src)):
            tmp = out + ".%d.tmp" % os.getpid()
            subprocess.run(["gcc", "-O3", "-march=native", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                            "-o", tmp, src, "-lm"], check=True, capture_output=True)
            os.replace(tmp, out)
        return out, "built on this host"
    except Exception:
        return prebuilt, "prebuilt (no compiler on this host)"


def cpu_baseline(frames_host, fmt, budget_s):
    """Oracle (C port of the reference CPU path, -O3 -march=native, still no FMA contraction),
    one thread, chain only (refined_saddle_points), on as many of the bench's own frames as fit
    in the budget."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    O.build()
    native, native_how = native_oracle_library()
    lib = C.CDLL(native)
    lib.orc_refined_saddle_points.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_void_p]
    prm = O.default_params()
    out = np.zeros(1 << 16, O.SADDLE_DTYPE)
    n_done, t_used = 0, 0.0
    h, w = frames_host.shape[1:3]
    ofmt = {"L8": O.FMT_L8, "L16": O.FMT_L16, "RGB8": O.FMT_RGB8}[fmt]
    stride = w * IN_BYTES[fmt]
    # one untimed warm-up call
    lib.orc_refined_saddle_points(frames_host[0].ctypes.data, w, h, stride, ofmt, C.addressof(prm),
                                  out.ctypes.data, len(out), None)
    while t_used < budget_s:
        f = frames_host[n_done % len(frames_host)]
        t0 = time.perf_counter()
        lib.orc_refined_saddle_points(f.ctypes.data, w, h, stride, ofmt, C.addressof(prm), out.ctypes.data,
                                      len(out), None)
        t_used += time.perf_counter() - t0
        n_done += 1
    mpix = n_done * w * h / t_used / 1e6
    # SURVEY.md 8(d) baseline #2: the same code, frame-parallel over the host cores this process may
    # use (ctypes releases the GIL; every thread has its own output buffer) -- a short extra sample
    import threading
    # every CPU the process may keep busy: its affinity mask narrowed by its cgroup CPU quota (a 1-GPU box of this pool shows
    # 256 CPUs and grants 16: cpu.max "1600000 100000"; threads beyond the quota only get the whole process throttled)
    n_thr = host_parallelism()
    counts = [0] * n_thr
    t_end = time.perf_counter() + min(5.0, budget_s)

    def worker(t):
        o = np.zeros(1 << 16, O.SADDLE_DTYPE)
        i = t
        while time.perf_counter() < t_end:
            f = frames_host[i % len(frames_host)]
            lib.orc_refined_saddle_points(f.ctypes.data, w, h, stride, ofmt, C.addressof(prm), o.ctypes.data, len(o), None)
            counts[t] += 1
            i += n_thr

    t_mt = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    t_mt = time.perf_counter() - t_mt
    return {"value": round(mpix, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
            "all_cores": {"value": round(sum(counts) * w * h / t_mt / 1e6, 1), "unit": "Mpix/s", "cores": n_thr,
                          "host_cores_shown": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1),
                          "sample": "%d frames in %.1f s on %d threads" % (sum(counts), t_mt, n_thr)},
            "sample": "%d frames %dx%d %s, chain only (refined_saddle_points), oracle/agx_oracle.c -O3 "
                      "-march=native (%s) -ffp-contract=off, %.1f s" % (n_done, w, h, fmt, native_how, t_used),
            "ms_per_frame": round(1e3 * t_used / n_done, 3)}


def k1_roofline(px, in_b, k1_ms, k1_n, traffic=None, traffic_src=None):
    """K1 reads the input once, writes the blur plane (f32) and 1 bit / px of candidate mask."""
    avg = k1_ms / max(k1_n, 1)
    nbytes = px * (in_b + 4 + 0.125)
    gbps = nbytes / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
    return {"kernel": "k_blur_hessian (K1)", "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
            "frac_of_measured_copy_ceiling": round(gbps / HBM_MEASURED_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_px": in_b + 4.125, "bytes_per_launch": nbytes,
            "avg_launch_ms": round(avg, 5), "launches_timed": k1_n}


def box_copy_gbps(torch, dev):
    """What a plain device-to-device copy of 1 GiB reaches on THIS box (read + write bytes per second,
    best of 5): boxes of the pool differ, and the guide's 6.29 TB/s is not what every one of them gives."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    best = None
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * n * 4 / (best * 1e-3) / 1e9


def recorded_traffic(key):
    """HBM bytes per K1 launch from rocprofv3 PMC passes (FETCH_SIZE x2 per the calibration +
    WRITE_SIZE), collected separately (tools/final_profile.sh) and kept under profiles/."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "k1_traffic.json")))
        if key in rec:
            return rec[key]["bytes_per_launch"], rec[key]["source"]
    except Exception:
        pass
    return None, None


def extra_leg(torch, A, dev, name, workload, n_frames, width, height, fmt, unique, noise, steps, warmup, verify_n, pitch=None,
              settle_ms=300.0):
    """One extra configuration on its own detector, strictly serial: wall-clock value, per-step
    median, K1 roofline from hipEvents in the timed region, oracle check of the first frames.
    pitch (L8 only): bytes per row of the device allocation the frames are cut out of (padding after every row).
    settle_ms: like the main workload, the leg keeps its chain running about that long, untimed, before its warm-up steps:
    the legs follow seconds of host-side work (rendering, the oracle check of the previous leg) during which the GPU's clocks
    fall back, and a timed region of 10 ms right after it read K1 15 - 19 % slower than the same kernel in steady state
    (profiles/r6_k1_plan_sweep_*.txt against r5's extra_configs)."""
    frames, uniq = make_workload(0, n_frames, width, height, fmt, unique, noise, dev)
    det = A.TagDetector(A.TagFamily.T36H11, None, device=dev.index)
    px = n_frames * width * height
    enqueue = lambda: det.saddles_batch_enqueue(frames)
    if pitch is not None:
        from aprilgrid_rs_amd import _ffi
        assert fmt == "L8" and pitch >= width
        padded = torch.full((n_frames, height, pitch), 0xA5, dtype=torch.uint8, device=dev)  # (the padding holds garbage)
        padded[:, :, :width] = frames
        torch.cuda.synchronize(dev)  # (the fill is done before anything reads the frames, whatever stream the chain is on)
        # (the raw-address enqueue launches on torch's current stream like the tensor forms: the per-step events below see the kernels)
        enqueue = lambda: det.saddles_batch_enqueue_ptr(padded.data_ptr(), n_frames, width, height, pitch, pitch * height, _ffi.AGX_L8)
    try:
        enqueue()  # (allocates the workspace)
        torch.cuda.synchronize(dev)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
            for _ in range(8):
                enqueue()
            torch.cuda.synchronize(dev)
        for _ in range(max(warmup, 1) + 2):
            enqueue()
        torch.cuda.synchronize(dev)
        det.set_option("profile_stride", 5 if steps >= 20 else 1)
        det.profile_enable(1)
        det.profile_reset()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            evs[i].record()
            enqueue()
        evs[steps].record()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        k1_ms, k1_n = det.profile_read()["k_blur_hessian"]
        det.profile_enable(0)
        per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
        res, status = det.saddles_batch_fetch(raise_on_overflow=False)
        import numpy as np
        bad = int((status != 0).sum())
        verified = 0
        if verify_n and not bad:
            host = host_view(frames[:min(verify_n, uniq)], fmt)
            got = [np.stack([r["x"], r["y"], r["k"], r["theta"], r["phi"]], axis=1) if len(r) else np.zeros((0, 5), np.float32)
                   for r in res[:len(host)]]
            verified = verify_against_oracle(got, host, name)
        key = "%dx%dx%d_%s%s" % (n_frames, width, height, fmt, "_noise" if noise else "")
        traffic, src = recorded_traffic(key)
        return {"workload": workload, "frames": n_frames, "distinct_frames": uniq, "width": width, "height": height,
                "format": fmt, "steps": steps, "value": round(px * steps / dt / 1e6, 1), "unit": "Mpix/s",
                "ms_per_step": round(1e3 * dt / steps, 4), "ms_per_step_median": round(statistics.median(per_step), 4),
                "frames_per_s": round(n_frames * steps / dt, 1),
                "saddles_per_frame": round(float(sum(len(r) for r in res)) / max(len(res), 1), 1),
                "frames_over_capacity": bad, "verified_frames": verified,
                "k1_rows_per_segment": det.get_option("k1_rows_per_segment"),
                "roofline": k1_roofline(px, IN_BYTES[fmt], k1_ms, k1_n, traffic, src)}
    finally:
        det.close()
        del frames
        torch.cuda.empty_cache()


def host_boundary_leg(torch, A, dev, n_frames, width, height, steps):
    """What the boundary delivers when it is handed HOST buffers (the reference's own call shape): the
    batch in pinned host memory -> HBM (PCIe) -> chain -> saddle lists back in host memory, batch after
    batch.  Never `value`.  The upload of batch i + 1 (a side stream, two staging buffers) runs under the chain and
    the fetch of batch i; the results come back through a kernel that writes the detector's mapped pinned memory
    (k_publish), not through device-to-host copies -- those queued behind the 262 MB upload on the DMA engine (8.3 ms per
    batch when that was tried in round 3) --, into arrays the caller owns (agx_saddles_batch_fetch as a C / Rust caller
    uses it).  `ms_per_batch_serial`: upload, chain, fetch strictly one after the other, lists as Python objects."""
    import numpy as np
    from aprilgrid_rs_amd.detector import SADDLE_DTYPE
    frames, _ = make_workload(0, n_frames, width, height, "L8", 0, False, dev)
    host = torch.empty(frames.shape, dtype=frames.dtype, pin_memory=True)
    host.copy_(frames)
    det = A.TagDetector(A.TagFamily.T36H11, None, device=dev.index)
    stages = [torch.empty_like(frames), torch.empty_like(frames)]
    copy_stream = torch.cuda.Stream(dev)
    uploaded = [torch.cuda.Event(), torch.cuda.Event()]
    consumed = [torch.cuda.Event(), torch.cuda.Event()]
    out = np.zeros((n_frames, 1024), SADDLE_DTYPE)
    counts = np.zeros(n_frames, np.uint32)
    status = np.zeros(n_frames, np.int32)
    main = torch.cuda.current_stream(dev)
    try:
        det.saddles_batch_enqueue(frames)  # workspace
        ref, ref_status = det.saddles_batch_fetch(cap_per_frame=1024, raise_on_overflow=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            stages[0].copy_(host, non_blocking=True)
        torch.cuda.synchronize(dev)
        t_h2d = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(max(2, steps // 3)):
            stages[0].copy_(host, non_blocking=True)
            det.saddles_batch_enqueue(stages[0])
            res, st = det.saddles_batch_fetch(cap_per_frame=1024, raise_on_overflow=False)
        t_serial = (time.perf_counter() - t0) / max(2, steps // 3)

        def upload(i):
            b = i & 1
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(consumed[b])  # the chain that read this staging buffer last is through
                stages[b].copy_(host, non_blocking=True)
                uploaded[b].record(copy_stream)

        for b in range(2):
            consumed[b].record(main)
        torch.cuda.synchronize(dev)
        upload(0)
        fill = 2  # batches before the clock starts: the steady state is what a stream of batches sees
        for i in range(steps + fill):
            b = i & 1
            if i == fill:
                t0 = time.perf_counter()  # (the fetch of batch fill - 1 has just returned: the host is in step with the device)
            if i + 1 < steps + fill:
                upload(i + 1)
            main.wait_event(uploaded[b])
            det.saddles_batch_enqueue(stages[b])
            consumed[b].record(main)
            det.saddles_batch_fetch_into(out, counts, status)
        t_pipe = (time.perf_counter() - t0) / steps
        assert (status == 0).all() and (counts > 0).all()
        # the lists that came through the pipelined loop equal the device-resident path's (fetched before the loop)
        for i in range(n_frames):
            assert ref_status[i] == 0 and out[i, : counts[i]].tobytes() == ref[i].tobytes(), "host-boundary results differ from the resident path (frame %d)" % i
        nbytes = frames.numel() * frames.element_size()
        return {"workload": "%d frames %dx%d L8 in pinned host memory -> saddle lists in host memory" % (n_frames, width, height),
                "h2d_GBps": round(nbytes / t_h2d / 1e9, 1), "ms_upload_alone": round(1e3 * t_h2d, 3),
                "ms_per_batch": round(1e3 * t_pipe, 3), "over_upload": round(t_pipe / t_h2d, 3),
                "frames_per_s": round(n_frames / t_pipe, 1), "Mpix_per_s": round(n_frames * width * height / t_pipe / 1e6, 1),
                "ms_per_batch_serial": round(1e3 * t_serial, 3),
                "results_equal_resident_path": True,
                "note": "PCIe-inclusive: the upload is nearly all of it (1 B/px over PCIe against 5.1 B/px of HBM traffic); never `value`. "
                        "ms_per_batch: the next batch's upload in flight (side stream, two staging buffers), results through k_publish into "
                        "caller-owned arrays; ms_per_batch_serial: upload -> chain -> fetch one after the other, Python lists"}
    finally:
        det.close()
        del frames, stages, host
        torch.cuda.empty_cache()


def host_memory_available():
    """Bytes of host memory this process may still take: MemAvailable, narrowed by the cgroup's limit (v2 memory.max -
    memory.current, v1 memory.limit_in_bytes - memory.usage_in_bytes).  Beyond a cgroup limit there is no MemoryError to
    catch -- the kernel kills the process -- so buffers of several GB are sized from this beforehand."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            l = open(lim).read().strip()
            if l != "max" and int(l) < (1 << 60):
                left = int(l) - int(open(cur).read().strip())
                avail = left if avail is None else min(avail, left)
        except (OSError, ValueError):
            pass
    return avail


def soft_leg(fn, *args, **kw):
    """A reported-beside leg (never `value`): a failure that is not a parity failure -- out of memory, a HIP error, a refused
    option -- becomes an "error" field of that leg instead of taking the whole bench line with it.  AssertionErrors (results that
    differ from the oracle or from another path) stay fatal: a wrong result must not read as a missing measurement."""
    try:
        return fn(*args, **kw)
    except AssertionError:
        raise
    except Exception as e:
        import traceback
        traceback.print_exc(file=sys.stderr)
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def detect_end_to_end_leg(torch, A, dev, n_frames, width, height, quick=False, stream_frames=8192):
    """TagDetector::detect over the batch, end to end (SURVEY.md 8(d) "what is NOT in t_chain ... reported separately as
    end-to-end frames/s with the host-thread count stated"; reference shape benches/bench_detection.rs:24-36): configs[1]'s
    frames in ordinary (pageable) HOST memory -> agx_detect_batch -> tag ids + corners in host arrays.  Upload, chain, board
    search, decode and the tags' way back are all inside; never `value`.  Timed: the C call alone (caller-owned output arrays,
    as a C / Rust caller has them).  Two tails, the same tags (checked here, and frame by frame in tests/test_gpu_device_tail.py):
      device  (the default where the process's atan2f is glibc's routine) board search + decode as a HIP kernel behind the
              chain, csrc/tail_kernels.hip: the call is then set by the upload over PCIe and the kernel's slowest frame;
      host    the reference's exhaustive search on a pool of host threads (~0.8 ms per frame and thread): set by the CPUs the
              box gives the process (agx_host_parallelism(): affinity mask or cgroup CPU quota, whichever is smaller) -- by
              thread count."""
    import numpy as np
    from aprilgrid_rs_amd import _ffi
    frames, _ = make_workload(0, n_frames, width, height, "L8", 0, False, dev)
    host = frames.cpu().numpy()
    del frames
    quota = int(_ffi.lib().agx_host_parallelism())
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cap = 64
    out = np.zeros((n_frames, cap), A.TagDetector.TAG_DTYPE)
    counts = np.zeros(n_frames, np.uint32)
    status = np.zeros(n_frames, np.int32)

    def run(det, frames_h, thr, n, reps):
        det.detect_batch_raw(frames_h[: min(n, 4 * thr)], n_threads=thr, cap=cap, out=out[: min(n, 4 * thr)], counts=counts[: min(n, 4 * thr)],
                             status=status[: min(n, 4 * thr)])  # pool start-up, staging, per-thread scratch: outside the clock
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            rc, _, _, _ = det.detect_batch_raw(frames_h[:n], n_threads=thr, cap=cap, out=out[:n], counts=counts[:n], status=status[:n])
            ts.append(time.perf_counter() - t0)
            assert rc == 0 and (status[:n] == 0).all(), (rc, status[:n])
        return statistics.median(ts)

    def formats_and_stream(det, res, big):
        """What both tails are timed on beyond the 256-frame call: a long stream, pinned input, RGB8 / L16 batches."""
        if big is not None:
            o2 = np.zeros((len(big), cap), det.TAG_DTYPE)
            c2 = np.zeros(len(big), np.uint32)
            s2 = np.zeros(len(big), np.int32)
            det.detect_batch_raw(big[:1024], n_threads=quota, cap=cap, out=o2[:1024], counts=c2[:1024], status=s2[:1024])  # (staging of the stream's chunk size)
            dt = None
            for _ in range(2):  # (the better of two: the box's other tenants share its host memory and PCIe root)
                t0 = time.perf_counter()
                rc, _, _, _ = det.detect_batch_raw(big, n_threads=quota, cap=cap, out=o2, counts=c2, status=s2)
                d1 = time.perf_counter() - t0
                dt = d1 if dt is None or d1 < dt else dt
                assert rc == 0 and np.array_equal(c2[:n_frames], c2[-n_frames:]) and (s2 == 0).all()
            res["frames_per_s_%d_frames" % len(big)] = round(len(big) / dt, 1)
            del o2, c2, s2
        # the same frames in PINNED host memory (hipHostMalloc / torch pin_memory): the runtime's pageable path already runs at
        # the PCIe rate (profiles/r5_ubench_h2d_pageable.txt), so little changes
        pinned_t = torch.empty((n_frames, height, width), dtype=torch.uint8, pin_memory=True)
        pinned_t.copy_(torch.from_numpy(host))
        res["frames_per_s_pinned_input"] = round(n_frames / run(det, pinned_t.numpy(), quota, n_frames, 3), 1)
        del pinned_t
        for fmt2 in ("RGB8", "L16"):  # the u8 luma the decode needs is computed on the device behind the chain
            fr2, _ = make_workload(0, n_frames, width, height, fmt2, 64, False, dev)
            h2 = host_view(fr2, fmt2)
            del fr2
            res["frames_per_s_" + fmt2] = round(n_frames / run(det, h2, quota, n_frames, 3), 1)
            del h2

    det = A.TagDetector(A.TagFamily.T36H11, None, device=dev.index)
    det.set_option("device_tail", 0)
    det_d = A.TagDetector(A.TagFamily.T36H11, None, device=dev.index)
    try:
        # per-frame detect of a sample: what the batch call must reproduce bit for bit
        sample = list(range(0, n_frames, max(1, n_frames // 8)))
        ref = {i: det.detect(host[i]) for i in sample}

        def check_sample(what):
            for i in sample:
                got = {int(t["id"]): t["xy"].reshape(4, 2) for t in out[i, : counts[i]]}
                assert sorted(got) == sorted(ref[i]) and all(np.array_equal(got[k], ref[i][k]) for k in got), "%s: detect_batch differs from detect (frame %d)" % (what, i)

        big = None
        if not quick:
            # a longer stream of frames (the same 256 thirty-two times over: 8 GB of host memory): start-up and drain amortised, and
            # long enough (several 100 ms scheduler periods) that a CPU quota binds -- the sustained rate.  Sized from the memory
            # this process may still take (a cgroup limit kills, it does not raise): the stream may use a quarter of it
            avail = host_memory_available()
            n_stream = max(n_frames, stream_frames // n_frames * n_frames)
            while avail is not None and n_stream > n_frames and 4 * n_stream * host[0].nbytes > avail:
                n_stream //= 2
            if n_stream >= 4 * n_frames:
                try:
                    big = np.concatenate([host] * (n_stream // n_frames))
                except MemoryError:
                    big = None
        # ---- the host tail, by thread count --------------------------------------------------------------------------------
        counts_t = [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t < quota] + [quota]
        if quick:
            counts_t = sorted(set([1, quota]))
        rows = {}
        for thr in counts_t:
            n = n_frames if thr >= 4 else max(16, min(n_frames, 32 * thr))
            dt = run(det, host, thr, n, 3 if thr >= 4 else 1)
            rows["threads_%d" % thr] = {"frames": n, "frames_per_s": round(n / dt, 1), "ms_per_frame_per_thread": round(1e3 * dt * thr / n, 3),
                                        "tags_per_frame": round(float(counts[:n].mean()), 1)}
            if thr == quota:  # the call just made covered every frame
                check_sample("host tail")
        base = rows["threads_1"]["frames_per_s"]
        for k, r in rows.items():
            r["parallel_efficiency"] = round(r["frames_per_s"] / (base * int(k.split("_")[1])), 3)
        best = rows["threads_%d" % quota]
        host_res = {"frames_per_s": best["frames_per_s"], "threads": quota, "ms_per_frame_per_thread": best["ms_per_frame_per_thread"],
                    "parallel_efficiency": best["parallel_efficiency"], "by_threads": rows}
        if not quick:
            formats_and_stream(det, host_res, big)
            if quota < affinity:  # what threads beyond the quota cost (the reason the default stops at it)
                thr = min(affinity, 4 * quota)
                host_res["beyond_quota"] = {"threads": thr, "frames_per_s": round(n_frames / run(det, host, thr, n_frames, 3), 1),
                                            "note": "a %d-frame call is a burst of a few milliseconds that can fit inside one quota period; a stream cannot" % n_frames}
            # DetectorParams::max_num_of_boards = 1 (src/detector.rs:25-41; default 2): one board search per frame instead of two -- on
            # frames that hold one board the second search (30 seeds that find nothing) is 95 % of the host tail.  Not the
            # reference's default, so not the headline
            p1 = A.DetectorParams.default_params()
            p1.max_num_of_boards = 1
            det1 = A.TagDetector(A.TagFamily.T36H11, p1, device=dev.index)
            det1.set_option("device_tail", 0)
            try:
                host_res["frames_per_s_max_num_of_boards_1"] = round(n_frames / run(det1, host, quota, n_frames, 3), 1)
                host_res["tags_per_frame_max_num_of_boards_1"] = round(float(counts.mean()), 1)
            finally:
                det1.close()
        host_res["note"] = ("set by the host tail (the reference's exhaustive board search, one frame per thread) and by the CPUs the box gives the process "
                            "(host_cpu_quota of the host_cores it shows: cgroup cpu.max); parallel_efficiency = frames/s over threads x the 1-thread rate")
        # ---- the device tail (the default) -----------------------------------------------------------------------------------
        dev_res = None
        dt = run(det_d, host, quota, n_frames, 5)
        if det_d.get_option("last_device_tail_frames") == n_frames:  # (the default took the device tail: libm is glibc's, the call is large enough)
            check_sample("device tail")
            dev_res = {"frames_per_s": round(n_frames / dt, 1), "ms_per_call": round(1e3 * dt, 2), "tags_per_frame": round(float(counts.mean()), 1),
                       "frames_handed_back_to_the_host_tail": det_d.get_option("last_device_tail_fallbacks"),
                       "threads": quota}
            if not quick:
                formats_and_stream(det_d, dev_res, big)
                # the stream again with the frames ALREADY on the device (d_frames: the chain and the decode read them there, the
                # host copy only serves frames handed back): no upload -- what chain + device tail deliver by themselves
                big_d = torch.from_numpy(big).to(dev)
                o2 = np.zeros((len(big), cap), det_d.TAG_DTYPE)
                c2 = np.zeros(len(big), np.uint32)
                s2 = np.zeros(len(big), np.int32)
                best_dt = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    rc, _, _, _ = det_d.detect_batch_raw(big, n_threads=quota, cap=cap, device_frames=big_d, out=o2, counts=c2, status=s2)
                    dt2 = time.perf_counter() - t0
                    best_dt = dt2 if best_dt is None or dt2 < best_dt else best_dt
                assert rc == 0 and (s2 == 0).all() and np.array_equal(c2[:n_frames], c2[-n_frames:])
                dev_res["frames_per_s_%d_frames_resident_on_the_device" % len(big)] = round(len(big) / best_dt, 1)
                del big_d, o2, c2, s2
            dev_res["note"] = ("board search + decode on the device behind the chain (csrc/tail_kernels.hip: a workgroup per frame, ~2.5 ms per frame, 5 ms for "
                               "the slowest of 256); the call = upload over PCIe (4.8 ms per 256 frames) + chain + the kernel's slowest frame; a stream of "
                               "1024-frame chunks runs at the upload's rate.  The host threads only move the frames and take the frames the kernel hands back")
        del big
        top = dev_res if dev_res is not None else host_res
        res = {"workload": "configs[1]'s %d frames %dx%d L8 in pageable host memory -> tag ids + corners in host arrays (agx_detect_batch, the C call alone)" % (n_frames, width, height),
               "frames_per_s": top["frames_per_s"], "tail": "device" if dev_res is not None else "host", "threads": quota, "host_cores": affinity,
               "host_cpu_quota": quota, "ms_per_frame_per_thread": host_res["ms_per_frame_per_thread"],
               "sample_equals_per_frame_detect": len(sample), "device_tail": dev_res, "host_tail": host_res,
               "note": "never `value` (the chain alone delivers config.frames_per_s).  frames_per_s: the default path; ms_per_frame_per_thread: the host tail's"}
        for k in top:
            if k.startswith("frames_per_s_"):
                res[k] = top[k]
        return res
    finally:
        det.close()
        det_d.close()
        torch.cuda.empty_cache()


class CudaRuntime:
    """What main() needs from the device side.  tests/bench_stub.py provides the same interface on the CPU (gloo
    backend, the oracle behind the detector's enqueue call) so that the N > 1 control flow of main() -- settle-round
    broadcast, gather, gather_check, all_reduce of the step time -- runs in the CPU test suite; selected by the
    environment variable AGX_BENCH_STUB=1 and never on a GPU box."""
    backend = "nccl"

    def __init__(self, torch):
        self.torch = torch
        import aprilgrid_rs_amd as A
        self.A = A
        self.detector_cls = None  # sharding.ChainPipeline's default: the package's TagDetector

    def device(self, local_rank):
        assert self.torch.cuda.is_available(), "bench.py needs an MI355X"
        self.torch.cuda.set_device(local_rank)
        return self.torch.device("cuda", local_rank)

    def init_process_group(self, dist, dev):
        dist.init_process_group(self.backend, device_id=dev)

    def synchronize(self, dev):
        self.torch.cuda.synchronize(dev)

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)


def visible_gpus():
    """GPUs a rank would see, counted WITHOUT initialising HIP in this process (the parent of the ranks must never touch the
    GPU): a short-lived child asks torch.  (The KFD topology in sysfs is no substitute: a container shows every GPU of the host
    there, also those its device cgroup hides.)  None when the child cannot say."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                           timeout=600, cwd=ROOT)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        return None


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
    bench.py <the same arguments>), hand rank 0's one JSON line through on stdout and return the child's exit code.
    This process has imported neither torch nor the package and never touches the GPU; nothing is exec'd."""
    import socket
    import subprocess
    stub = os.environ.get("AGX_BENCH_STUB") == "1"
    if not stub:
        seen = visible_gpus()
        if seen is not None and seen < args.gpus:
            print("bench.py: --gpus %d but %d GPU%s visible to a rank on this node (torch.cuda.device_count() in a child process)"
                  % (args.gpus, seen, "" if seen == 1 else "s"), file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks on this pool
    env.setdefault("OMP_NUM_THREADS", "1")  # (what the launcher would set itself, with a warning)
    env["PYTHONPATH"] = ROOT + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    # cwd = the repository: `python -m` puts the working directory on sys.path, and whatever lies there must not shadow a module
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1, cwd=ROOT)
    try:
        for line in child.stdout:  # rank 0's JSON line -> stdout; anything else a rank printed -> stderr
            if line.startswith("{") and '"metric"' in line:
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(line)
        return child.wait()
    except BaseException:
        child.terminate()  # (the exact process started here; the launcher takes its ranks down with it)
        try:
            child.wait(30)
        except subprocess.TimeoutExpired:
            child.kill()
        raise


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: become the parent of the ranks -- BEFORE torch, the package or HIP
        raise SystemExit(launch_ranks(args, argv))
    if args.images:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_images
        return bench_images.main([])
    # stdout carries rank 0's ONE JSON line and nothing else: whatever a library prints on file descriptor 1 from here on (RCCL's
    # five-line version banner at the first collective, for one) goes to stderr; the line itself is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    from aprilgrid_rs_amd import sharding
    stub = os.environ.get("AGX_BENCH_STUB") == "1"  # CPU test of the N > 1 control flow (tests/test_bench_cpu.py)
    if stub and torch.cuda.is_available():
        raise SystemExit("AGX_BENCH_STUB=1 on a box with a GPU: the stub backend (CPU oracle behind the enqueue call) exists for the "
                         "CPU test of the control flow only and must never produce a benchmark line here -- unset it")
    if stub:
        from tests import bench_stub
        rt = bench_stub.StubRuntime(torch)
    else:
        rt = CudaRuntime(torch)
    A = rt.A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # (a launcher around this process that disagrees with --gpus: say so, do not guess)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = rt.device(local_rank)
    coll1 = world == 1 and args.collective_world_1
    if world > 1 or coll1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if coll1:
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        rt.init_process_group(dist, dev)

    W, H, F = args.width, args.height, args.frames
    # ---- synthetic workload: frames [rank*F, rank*F + F) of the seeded generator, rendered
    # straight into HBM (rendering is outside the timed region)
    first_frame, _ = sharding.shard_range(rank, world, F)
    frames, uniq = make_workload(first_frame, F, W, H, args.format, args.unique, args.noise, dev)
    px_per_step_rank = F * W * H

    # `--pipeline` detectors (own workspace + HIP stream each) take the steps in turn: K1 of step
    # i+1 overlaps the short, latency-bound sparse kernels of step i.  Result buffers stay in
    # HBM; for N > 1 they are gathered to rank 0 asynchronously (RCCL over xGMI) -- the one
    # collective of the path -- so the gather of a step overlaps the chain of the next.
    slab = 8192 if args.noise else sharding.SLAB_RECORDS  # pure noise: ~7100 saddles per 1280x800 frame
    pipe = sharding.ChainPipeline(A.TagFamily.T36H11, F, dev, depth=args.pipeline, dst=0, slab_records=slab,
                                  detector_cls=rt.detector_cls, force_collective=coll1, gather_every=args.gather_every,
                                  steps_per_gather=args.gather_steps if args.gather_every <= 1 else 1)

    def step():
        pipe.submit(frames)

    def fence():
        res = pipe.finish()
        rt.synchronize(dev)
        if world > 1 or coll1:
            dist.barrier()
            rt.synchronize(dev)
        return res

    # setup, untimed: first call allocates the workspace; then run until the clocks have settled.
    # Every rank issues the SAME sequence of collectives: the number of settle rounds is fixed from the
    # first round's duration as rank 0 measured it (broadcast), never from a rank's own clock.
    step()
    fence()
    t_round = time.perf_counter()
    for _ in range(8):
        step()
    fence()
    t_round = time.perf_counter() - t_round  # one round of 8 steps, workspace already allocated
    rounds = int(min(128, max(0, args.settle_ms * 1e-3 / max(t_round, 1e-4))))
    if world > 1:
        rounds_t = torch.tensor([rounds], dtype=torch.int64, device=dev)
        dist.broadcast(rounds_t, src=0)
        rounds = int(rounds_t.item())
    for _ in range(rounds):
        for _ in range(8):
            step()
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    tb = pipe.last_table.cpu().numpy()
    over = int(((tb[:, 2] & 7) != 0).sum())
    assert over == 0, "capacity overflow in the bench workload (%d frames): %s" % (over, tb[(tb[:, 2] & 7) != 0][:4])
    generic_frames = int(((tb[:, 2] & 16) != 0).sum())
    saddles_per_frame = float(tb[:, 0].mean())
    clusters_per_frame = float(tb[:, 3].mean())

    # timed region: hipEvents (on the launch stream) around K1 only -- 2 records per step -- and one
    # event between the steps (strictly serial: the chain launches on the current stream)
    # (an event pair costs the stream two ~5 us gaps around the kernel: K1 of every 5th step is timed)
    for d in pipe.dets:
        d.set_option("profile_stride", 5 if args.steps >= 20 else 1)
        d.profile_enable(1)
        d.profile_reset()
    serial = pipe.depth == 1
    evs = [rt.event() for _ in range(args.steps + 1)] if serial else []
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if serial:
            evs[i].record()
        step()
    if serial:
        evs[args.steps].record()
    gathered = fence()
    dt = time.perf_counter() - t0
    per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)] if serial else []
    k1_ms, k1_n = 0.0, 0
    for d in pipe.dets:
        ms, n = d.profile_read()["k_blur_hessian"]
        k1_ms, k1_n = k1_ms + ms, k1_n + n
        d.profile_enable(0)
        d.set_option("profile_stride", 1)

    # results of the last timed step: the gathered tables must be complete on rank 0 (every rank's
    # frames present, no blocking status) -- the first multi-GPU hardware run checks itself
    own = sharding.unpack_frames(pipe.gather.bufs[pipe.gather.i][0], pipe.gather.bufs[pipe.gather.i][1])
    gather_check = None
    if rank == 0:
        gs, gt = gathered
        assert len(gs) == world and len(gt) == world, "gather returned %d ranks" % len(gt)
        tot = 0
        for r in range(world):
            t = gt[r].cpu().numpy()
            assert t.shape == (F, 4) and (t[:, 2] & 7 == 0).all(), "rank %d: blocking frame status in the gathered table" % r
            assert (t[:, 0] > 0).all(), "rank %d: frames without saddles in the gathered table" % r
            tot += int(t[:, 0].sum())
        mine = sharding.unpack_frames(gs[0], gt[0])
        assert all(a.tobytes() == b.tobytes() for a, b in zip(mine, own)), "rank 0's gathered slab differs from its own results"
        gather_check = {"ranks": world, "frames": world * F, "saddles": tot,
                        "through_collective": bool(pipe.gather.multi), "backend": rt.backend if pipe.gather.multi else None}
        if world > 1 and not args.no_verify:
            # first frame of every other rank, re-rendered here from its seed, against the oracle
            from aprilgrid_rs_amd import synth
            for r in range(1, world):
                fr, _ = synth.render_batch(sharding.shard_range(r, world, F)[0], 1, W, H, device=dev, fmt=args.format,
                                           pure_noise=args.noise)
                verify_against_oracle(sharding.unpack_frames(gs[r], gt[r])[:1], host_view(fr, args.format), "rank %d" % r)
            gather_check["oracle_checked_remote_frames"] = world - 1

    # separate, untimed pass for the per-kernel breakdown: one detector, strictly serial, events
    # around every launch (each kernel alone on the GPU)
    det = pipe.dets[0]
    out_s, out_t = sharding.alloc_result_buffers(F, dev, slab)
    prof = {}
    names = list(det.profile_read().keys())
    for k, name in enumerate(names):  # one kernel per pass, every 5th batch: everything else stays back to back
        det.set_option("profile_kernel", k)
        det.set_option("profile_stride", 5 if args.steps >= 20 else 1)
        det.profile_enable(1)
        det.profile_reset()
        for _ in range(30 if args.steps >= 20 else min(args.steps, 10)):
            det.saddles_batch_enqueue_to(frames, out_s, out_t)
        fence()
        prof[name] = det.profile_read()[name]
    prof = {k: v for k, v in prof.items() if v[1]}  # the launches this batch size takes (256 frames: K1, k_verify_seeds, k_sparse_frame)
    det.profile_enable(0)
    det.set_option("profile_kernel", 0)
    det.set_option("profile_stride", 1)
    rows_per_seg = det.get_option("k1_rows_per_segment")

    # N = 1: the same K steps again with several batches in flight (reported as "pipelined")
    pipelined = None
    if world == 1 and args.extra_pipeline > 1 and args.pipeline == 1:
        pipe2 = sharding.ChainPipeline(A.TagFamily.T36H11, F, dev, depth=args.extra_pipeline, dst=0, slab_records=slab)
        t_settle = time.perf_counter()  # same untimed settle as the main pass: fresh workspaces, clocks
        while True:
            for _ in range(max(args.warmup, 2 * pipe2.depth)):
                pipe2.submit(frames)
            pipe2.finish()
            torch.cuda.synchronize(dev)
            if (time.perf_counter() - t_settle) * 1e3 >= args.settle_ms:
                break
        blocks = []  # three timed blocks of K steps, the median reported: a short block right after the detectors were
        for _ in range(3):  # created has read 0.37 .. 0.46 ms on one build by box and moment (tools/exp/pipe_paths.py)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                pipe2.submit(frames)
            pipe2.finish()
            torch.cuda.synchronize(dev)
            blocks.append(time.perf_counter() - t1)
        dt2 = sorted(blocks)[1]
        pipelined = {"batches_in_flight": pipe2.depth, "ms_per_step": round(1e3 * dt2 / args.steps, 4),
                     "value": round(px_per_step_rank * args.steps / dt2 / 1e6, 1), "unit": "Mpix/s",
                     "blocks_ms_per_step": [round(1e3 * b / args.steps, 4) for b in blocks],
                     "note": "K1 of one batch overlaps the sparse kernels of the previous one (sharding.ChainPipeline); median of three blocks of K steps"}
        pipe2.close()

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # N > 1: the same K steps again with the slabs of --grouped-gather consecutive steps per collective (every rank
    # walks the same sequence; reported beside `value`, which stays on one collective per step)
    grouped = None
    if world > 1 and args.grouped_gather > 1 and args.gather_steps == 1 and args.gather_every <= 1 and args.pipeline == 1:
        pipe_g = sharding.ChainPipeline(A.TagFamily.T36H11, F, dev, depth=1, dst=0, slab_records=slab,
                                        detector_cls=rt.detector_cls, steps_per_gather=args.grouped_gather)

        def fence_g():
            pipe_g.finish()
            rt.synchronize(dev)
            dist.barrier()
            rt.synchronize(dev)

        for _ in range(max(args.warmup, 2 * args.grouped_gather)):
            pipe_g.submit(frames)
        fence_g()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            pipe_g.submit(frames)
        fence_g()
        tg = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        dtg = float(tg.item())
        grouped = {"steps_per_collective": pipe_g.gather.k, "ms_per_step": round(1e3 * dtg / args.steps, 4),
                   "value": round(px_per_step_rank * world * args.steps / dtg / 1e6, 1), "unit": "Mpix/s",
                   "note": "the same K steps, result slabs of %d consecutive steps per collective (results up to %d steps late): "
                           "never `value`" % (pipe_g.gather.k, pipe_g.gather.k - 1)}
        pipe_g.close()
        del pipe_g

    result = None
    if rank == 0:
        total_px = px_per_step_rank * world * args.steps
        ms_per_step = 1e3 * dt / args.steps
        mpix = total_px / dt / 1e6
        in_b = IN_BYTES[args.format]
        k1_alone_ms = prof["k_blur_hessian"][0] / max(prof["k_blur_hessian"][1], 1)
        if serial and k1_n:  # the blur kernel's figure of the breakdown is the one measured inside the timed region
            prof["k_blur_hessian"] = (k1_ms, k1_n)
        chain_ms = sum(v[0] / max(v[1], 1) for v in prof.values())
        traffic, traffic_src = args.pmc_traffic, "--pmc-traffic" if args.pmc_traffic else None
        if traffic is None and world == 1:
            traffic, traffic_src = recorded_traffic("%dx%dx%d_%s%s" % (F, W, H, args.format, "_noise" if args.noise else ""))
        a_mat = in_b + 12  # SURVEY.md 8(d) A_mat: input + blur write + response write + response re-read
        # the bytes THIS design is built to move per pixel (the response plane is never materialised): input +
        # blur write + mask write (K1) + mask re-read (K2) + 0.125 for the sparse stages' algorithmic gathers and
        # lists (about 970 nine-row windows and 5 800 3x3 re-tests per megapixel frame): 5.375 B/px for L8
        a_design = in_b + 4 + 0.125 + 0.125 + 0.125
        a_min = 2 * in_b  # SURVEY.md 8(d) A_min: the input read twice, nothing dense written
        roof = k1_roofline(px_per_step_rank, in_b, k1_ms, k1_n, traffic, traffic_src)
        # the same kernel alone on the GPU (serial pass below the timed region): with
        # --pipeline > 1 the timed launches share the chip with another step's sparse kernels
        if world == 1:  # the same box's plain copy rate, measured now: K1's HBM traffic against it
            copy_gbps = box_copy_gbps(torch, dev)
            roof["copy_GBps_this_box"] = round(copy_gbps, 1)
            if traffic:
                roof["traffic_frac_of_copy_this_box"] = round(traffic / (roof["avg_launch_ms"] * 1e-3) / 1e9 / copy_gbps, 4)
        # the CHAIN against the roofline -- bytes the design moves (and A_min beside it) over the
        # whole step as timed (first launch of K1 to the end of the last kernel, ms_per_step), against 8 TB/s
        step_s = ms_per_step * 1e-3  # every rank runs its own 256 frames in this time
        roof["chain_design_bytes_per_px"] = a_design
        roof["chain_GBps"] = round(px_per_step_rank * a_design / step_s / 1e9, 1)
        roof["chain_frac"] = round(px_per_step_rank * a_design / step_s / 1e9 / HBM_PEAK_GBPS, 4)
        roof["a_min_bytes_per_px"] = a_min
        roof["a_min_frac"] = round(px_per_step_rank * a_min / step_s / 1e9 / HBM_PEAK_GBPS, 4)
        if pipelined:  # the same over the step with several batches in flight (reported beside, never `value`)
            pipelined["chain_frac"] = round(px_per_step_rank * a_design / (pipelined["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            roof["chain_frac_in_flight"] = pipelined["chain_frac"]
        roof["chain_note"] = ("chain_frac = design bytes per GPU / ms_per_step / 8 TB/s (what the whole step achieves); frac = the blur "
                              "kernel alone; a_min_frac = the fused lower bound of SURVEY.md 8(d) over the same step time")
        if not serial:  # batches in flight: the timed launches shared the chip; the serial pass gives the kernel alone
            roof["alone_avg_launch_ms"] = round(k1_alone_ms, 5)
            roof["alone_frac"] = round(roof["bytes_per_launch"] / (k1_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        result = {
            "metric": ("STUB -- NOT A MEASUREMENT (CPU oracle behind the enqueue call, control-flow test) -- " if stub else "")
                      + "Mpix/s through the saddle chain (blur->threshold->gradient->saddle), frames resident in HBM",
            "backend": "cpu-stub" if stub else "hip-gfx950",
            "value": round(mpix, 1),
            "unit": "Mpix/s (cpu stub)" if stub else "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[%d]: %d synthetic %dx%d %s AprilGrid frames per GPU (T36H11 6x6 board, seeded "
                            "renderer, frame indices %d..%d per rank: %d distinct%s)%s" % (
                                1 if world == 1 else 2, F, W, H, args.format, 0, uniq - 1, uniq,
                                "" if uniq == F else " tiled", ", pure noise" if args.noise else ""),
                "frames_per_gpu": F, "width": W, "height": H, "format": args.format, "distinct_frames_per_gpu": uniq,
                "frames_per_s": round(F * world * args.steps / dt, 1),
                "saddles_per_frame": round(saddles_per_frame, 1),
                "clusters_per_frame": round(clusters_per_frame, 1),
                "parallelism": "frame-sharded x%d, RCCL gather of result slabs (%d steps per collective)" % (world, pipe.gather.k) if world > 1 else "1 GPU",
                "steps_per_gather": pipe.gather.k,
                "batches_in_flight": pipe.depth,
                "k1_rows_per_segment": rows_per_seg,
            },
            "roofline": roof,
            "chain": {
                "a_mat_bytes_per_px": a_mat,
                "kernel_ms_per_step": {k: round(v[0] / max(v[1], 1), 5) for k, v in prof.items()},
                "sparse_path": {1: "k_verify_seeds + k_flood_refine + k_rare (three launches)",
                                2: "k_sparse_frame alone (one workgroup per frame: verify, seeds, floods, refinement, emission)",
                                3: "k_verify_seeds + k_sparse_frame (one workgroup per frame: floods, refinement, emission)"}.get(det.get_option("last_sparse_path")),
                "sum_kernel_ms_per_step": round(chain_ms, 5),
                "a_mat_equivalent_GBps": round(px_per_step_rank * a_mat / (chain_ms * 1e-3) / 1e9, 1),
                "a_mat_equivalent_frac_of_peak": round(px_per_step_rank * a_mat / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "a_mat_equivalent_note": "NOT bytes moved: what a design that materialises the response plane (13 B/px) would "
                                         "have to sustain to finish in the same time; the bytes this design moves are in "
                                         "roofline.chain_frac",
                "design_bytes_per_px": a_design,
                "design_GBps": round(px_per_step_rank * a_design / (chain_ms * 1e-3) / 1e9, 1),
                "design_frac_of_peak": round(px_per_step_rank * a_design / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "frames_on_generic_path": generic_frames,
                "note": "kernel_ms_per_step: hipEvent pairs around ONE kernel of every 5th batch (everything else stays back "
                        "to back on the stream): the blur kernel inside the timed region (= roofline.avg_launch_ms), the "
                        "others in untimed passes of their own; each figure includes the ~2 us the pair itself costs; the "
                        "kernels' durations by rocprofv3 --kernel-trace --stats are under profiles/",
            },
            "gather_check": gather_check,
        }
        if per_step:
            result["ms_per_step_median"] = round(statistics.median(per_step), 4)
            result["ms_per_step_min"] = round(min(per_step), 4)
        if pipelined:
            result["pipelined"] = pipelined
        if grouped:
            result["grouped_gather"] = grouped
        if world == 1:
            host = host_view(frames[:uniq], args.format)
            if not args.no_verify:
                result["verified_frames"] = verify_against_oracle(own[:uniq], host, "main workload",
                                                                  threads=min(16, os.cpu_count() or 1))
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(host, args.format, args.cpu_seconds)
                result["cpu_baseline"]["host_cores_available"] = host_parallelism()  # = all_cores.cores: affinity mask or cgroup quota
                result["cpu_baseline"]["host_cores_shown"] = os.cpu_count()
    pipe.close()
    del pipe
    if world == 1 and rank == 0 and not args.no_extra:
        del frames, out_s, out_t
        torch.cuda.empty_cache()
        st = max(10, args.steps // 2)
        vf = 0 if args.no_verify else 1
        result["extra_configs"] = {
            "configs[3]_4K": extra_leg(torch, A, dev, "4K", "configs[3]: 32 synthetic 3840x2160 L8 frames (8 distinct, tiled: "
                                       "rendering a 4K frame costs 8 frames of 1280x800)", 32, 3840, 2160, "L8", 8, False, st,
                                       args.warmup, 2 * vf, settle_ms=args.settle_ms),
            "configs[4]_RGB8": extra_leg(torch, A, dev, "RGB8", "configs[4]: 256 synthetic 1280x800 RGB8 frames, HWC "
                                         "interleaved as kornia::Image<u8,3> (64 distinct, tiled)", 256, 1280, 800, "RGB8", 64,
                                         False, st, args.warmup, 64 * vf, settle_ms=args.settle_ms),
            "L16": extra_leg(torch, A, dev, "L16", "256 synthetic 1280x800 L16 frames (64 distinct, tiled)", 256, 1280, 800,
                             "L16", 64, False, st, args.warmup, 64 * vf, settle_ms=args.settle_ms),
            # frames that miss K1's aligned form: a width that is not a multiple of 4, rows that are not 4-byte aligned
            "unaligned_width": extra_leg(torch, A, dev, "1282 wide", "256 synthetic 1282x800 L8 frames (64 distinct, tiled), tightly packed (width % 4 = 2: rows start "
                                         "at odd multiples of 2 bytes; K1's unaligned-dword form; a batch that fills the chip like the headline's)", 256, 1282, 800, "L8", 64, False, st, args.warmup, 4 * vf, settle_ms=args.settle_ms),
            "unaligned_pitch": extra_leg(torch, A, dev, "pitch 1283", "256 synthetic 1280x800 L8 frames (64 distinct, tiled) cut out of an allocation with 1283 bytes "
                                         "per row (odd pitch; K1's unaligned-dword form)", 256, 1280, 800, "L8", 64, False, st, args.warmup, 4 * vf, pitch=1283, settle_ms=args.settle_ms),
            "pure_noise": extra_leg(torch, A, dev, "noise", "sensitivity row: 64 pure-noise 1280x800 L8 frames (16 distinct, "
                                    "tiled; ~7000 saddles per frame: worst case for the sparse stages)", 64, 1280, 800, "L8",
                                    16, True, st, args.warmup, 4 * vf, settle_ms=args.settle_ms),
        }
        result["host_boundary"] = soft_leg(host_boundary_leg, torch, A, dev, F, W, H, 20)
        result["detect_end_to_end"] = soft_leg(detect_end_to_end_leg, torch, A, dev, F, W, H, stream_frames=args.stream_frames)
        # BASELINE.json configs[0] and the reference's own bench shape (benches/bench_detection.rs:24-36): ONE frame
        # through detect -- latency, GPU path beside the oracle on this box's host -- and the 7-image table
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_images
        cpu = bench_images.Cpu()
        det1 = A.TagDetector(A.TagFamily.T36H11, None, device=dev.index)
        try:
            result["extra_configs"]["configs[0]_single_frame"] = soft_leg(bench_images.config0, det1, cpu, 9)
            result["reference_bench_detection"] = soft_leg(bench_images.detection_table, det1, cpu, 5)
        finally:
            det1.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)
    if world > 1 or coll1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
