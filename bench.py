#!/usr/bin/env python3
"""bench.py -- throughput of the AprilGrid saddle chain on MI355X.

A "step" is one pass of the hot path (luma -> blur -> Hessian response -> min/threshold ->
clustering -> rochade_refine -> k/phi filter, kernels K1..K5) over one batch of synthetic
1280x800 u8 frames that is already resident in HBM.  N = 1 is BASELINE.json configs[1]
(256 frames on one MI355X); N > 1 is configs[2]: every rank owns 256 frames (weak scaling,
2048 frames on 8 GPUs) and the only collective is the per-step RCCL gather of the result
slabs to rank 0.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  value = pixels of all ranks through the chain per second
(Mpix/s), inputs resident in HBM.  roofline = K1 (the dominant kernel): algorithmic bytes per
launch / average launch duration from hipEvents recorded on the launch stream inside the
timed region.  cpu_baseline = the C oracle (port of the reference CPU path, 1 thread; all_cores =
the same frame-parallel on the host cores) on a bounded sample of the same frames, rank 0, N = 1
only.  The timed region is strictly serial (one batch at a time); at N = 1 a second pass with
--extra-pipeline batches in flight (sharding.ChainPipeline) is reported as "pipelined".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec)
HBM_MEASURED_GBPS = 6290.0    # float4-copy ceiling measured on MI355X (same guide)
IN_BYTES = {"L8": 1, "L16": 2, "RGB8": 3}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
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
    ap.add_argument("--noise", action="store_true", help="pure-noise frames (sensitivity row)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="batches in flight per GPU in the timed region (detectors on separate HIP streams; "
                         "1 = strictly serial: value, ms_per_step and roofline then describe the same launches)")
    ap.add_argument("--extra-pipeline", type=int, default=3,
                    help="N = 1 only: a second, separately reported timed pass with this many batches in flight (0 = skip)")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed: keep the chain running this long before the W warm-up steps so that workspace "
                         "allocation is done and the GPU clocks have ramped (0 = off)")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per K1 launch from a separate rocprofv3 --pmc run (profiles/)")
    return ap.parse_args()


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


def cpu_baseline(frames_host, fmt, budget_s):
    """Oracle (C port of the reference CPU path, -O3 -march=native, still no FMA contraction),
    one thread, chain only (refined_saddle_points), on as many of the bench's own frames as fit
    in the budget."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    O.build()
    native = os.path.join(ROOT, "oracle", "liborc_native.so")
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
    n_thr = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64))
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
                          "sample": "%d frames in %.1f s on %d threads" % (sum(counts), t_mt, n_thr)},
            "sample": "%d frames %dx%d %s, chain only (refined_saddle_points), oracle/agx_oracle.c -O3 "
                      "-march=native -ffp-contract=off, %.1f s" % (n_done, w, h, fmt, t_used),
            "ms_per_frame": round(1e3 * t_used / n_done, 3)}


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import aprilgrid_rs_amd as A
    from aprilgrid_rs_amd import synth, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (WORLD_SIZE=%d)" % (args.gpus, world))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    W, H, F = args.width, args.height, args.frames
    # ---- synthetic workload: frames [rank*F, rank*F + F) of the seeded generator, rendered
    # straight into HBM; `unique` distinct frames tiled (rendering is outside the timed region)
    first_frame, _ = sharding.shard_range(rank, world, F)
    frames, uniq = make_workload(first_frame, F, W, H, args.format, args.unique, args.noise, dev)
    px_per_step_rank = F * W * H

    # `--pipeline` detectors (own workspace + HIP stream each) take the steps in turn: K1 of step
    # i+1 overlaps the short, latency-bound sparse kernels of step i.  Result buffers stay in
    # HBM; for N > 1 they are gathered to rank 0 asynchronously (RCCL over xGMI) -- the one
    # collective of the path -- so the gather of a step overlaps the chain of the next.
    pipe = sharding.ChainPipeline(A.TagFamily.T36H11, F, dev, depth=args.pipeline, dst=0)

    def step():
        pipe.submit(frames)

    def fence():
        pipe.finish()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # setup, untimed: first call allocates the workspace; then run until the clocks have settled
    step()
    fence()
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(8):
            step()
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    tb = pipe.last_table.cpu().numpy()
    assert (tb[:, 2] & 7 == 0).all(), "capacity overflow in the bench workload: %s" % tb[tb[:, 2] != 0][:4]
    generic_frames = int(((tb[:, 2] & 16) != 0).sum())
    saddles_per_frame = float(tb[:, 0].mean())
    clusters_per_frame = float(tb[:, 3].mean())

    # timed region: hipEvents (on the launch stream) around K1 only -- 2 records per step
    for d in pipe.dets:
        d.profile_enable(1)
        d.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    k1_ms, k1_n = 0.0, 0
    for d in pipe.dets:
        ms, n = d.profile_read()["k_blur_hessian"]
        k1_ms, k1_n = k1_ms + ms, k1_n + n
        d.profile_enable(0)
    # separate, untimed pass for the per-kernel breakdown: one detector, strictly serial, events
    # around every launch (each kernel alone on the GPU)
    det = pipe.dets[0]
    out_s, out_t = sharding.alloc_result_buffers(F, dev)
    det.profile_enable(2)
    det.profile_reset()
    for _ in range(min(args.steps, 10)):
        det.saddles_batch_enqueue_to(frames, out_s, out_t)
    fence()
    prof = det.profile_read()
    det.profile_enable(0)

    # N = 1: the same K steps again with several batches in flight (reported as "pipelined")
    pipelined = None
    if world == 1 and args.extra_pipeline > 1 and args.pipeline == 1:
        pipe2 = sharding.ChainPipeline(A.TagFamily.T36H11, F, dev, depth=args.extra_pipeline, dst=0)
        t_settle = time.perf_counter()  # same untimed settle as the main pass: fresh workspaces, clocks
        while True:
            for _ in range(max(args.warmup, 2 * pipe2.depth)):
                pipe2.submit(frames)
            pipe2.finish()
            torch.cuda.synchronize(dev)
            if (time.perf_counter() - t_settle) * 1e3 >= args.settle_ms:
                break
        t1 = time.perf_counter()
        for _ in range(args.steps):
            pipe2.submit(frames)
        pipe2.finish()
        torch.cuda.synchronize(dev)
        dt2 = time.perf_counter() - t1
        pipelined = {"batches_in_flight": pipe2.depth, "ms_per_step": round(1e3 * dt2 / args.steps, 4),
                     "value": round(px_per_step_rank * args.steps / dt2 / 1e6, 1), "unit": "Mpix/s",
                     "note": "K1 of one batch overlaps the sparse kernels of the previous one (sharding.ChainPipeline)"}
        pipe2.close()

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    result = None
    if rank == 0:
        total_px = px_per_step_rank * world * args.steps
        ms_per_step = 1e3 * dt / args.steps
        mpix = total_px / dt / 1e6
        in_b = IN_BYTES[args.format]
        k1_avg_ms = k1_ms / max(k1_n, 1)
        k1_alone_ms = prof["k_blur_hessian"][0] / max(prof["k_blur_hessian"][1], 1)
        # K1 reads the input once, writes the blur plane (f32) and 1 bit / px of candidate mask
        k1_bytes = px_per_step_rank * (in_b + 4 + 0.125)
        k1_gbps = k1_bytes / (k1_avg_ms * 1e-3) / 1e9
        chain_ms = sum(v[0] / max(v[1], 1) for v in prof.values())
        # HBM bytes per K1 launch from rocprofv3 PMC passes (FETCH_SIZE x2 per the calibration,
        # + WRITE_SIZE), collected separately (tools/final_profile.sh) and kept under profiles/
        traffic, traffic_src = args.pmc_traffic, "--pmc-traffic" if args.pmc_traffic else None
        if traffic is None and world == 1:
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "k1_traffic.json")))
                key = "%dx%dx%d_%s" % (F, W, H, args.format)
                if key in rec:
                    traffic, traffic_src = rec[key]["bytes_per_launch"], rec[key]["source"]
            except Exception:
                pass
        a_mat = in_b + 12  # SURVEY.md 8(d) A_mat: input + blur write + response write + response re-read
        a_design = in_b + 4 + 0.125 + 0.125  # this design: input + blur write + mask write (K1) + mask read (K2);
        # the sparse stages (verify / refine gathers at ~2.4 % of the pixels, lists) add < 0.5 B/px
        result = {
            "metric": "Mpix/s through the saddle chain (blur->threshold->gradient->saddle), frames resident in HBM",
            "value": round(mpix, 1),
            "unit": "Mpix/s",
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
                "workload": "configs[%d]: %d synthetic %dx%d %s AprilGrid frames per GPU (T36H11 6x6 board, "
                            "seeded renderer, %d distinct frames tiled)%s" % (
                                1 if world == 1 else 2, F, W, H, args.format, uniq,
                                ", pure noise" if args.noise else ""),
                "frames_per_gpu": F, "width": W, "height": H, "format": args.format,
                "frames_per_s": round(F * world * args.steps / dt, 1),
                "saddles_per_frame": round(saddles_per_frame, 1),
                "clusters_per_frame": round(clusters_per_frame, 1),
                "parallelism": "frame-sharded x%d, RCCL gather of result slabs" % world if world > 1 else "1 GPU",
                "batches_in_flight": pipe.depth,
            },
            "roofline": {
                "kernel": "k_blur_hessian (K1)",
                "bound": "hbm",
                "achieved": round(k1_gbps, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(k1_gbps / HBM_PEAK_GBPS, 4),
                "frac_of_measured_copy_ceiling": round(k1_gbps / HBM_MEASURED_GBPS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_px": in_b + 4.125,
                "bytes_per_launch": k1_bytes,
                "avg_launch_ms": round(k1_avg_ms, 5),
                "launches_timed": k1_n,
                # the same kernel alone on the GPU (serial pass below the timed region): with
                # --pipeline > 1 the timed launches share the chip with another step's sparse kernels
                "alone_avg_launch_ms": round(k1_alone_ms, 5),
                "alone_frac": round(k1_bytes / (k1_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            },
            "chain": {
                "a_mat_bytes_per_px": a_mat,
                "kernel_ms_per_step": {k: round(v[0] / max(v[1], 1), 5) for k, v in prof.items()},
                "sum_kernel_ms_per_step": round(chain_ms, 5),
                "a_mat_GBps": round(px_per_step_rank * a_mat / (chain_ms * 1e-3) / 1e9, 1),
                "a_mat_frac_of_peak": round(px_per_step_rank * a_mat / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "design_bytes_per_px": a_design,
                "design_GBps": round(px_per_step_rank * a_design / (chain_ms * 1e-3) / 1e9, 1),
                "frames_on_generic_path": generic_frames,
            },
        }
        if pipelined:
            result["pipelined"] = pipelined
        if world == 1 and not args.no_cpu_baseline:
            sample = frames[:uniq].cpu().numpy()
            if args.format == "L16":
                sample = sample.view(np.uint16)
            result["cpu_baseline"] = cpu_baseline(sample, args.format, args.cpu_seconds)
            result["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        print(json.dumps(result), flush=True)
    pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
