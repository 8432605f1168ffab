#!/usr/bin/env python3
"""The reference's own bench shape on its own inputs, GPU path next to the CPU
oracle in one run:

  detection   benches/bench_detection.rs:7-36 -- the 7 images, decode OUTSIDE the timed region, time
              TagDetector::detect only: here det.detect(img) through the host API (PCIe upload, chain
              on the MI355X, saddles back, board search + decode on one host thread) vs the C oracle's
              detect (oracle/agx_oracle.c -O3 -march=native, 1 thread), median of >= 10 runs each.
  blur        benches/bench_blur.rs:20-46 -- the 3 images, img.to_luma32f() OUTSIDE the timed region,
              time gaussian_blur_f32(&luma, 1.5) only: here the blur kernel on the f32 plane resident in
              HBM (AGX_LF32; hipEvents around K1 -- K1 also evaluates the Hessian response, the running
              minimum and the candidate mask, a "blur only" ablation build of it is timed beside it) vs
              the oracle's gaussian_blur_f32, single frame and a batch of 64 copies.

    python tools/bench_images.py [--runs 15] [--no-batch] > profiles/r3_bench_images.json
(needs tests/golden/images, i.e. the reference's fixture PNGs; prints ONE JSON object.)
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DETECTION = ["iphone.png", "EuRoC.png", "TUM_VI.png", "right.png", "r45.png", "top.png", "two_boards.png"]
BLUR = ["iphone.png", "EuRoC.png", "TUM_VI.png"]


def median_ms(fn, runs):
    fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return round(statistics.median(ts), 4)


EXPECTED_TAGS = {"iphone.png": 66, "EuRoC.png": 36, "TUM_VI.png": 36, "right.png": 36, "r45.png": 36, "top.png": 36,
                 "two_boards.png": 72}  # the reference's own pins, /root/reference/tests/test_detector.rs:26-32


class Cpu:
    """The oracle built -O3 -march=native (kind: port), one thread."""

    def __init__(self):
        from oracle import oracle as O
        O.build()
        self.O = O
        import bench
        self.native = C.CDLL(bench.native_oracle_library()[0])  # -march=native of THIS host, not of the build container
        self.native.orc_detect.argtypes = O.lib().orc_detect.argtypes
        self.native.orc_gaussian_blur_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        self.native.orc_refined_saddle_points.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p,
                                                          C.c_void_p, C.c_int, C.c_void_p]
        self.prm = O.default_params()
        self.edge, self.border, self.hamming, _ = O.FAMILIES["T36H11"]
        self.codes = O.family_codes("T36H11")
        self.tags_buf = (O.Tag * 1024)()
        import numpy as np
        self.saddles = np.zeros(1 << 16, O.SADDLE_DTYPE)

    def detect(self, img):
        a, fmt, stride = self.O.image_fmt(img)
        h, w = a.shape[:2]
        return self.native.orc_detect(a.ctypes.data, w, h, stride, fmt, C.addressof(self.prm), self.border, self.edge,
                                      self.hamming, self.codes.ctypes.data, len(self.codes), self.tags_buf, 1024)

    def chain(self, img):
        a, fmt, stride = self.O.image_fmt(img)
        h, w = a.shape[:2]
        return self.native.orc_refined_saddle_points(a.ctypes.data, w, h, stride, fmt, C.addressof(self.prm),
                                                     self.saddles.ctypes.data, len(self.saddles), None)


def detect_row(det, cpu, img, runs, cpu_runs=None):
    """One image the way benches/bench_detection.rs:24-36 times it (decode outside, `detect` inside), GPU path
    beside the oracle, plus the saddle chain alone through the host-image entry (PCIe included)."""
    a, fmt, _ = cpu.O.image_fmt(img)
    h, w = a.shape[:2]
    n_cpu = cpu.detect(img)
    n_gpu = len(det.detect(img))
    assert n_cpu == n_gpu, (n_cpu, n_gpu)
    g_chain = median_ms(lambda: det.refined_saddle_points(img, as_array=True), runs)
    g = median_ms(lambda: det.detect(img), runs)
    c_chain = median_ms(lambda: cpu.chain(img), cpu_runs or max(3, runs // 2))
    c = median_ms(lambda: cpu.detect(img), cpu_runs or max(3, runs // 2))
    return {"size": "%dx%d" % (w, h), "format": {0: "L8", 1: "L16", 2: "RGB8"}[fmt], "tags": n_gpu,
            "gpu_detect_ms": g, "gpu_saddle_chain_ms_incl_pcie": g_chain, "cpu_detect_ms": c, "cpu_saddle_chain_ms": c_chain,
            "speedup": round(c / g, 2)}


def detection_table(det, cpu, runs, names=DETECTION):
    """benches/bench_detection.rs:7-36: the 7 images."""
    from tests.util import load_image
    out = {}
    for name in names:
        out[name] = detect_row(det, cpu, load_image(name), runs)
        if name in EXPECTED_TAGS:
            assert out[name]["tags"] == EXPECTED_TAGS[name], (name, out[name]["tags"])
    return out


def config0(det, cpu, runs):
    """BASELINE.json configs[0]: ONE frame through TagDetector::detect -- the reference's
    data/1520525725372653511.png (1024x1024, 16-bit grey) and, beside it, one synthetic
    1280x800 u8 frame of the bench's generator: latency of the GPU chain, of the GPU path's detect and of
    the oracle's (1 thread), same process, same box."""
    from tests.util import load_image
    from aprilgrid_rs_amd import synth
    out = {"data/1520525725372653511.png": detect_row(det, cpu, load_image("1520525725372653511.png"), runs)}
    fr, _ = synth.render_batch(0, 1, 1280, 800, device="cuda")
    out["synthetic_1280x800_L8_frame0"] = detect_row(det, cpu, fr[0].cpu().numpy(), runs)
    return out


def blur_table(det, cpu, runs, names=BLUR):
    """benches/bench_blur.rs:20-46."""
    import numpy as np
    import torch
    from tests.util import load_image
    O, native = cpu.O, cpu.native
    out = {}
    for name in names:
        img = load_image(name)
        luma = O.luma_f32(img)  # to_luma32f, outside the timed region
        h, w = luma.shape
        blur_out = np.empty_like(luma)
        c = median_ms(lambda: native.orc_gaussian_blur_f32(luma.ctypes.data, w, h, 1.5, blur_out.ctypes.data), runs)
        row = {"size": "%dx%d" % (w, h), "cpu_gaussian_blur_f32_ms": c}
        for label, nb in (("single_frame", 1), ("batch_64", 64)):
            planes = torch.from_numpy(luma).cuda()[None].repeat(nb, 1, 1).contiguous()
            for mode, dbg in (("k1_fused", 0), ("k1_blur_only_ablation", 4)):
                det.set_option("debug_ablation", dbg)
                for _ in range(3):
                    det.saddles_batch_enqueue(planes)
                det.sync()
                det.profile_enable(1)
                det.profile_reset()
                for _ in range(runs):
                    det.saddles_batch_enqueue(planes)
                ms, n = det.profile_read()["k_blur_hessian"]
                det.profile_enable(0)
                row["gpu_%s_%s_ms_per_frame" % (mode, label)] = round(ms / n / nb, 5)
            det.set_option("debug_ablation", 0)
            if nb == 1:  # the timed kernel's blur plane is the oracle's, bit for bit
                det.saddles_batch_enqueue(planes)
                det.sync()
                got = det.debug_fetch(0, "blur", (h, w))
                native.orc_gaussian_blur_f32(luma.ctypes.data, w, h, 1.5, blur_out.ctypes.data)
                assert np.array_equal(got.view(np.uint32), blur_out.view(np.uint32)), name
                row["blur_plane_bit_exact"] = True
        row["speedup_single_frame_fused"] = round(c / row["gpu_k1_fused_single_frame_ms_per_frame"], 1)
        row["speedup_batch_64_fused"] = round(c / row["gpu_k1_fused_batch_64_ms_per_frame"], 1)
        out[name] = row
    return out


def detect_batch_table(det, thread_counts=None):
    """End-to-end detect() over a batch (agx_detect_batch): chain per chunk on the device, board search +
    decode on a pool of host threads -- frames per second by thread count (the host tail is the
    reference's exhaustive search: this rate is set by the host, never the bench's `value`)."""
    from aprilgrid_rs_amd import synth, _ffi
    if thread_counts is None:  # up to what the process may keep busy (affinity mask or cgroup CPU quota: 16 on a 1-GPU box of this pool)
        quota = int(_ffi.lib().agx_host_parallelism())
        thread_counts = sorted(set(t for t in (1, 8, 16, 32, 64) if t < quota) | {quota})
    fr, _ = synth.render_batch(0, 256, 1280, 800, device="cuda")
    host = fr.cpu().numpy()
    out = {}
    for thr in thread_counts:
        n = 32 if thr == 1 else 256
        det.detect_batch(host[:8], n_threads=thr)  # pool start-up outside the timed call
        t0 = time.perf_counter()
        tags = det.detect_batch(host[:n], n_threads=thr)
        dt = time.perf_counter() - t0
        out["threads_%d" % thr] = {"frames": n, "frames_per_s": round(n / dt, 1),
                                   "tags_per_frame": round(sum(len(t) for t in tags) / n, 1)}
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=15)
    ap.add_argument("--no-batch", action="store_true", help="skip the detect_batch thread sweep")
    args = ap.parse_args(argv)
    import aprilgrid_rs_amd as A
    cpu = Cpu()
    det = A.TagDetector("t36h11", None, device=0)
    out = {"runs": args.runs, "cpu": "oracle/agx_oracle.c -O3 -march=native -ffp-contract=off, 1 thread (kind: port)"}
    out["configs[0]"] = config0(det, cpu, args.runs)
    out["detection"] = detection_table(det, cpu, args.runs)
    out["blur"] = blur_table(det, cpu, args.runs)
    if not args.no_batch:
        out["detect_batch_256x1280x800_L8"] = detect_batch_table(det)
    det.close()
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
