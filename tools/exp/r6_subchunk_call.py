"""VERDICT r5 next #6, measured before anything is built into the library: what ONE 256-frame agx_detect_batch call could
gain from sub-chunks whose chain + device tail run (each on a detector / stream of its own) under the upload of the
sub-chunks behind them.

The prototype does in Python what the C call would do: the main thread uploads the sub-chunks in order (torch copy on a side
stream from pageable or pinned host memory), one thread per sub-chunk waits for its upload and calls agx_detect_batch on its
own detector with the frames already on the device (d_frames: chain + board search + decode on that detector's stream,
tags back in host arrays).  Everything the call needs beyond that (one workspace per sub-chunk in flight, result tables per
sub-chunk) is what the prototype's S detectors already pay.  Tags are compared with the single call's.

    python tools/exp/r6_subchunk_call.py [reps]        -> table + one JSON line (profiles/rejected/r6_subchunk_call.txt)"""
import json
import os
import statistics
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import aprilgrid_rs_amd as A  # noqa: E402
from aprilgrid_rs_amd import _ffi, synth  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 7
N, W, H, CAP = 256, 1280, 800, 64
dev = torch.device("cuda", 0)
fr, _ = synth.render_batch(0, N, W, H, device=dev)
host_pageable = fr.cpu().numpy()
pinned_t = torch.empty((N, H, W), dtype=torch.uint8, pin_memory=True)
pinned_t.copy_(fr.cpu())
host_pinned = pinned_t.numpy()
quota = int(_ffi.lib().agx_host_parallelism())


def single_call(det, host):
    out = np.zeros((N, CAP), det.TAG_DTYPE)
    cnt = np.zeros(N, np.uint32)
    st = np.zeros(N, np.int32)
    det.detect_batch_raw(host, n_threads=quota, cap=CAP, out=out, counts=cnt, status=st)
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        rc, _, _, _ = det.detect_batch_raw(host, n_threads=quota, cap=CAP, out=out, counts=cnt, status=st)
        ts.append(time.perf_counter() - t0)
        assert rc == 0 and (st == 0).all()
    return statistics.median(ts), min(ts), out.copy(), cnt.copy()


def sub_chunk_call(dets, sizes, host, stage, copy_stream):
    bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    assert bounds[-1] == N
    out = np.zeros((N, CAP), dets[0].TAG_DTYPE)
    cnt = np.zeros(N, np.uint32)
    st = np.zeros(N, np.int32)
    host_t = torch.from_numpy(host)

    def once():
        events = [torch.cuda.Event() for _ in sizes]
        errors = []

        def work(k):
            a, b = bounds[k], bounds[k + 1]
            events[k].synchronize()  # the sub-chunk is on the device
            rc, _, _, _ = dets[k].detect_batch_raw(host[a:b], n_threads=2, cap=CAP, device_frames=stage[a:b], out=out[a:b], counts=cnt[a:b], status=st[a:b])
            if rc != 0:
                errors.append((k, rc))

        threads = [threading.Thread(target=work, args=(k,)) for k in range(len(sizes))]
        t0 = time.perf_counter()
        with torch.cuda.stream(copy_stream):
            for k in range(len(sizes)):
                a, b = bounds[k], bounds[k + 1]
                stage[a:b].copy_(host_t[a:b], non_blocking=True)
                events[k].record(copy_stream)
                threads[k].start()  # (started once its upload is queued: the thread waits on the event)
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        assert not errors and (st == 0).all(), errors
        return dt

    once()
    once()
    ts = [once() for _ in range(REPS)]
    return statistics.median(ts), min(ts), out.copy(), cnt.copy()


base = A.TagDetector("t36h11", None, device=0)
base.set_option("device_tail", 1)
rows = []
for name, host in (("pageable", host_pageable), ("pinned", host_pinned)):
    med, best, ref_out, ref_cnt = single_call(base, host)
    rows.append({"input": name, "scheme": "one call (the library today)", "ms_median": round(1e3 * med, 3), "ms_min": round(1e3 * best, 3),
                 "frames_per_s": round(N / med, 1)})
    stage = torch.empty((N, H, W), dtype=torch.uint8, device=dev)
    copy_stream = torch.cuda.Stream(dev)
    for sizes in ([256], [128, 128], [64] * 4, [96, 64, 48, 32, 16], [32] * 8, [128, 64, 32, 16, 8, 8]):
        dets = []
        for s in sizes:
            d = A.TagDetector("t36h11", None, device=0)
            d.set_option("device_tail", 1)
            dets.append(d)
        try:
            med, best, out, cnt = sub_chunk_call(dets, sizes, host, stage, copy_stream)
        finally:
            for d in dets:
                d.close()
        same = bool(np.array_equal(cnt, ref_cnt) and all(out[f, : cnt[f]].tobytes() == ref_out[f, : ref_cnt[f]].tobytes() for f in range(N)))
        rows.append({"input": name, "scheme": "sub-chunks %s, a detector + thread each, uploads in order" % sizes, "ms_median": round(1e3 * med, 3),
                     "ms_min": round(1e3 * best, 3), "frames_per_s": round(N / med, 1), "tags_equal_single_call": same})
        torch.cuda.synchronize()
base.close()
print("%-9s %-75s %9s %9s %11s %s" % ("input", "scheme", "ms median", "ms min", "frames/s", "tags equal"))
for r in rows:
    print("%-9s %-75s %9.3f %9.3f %11.1f %s" % (r["input"], r["scheme"], r["ms_median"], r["ms_min"], r["frames_per_s"], r.get("tags_equal_single_call", "")))
print(json.dumps({"frames": N, "reps": REPS, "host_threads_granted": quota, "rows": rows}))
