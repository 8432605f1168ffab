"""Extra measurement rows for DESIGN.md / profiles (not the bench line): BASELINE.json configs
4 (4K) and 5 (RGB8), L16, pure-noise frames, single-frame latency through the host API, and
end-to-end detect() throughput with the host tail on N threads."""
import json, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth

def chain(det, frames, steps=10, warm=3):
    for _ in range(warm): det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): det.saddles_batch_enqueue(frames)
    det.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res, st = det.saddles_batch_fetch()
    px = frames.shape[0] * frames.shape[1] * frames.shape[2]
    return {"frames": int(frames.shape[0]), "ms_per_step": round(dt * 1e3, 4), "Mpix_s": round(px / dt / 1e6, 1),
            "frames_s": round(frames.shape[0] / dt, 1), "saddles_per_frame": round(float(np.mean([len(r) for r in res])), 1),
            "status_ok": bool((st == 0).all())}

out = {}
det = A.TagDetector("t36h11")
def tiled(first, uniq, n, w, h, **kw):
    b, _ = synth.render_batch(first, uniq, w, h, device="cuda", **kw)
    return b.repeat((n // uniq + 1,) + (1,) * (b.dim() - 1))[:n].contiguous()
out["config2_L8_1280x800_x256"] = chain(det, tiled(0, 32, 256, 1280, 800))
out["config5_RGB8_1280x800_x256"] = chain(det, tiled(0, 16, 256, 1280, 800, fmt="RGB8"))
out["L16_1280x800_x256"] = chain(det, tiled(0, 16, 256, 1280, 800, fmt="L16"))
out["noise_L8_1280x800_x64"] = chain(det, tiled(0, 8, 64, 1280, 800, pure_noise=True))
out["config4_4K_L8_3840x2160_x32"] = chain(det, tiled(0, 4, 32, 3840, 2160))
# config 1: single frame through the host API (PCIe copy in, results out)
one = tiled(0, 1, 1, 1280, 800)[0].cpu().numpy()
det.refined_saddle_points(one)
t0 = time.perf_counter()
for _ in range(50): det.refined_saddle_points(one, as_array=True)
out["single_frame_host_api_1280x800"] = {"ms_per_frame": round((time.perf_counter() - t0) / 50 * 1e3, 3)}
from PIL import Image
real = np.array(Image.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "images", "1520525725372653511.png"))).astype(np.uint16)
det.refined_saddle_points(real)
t0 = time.perf_counter()
for _ in range(50): s = det.refined_saddle_points(real, as_array=True)
out["config1_real_1024x1024_L16_host_api"] = {"ms_per_frame": round((time.perf_counter() - t0) / 50 * 1e3, 3), "saddles": int(len(s)), "tags": len(det.detect(real))}
# end-to-end detect(): GPU chain per batch + host tail on T threads
frames = tiled(0, 32, 256, 1280, 800)
host = frames.cpu().numpy()
det.saddles_batch_enqueue(frames); res, st = det.saddles_batch_fetch()
for T in (1, 8, 32):
    def work(lo, hi, acc):
        n = 0
        for i in range(lo, hi): n += len(A.TagDetector.detect_tail("t36h11", res[i], host[i]))
        acc.append(n)
    t0 = time.perf_counter()
    det.saddles_batch_enqueue(frames); res, st = det.saddles_batch_fetch()
    acc, th = [], []
    for t in range(T):
        th.append(threading.Thread(target=work, args=(t * 256 // T, (t + 1) * 256 // T, acc))); th[-1].start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    out["end_to_end_detect_256x1280x800_threads_%d" % T] = {"frames_s": round(256 / dt, 1), "tags_per_frame": round(sum(acc) / 256, 1)}
print(json.dumps(out, indent=1))
