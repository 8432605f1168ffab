"""What a 1 MB frame costs on its way to the device: pageable hipMemcpy2D (what agx_refined_saddle_points does), pageable
hipMemcpy, pinned hipMemcpyAsync, host memcpy into a pinned buffer; and the chain on a resident frame."""
import os, sys, time, statistics, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
import aprilgrid_rs_amd as A
from aprilgrid_rs_amd import synth
img = np.ascontiguousarray(np.asarray(synth.render_frame(0, 1280, 800)[0]))
hip = C.CDLL("libamdhip64.so")
def med(f, n=200):
    for _ in range(20): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e6 * statistics.median(ts)
dev = torch.empty(img.size, dtype=torch.uint8, device="cuda")
pin = torch.empty(img.size, dtype=torch.uint8).pin_memory()
pin_np = pin.numpy()
src = img.reshape(-1)
dptr, hptr, pptr = C.c_void_p(dev.data_ptr()), C.c_void_p(src.ctypes.data), C.c_void_p(pin.data_ptr())
n = C.c_size_t(img.size)
print("pageable hipMemcpy H2D 1 MB:            %.1f us" % med(lambda: hip.hipMemcpy(dptr, hptr, n, 1)))
print("pageable hipMemcpy2D (1280 x 800):      %.1f us" % med(lambda: hip.hipMemcpy2D(dptr, C.c_size_t(1280), hptr, C.c_size_t(1280), C.c_size_t(1280), C.c_size_t(800), 1)))
def pinned():
    hip.hipMemcpyAsync(dptr, pptr, n, 1, None); hip.hipStreamSynchronize(None)
print("pinned hipMemcpyAsync + sync 1 MB:      %.1f us" % med(pinned))
print("host memcpy 1 MB into pinned:           %.1f us" % med(lambda: np.copyto(pin_np, src)))
def both():
    np.copyto(pin_np, src); hip.hipMemcpyAsync(dptr, pptr, n, 1, None); hip.hipStreamSynchronize(None)
print("memcpy into pinned + async copy + sync: %.1f us" % med(both))
def chunked(k=4):
    c = img.size // k
    for i in range(k):
        np.copyto(pin_np[i * c:(i + 1) * c], src[i * c:(i + 1) * c])
        hip.hipMemcpyAsync(C.c_void_p(dev.data_ptr() + i * c), C.c_void_p(pin.data_ptr() + i * c), C.c_size_t(c), 1, None)
    hip.hipStreamSynchronize(None)
print("the same in 4 chunks, pipelined:        %.1f us" % med(chunked))
det = A.TagDetector("t36h11")
fr = dev.view(1, 800, 1280)
def chain():
    det.saddles_batch_enqueue(fr); det.saddles_batch_fetch()
print("chain on the resident frame + fetch:    %.1f us" % med(chain))
print("agx_refined_saddle_points (host image): %.1f us" % med(lambda: det.refined_saddle_points(img, as_array=True)))
