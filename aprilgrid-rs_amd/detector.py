"""Host-side mirror of aprilgrid::detector (reference src/detector.rs) over the C ABI.

Same names, argument meaning and error behaviour as the reference where Python allows:
  TagFamily / TagFamily.from_str      src/tag_families.rs:5-28
  DetectorParams.default_params()     src/detector.rs:25-41
  Saddle                              src/saddle.rs:3-15
  TagDetector.new / __init__          src/detector.rs:364-406
  TagDetector.refined_saddle_points   src/detector.rs:408-446
  TagDetector.detect                  src/detector.rs:505-540
  TagDetector.detect_kornia           src/detector.rs:478-503
An image is a numpy array standing for the DynamicImage variants the reference is fed:
HxW uint8 (ImageLuma8), HxW uint16 (ImageLuma16), HxWx3 uint8 (ImageRgb8).
"""
import ctypes as C
import enum
from collections import namedtuple
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import LIB_PATH, build_library

SADDLE_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("k", "f4"), ("theta", "f4"), ("phi", "f4")])
_CLUSTER_DTYPE = np.dtype([("first_index", "u4"), ("size", "u4"), ("cx", "f4"), ("cy", "f4")])

Saddle = namedtuple("Saddle", ["p", "k", "theta", "phi"])  # p = (x, y)


def library_path():
    return LIB_PATH


class AgxError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = _ffi.lib().agx_status_string(status).decode()
        super().__init__("%s (%d)%s" % (msg, status, (": " + detail) if detail else ""))


class TagFamily(enum.IntEnum):
    T16H5 = 0
    T25H7 = 1
    T25H9 = 2
    T36H11 = 3
    T36H11B1 = 4

    @staticmethod
    def from_str(s):
        out = C.c_int(-1)
        st = _ffi.lib().agx_family_from_str(s.encode(), C.byref(out))
        if st != _ffi.AGX_OK:
            raise ValueError("unknown tag family %r" % (s,))  # reference: Err(std::fmt::Error)
        return TagFamily(out.value)


@dataclass
class DetectorParams:
    tag_spacing_ratio: float = 0.3
    min_saddle_angle: float = 30.0
    max_saddle_angle: float = 60.0
    max_num_of_boards: int = 2

    @staticmethod
    def default_params():
        p = _ffi.Params()
        _ffi.lib().agx_default_params(C.byref(p))
        return DetectorParams(p.tag_spacing_ratio, p.min_saddle_angle, p.max_saddle_angle, p.max_num_of_boards)

    def _c(self):
        return _ffi.Params(self.tag_spacing_ratio, self.min_saddle_angle, self.max_saddle_angle,
                           self.max_num_of_boards)


def _image_args(img):
    a = np.ascontiguousarray(img)
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
        a = np.ascontiguousarray(a)
    if a.ndim == 2 and a.dtype == np.uint8:
        return a, _ffi.AGX_L8, a.shape[1]
    if a.ndim == 2 and a.dtype == np.uint16:
        return a, _ffi.AGX_L16, a.shape[1] * 2
    if a.ndim == 3 and a.shape[2] == 3 and a.dtype == np.uint8:
        return a, _ffi.AGX_RGB8, a.shape[1] * 3
    if a.ndim == 2 and a.dtype == np.float32:  # the caller's own to_luma32f plane (any DynamicImage variant)
        return a, _ffi.AGX_LF32, a.shape[1] * 4
    raise AgxError(_ffi.AGX_ERR_FORMAT, "image must be HxW uint8/uint16/float32 or HxWx3 uint8, got %s %s"
                   % (a.shape, a.dtype))


class TagDetector:
    """aprilgrid::detector::TagDetector on one MI355X (one handle = one device + stream;
    use one instance per thread)."""

    def __init__(self, tag_family, optional_detector_params=None, device=0):
        self._lib = _ffi.lib()
        self._h = C.c_void_p()
        self._saddle_buf = None  # refined_saddle_points' output buffer: (array [cap][5] f32, its address, count word)
        fam = TagFamily.from_str(tag_family) if isinstance(tag_family, str) else TagFamily(tag_family)
        prm = optional_detector_params._c() if optional_detector_params is not None else None
        st = self._lib.agx_detector_create(int(fam), C.byref(prm) if prm is not None else None, int(device),
                                           C.byref(self._h))
        if st != _ffi.AGX_OK:
            self._h = C.c_void_p()
            raise AgxError(st, "agx_detector_create(device=%d): %s" % (device, self._lib.agx_last_error(None).decode()))
        self.tag_family = fam
        self.detector_params = optional_detector_params or DetectorParams.default_params()
        self.device = device
        self._batch = None

    new = classmethod(lambda cls, tag_family, optional_detector_params=None, device=0:
                      cls(tag_family, optional_detector_params, device))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.agx_detector_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != _ffi.AGX_OK:
            raise AgxError(st, self._lib.agx_last_error(self._h).decode())

    # ---- reference API -------------------------------------------------------------------
    def refined_saddle_points(self, img, as_array=False, cap=16384):
        """-> Vec<Saddle> (list of Saddle) or, with as_array=True, a SADDLE_DTYPE array."""
        a, fmt, stride = _image_args(img)
        h, w = a.shape[:2]
        # the library writes into a buffer this handle keeps (plain f32 rows: allocating and slicing a structured array per call
        # cost 12 us of a 116 us call); the caller gets its own copy
        buf = self._saddle_buf
        if buf is None or buf[0].shape[0] < cap:
            arr = np.empty((cap, 5), np.float32)
            buf = self._saddle_buf = (arr, arr.ctypes.data, C.c_uint32(0))
        arr, ptr, n = buf
        img_ptr = a.__array_interface__["data"][0]
        st = self._lib.agx_refined_saddle_points(self._h, img_ptr, w, h, stride, fmt, ptr, arr.shape[0], C.byref(n))
        if st == _ffi.AGX_ERR_CAPACITY and n.value > arr.shape[0]:  # the reference's Vec has no limit: retry with room
            arr = np.empty((int(n.value), 5), np.float32)
            self._saddle_buf = (arr, arr.ctypes.data, n)
            ptr = arr.ctypes.data
            st = self._lib.agx_refined_saddle_points(self._h, img_ptr, w, h, stride, fmt, ptr, arr.shape[0], C.byref(n))
        self._check(st)
        res = arr[: n.value].copy().view(SADDLE_DTYPE).reshape(-1)
        if as_array:
            return res
        return [Saddle((float(s["x"]), float(s["y"])), float(s["k"]), float(s["theta"]), float(s["phi"]))
                for s in res]

    def detect(self, img, cap=4096):
        """-> HashMap<u32, [(f32,f32);4]> as {tag_id: 4x2 float32 array}."""
        a, fmt, stride = _image_args(img)
        h, w = a.shape[:2]
        out = (_ffi.TagC * cap)()
        n = C.c_uint32(0)
        self._check(self._lib.agx_detect(self._h, a.ctypes.data, w, h, stride, fmt, out, cap, C.byref(n)))
        return {int(out[i].id): np.array(out[i].xy, np.float32).reshape(4, 2) for i in range(n.value)}

    def detect_planes(self, luma32f, luma8, cap=4096):
        """detect() of any DynamicImage variant from its two planes: img.to_luma32f() (HxW float32)
        for the saddle chain and img.to_luma8() (HxW uint8) for the decode."""
        f = np.ascontiguousarray(luma32f, np.float32)
        g = np.ascontiguousarray(luma8, np.uint8)
        assert f.shape == g.shape and f.ndim == 2
        h, w = f.shape
        out = (_ffi.TagC * cap)()
        n = C.c_uint32(0)
        self._check(self._lib.agx_detect_planes(self._h, f.ctypes.data, w * 4, g.ctypes.data, w, w, h, out, cap,
                                                C.byref(n)))
        return {int(out[i].id): np.array(out[i].xy, np.float32).reshape(4, 2) for i in range(n.value)}

    TAG_DTYPE = np.dtype([("id", "u4"), ("xy", "f4", (8,))])

    def detect_batch_raw(self, frames, n_threads=0, cap=1024, device_frames=None, out=None, counts=None, status=None):
        """agx_detect_batch as a C / Rust caller uses it: frames = numpy [N,H,W] uint8 / uint16 or [N,H,W,3] uint8 in
        host memory (optionally also resident on the GPU as the torch tensor device_frames); the tags go into the
        caller's arrays out [N, cap] of TAG_DTYPE, counts [N] uint32, status [N] int32 (allocated when None).
        -> (rc, out, counts, status); nothing is raised for a capacity status."""
        a = np.ascontiguousarray(frames)
        _, fmt, stride = _image_args(a[0])
        if fmt == _ffi.AGX_LF32:
            raise AgxError(_ffi.AGX_ERR_FORMAT, "detect_batch takes L8 / L16 / RGB8 frames")
        n, h, w = a.shape[:3]
        if out is None:
            out = np.zeros((n, cap), self.TAG_DTYPE)
        if counts is None:
            counts = np.zeros(n, np.uint32)
        if status is None:
            status = np.zeros(n, np.int32)
        assert out.shape == (n, cap) and out.dtype == self.TAG_DTYPE and counts.shape == (n,) and status.shape == (n,)
        status[:] = _ffi.AGX_ERR_STATE  # (every slot is written by the call; one that is not stays an error)
        dptr = None
        if device_frames is not None:
            # the chain reads device_frames, the decode reads `frames`: they must be the same pixels
            t = device_frames
            if not (getattr(t, "is_cuda", False) and t.is_contiguous()):
                raise AgxError(_ffi.AGX_ERR_ARG, "device_frames must be a contiguous CUDA tensor")
            if t.device.index != self.device:
                raise AgxError(_ffi.AGX_ERR_ARG, "device_frames is on device %s, the detector on %d" % (t.device.index, self.device))
            if tuple(t.shape) != tuple(a.shape) or t.element_size() != a.dtype.itemsize:
                raise AgxError(_ffi.AGX_ERR_ARG, "device_frames %s / %d-byte elements differ from frames %s / %d-byte elements"
                               % (tuple(t.shape), t.element_size(), tuple(a.shape), a.dtype.itemsize))
            dptr = t.data_ptr()
        rc = self._lib.agx_detect_batch(self._h, a.ctypes.data, dptr, n, w, h, stride, stride * h, fmt,
                                        out.ctypes.data, cap, counts.ctypes.data, status.ctypes.data, n_threads)
        return rc, out, counts, status

    def detect_batch(self, frames, n_threads=0, cap=1024, device_frames=None, raise_on_overflow=True):
        """detect() over a batch: frames = numpy [N,H,W] uint8 / uint16 or [N,H,W,3] uint8 in host
        memory (optionally also resident on the GPU as the torch tensor device_frames).  The chain
        runs on the device chunk by chunk while n_threads host threads (0 = agx_host_parallelism(): the
        CPUs this process may keep busy) run the uploads and the
        board search + decode.  -> list of {tag_id: 4x2 corners}.  A frame with more than `cap` tags (or
        over the detector's saddle capacity) raises by default; raise_on_overflow=False returns
        (results, status) instead: status[i] != 0 marks such a frame (its entry is None), every other
        frame keeps its result."""
        rc, out, counts, status = self.detect_batch_raw(frames, n_threads, cap, device_frames)
        n = len(counts)
        if rc != _ffi.AGX_OK and (raise_on_overflow or rc != _ffi.AGX_ERR_CAPACITY):
            self._check(rc)
        res = [None if status[i] != 0 else {int(t["id"]): t["xy"].reshape(4, 2).copy() for t in out[i, : counts[i]]} for i in range(n)]
        return res if raise_on_overflow else (res, status)

    def detect_kornia(self, img):
        """kornia::image::Image<u8, N>: an HxWxN uint8 array, N in {1, 3} (else the reference
        panics 'Only support u8c1 and u8c3')."""
        a = np.asarray(img)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] not in (1, 3):
            raise AgxError(_ffi.AGX_ERR_FORMAT, "Only support u8c1 and u8c3")
        return self.detect(a)

    def detect_from_saddles(self, saddles, luma8, cap=4096):
        """Host tail only (board search + decode) from a SADDLE_DTYPE array and the u8 luma."""
        s = np.ascontiguousarray(saddles, SADDLE_DTYPE)
        g = np.ascontiguousarray(luma8, np.uint8)
        h, w = g.shape
        out = (_ffi.TagC * cap)()
        n = C.c_uint32(0)
        self._check(self._lib.agx_detect_from_saddles(self._h, s.ctypes.data, len(s), g.ctypes.data, w, h, w, out,
                                                      cap, C.byref(n)))
        return {int(out[i].id): np.array(out[i].xy, np.float32).reshape(4, 2) for i in range(n.value)}

    @staticmethod
    def detect_tail(tag_family, saddles, luma8, optional_detector_params=None, cap=4096, n_threads=1):
        """Host tail without a device: board search + decode from a SADDLE_DTYPE array (n_threads > 1: the
        frame's board search on several host threads, same result)."""
        fam = TagFamily.from_str(tag_family) if isinstance(tag_family, str) else TagFamily(tag_family)
        prm = optional_detector_params._c() if optional_detector_params is not None else None
        s = np.ascontiguousarray(saddles, SADDLE_DTYPE)
        g = np.ascontiguousarray(luma8, np.uint8)
        h, w = g.shape
        out = (_ffi.TagC * cap)()
        n = C.c_uint32(0)
        st = _ffi.lib().agx_detect_tail_threads(int(fam), C.byref(prm) if prm is not None else None, s.ctypes.data, len(s),
                                                g.ctypes.data, w, h, w, out, cap, C.byref(n), int(n_threads))
        if st != _ffi.AGX_OK:
            raise AgxError(st)
        return {int(out[i].id): np.array(out[i].xy, np.float32).reshape(4, 2) for i in range(n.value)}

    @staticmethod
    def luma8(img):
        a, fmt, stride = _image_args(img)
        h, w = a.shape[:2]
        out = np.empty((h, w), np.uint8)
        st = _ffi.lib().agx_luma8(a.ctypes.data, w, h, stride, fmt, out.ctypes.data)
        if st != _ffi.AGX_OK:
            raise AgxError(st)
        return out

    # ---- batches resident in device memory ----------------------------------------------
    def set_limits(self, max_candidates=0, max_clusters=0, max_saddles=0):
        self._check(self._lib.agx_detector_set_limits(self._h, max_candidates, max_clusters, max_saddles))

    def set_stream(self, hip_stream_ptr, external=True):
        """external=True: launch on the given hipStream_t (0 = HIP's default stream);
        external=False: back to the detector's own stream."""
        self._check(self._lib.agx_detector_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0), 1 if external else 0))
        self._stream_ptr = hip_stream_ptr if external else "own"

    def set_option(self, name, value):
        self._check(self._lib.agx_detector_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int(0)
        self._check(self._lib.agx_detector_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def sync(self):
        self._check(self._lib.agx_detector_sync(self._h))

    def _follow_torch_stream(self, frames=None):
        """Stream-order the chain behind whatever produced `frames`: launch on torch's current
        stream of that device (a detector otherwise uses its own non-blocking stream, which
        does not wait for work queued on torch's streams).  frames None: the detector's own device."""
        import torch
        s = torch.cuda.current_stream(frames.device if frames is not None else self.device).cuda_stream
        if getattr(self, "_stream_ptr", "own") != s:
            self.set_stream(s)

    @staticmethod
    def _tensor_format(frames):
        import torch
        if not frames.is_cuda or not frames.is_contiguous():
            raise AgxError(_ffi.AGX_ERR_ARG, "frames must be a contiguous device tensor")
        if frames.dim() == 3 and frames.dtype == torch.uint8:
            return _ffi.AGX_L8, 1
        if frames.dim() == 3 and frames.dtype in (torch.int16, getattr(torch, "uint16", torch.int16)):
            return _ffi.AGX_L16, 2
        if frames.dim() == 4 and frames.shape[3] == 3 and frames.dtype == torch.uint8:
            return _ffi.AGX_RGB8, 3
        if frames.dim() == 3 and frames.dtype == torch.float32:
            return _ffi.AGX_LF32, 4
        raise AgxError(_ffi.AGX_ERR_FORMAT, "unsupported frame tensor %s %s" % (tuple(frames.shape), frames.dtype))

    def saddles_batch_enqueue(self, frames):
        """frames: a torch tensor on this detector's GPU -- [N,H,W] uint8 (L8), [N,H,W] int16/uint16
        (L16) or [N,H,W,3] uint8 (RGB8), contiguous.  Returns immediately; see saddles_batch_fetch."""
        fmt, bpp = self._tensor_format(frames)
        self._follow_torch_stream(frames)
        n, h, w = frames.shape[:3]
        self._check(self._lib.agx_saddles_batch_enqueue(self._h, frames.data_ptr(), n, w, h, w * bpp, w * h * bpp,
                                                        fmt))
        self._batch = (n, frames)  # keep the tensor alive until fetched

    def saddles_batch_enqueue_to(self, frames, out_saddles, frame_table):
        """Device-resident results: out_saddles float32 [capacity, 5] and frame_table int32
        [n_frames, 4] (count, offset, status, clusters) are torch tensors on the same GPU."""
        fmt, bpp = self._tensor_format(frames)
        self._follow_torch_stream(frames)
        n, h, w = frames.shape[:3]
        assert out_saddles.is_cuda and out_saddles.is_contiguous() and out_saddles.shape[1] == 5
        assert frame_table.is_cuda and frame_table.is_contiguous() and tuple(frame_table.shape) == (n, 4)
        self._check(self._lib.agx_saddles_batch_enqueue_to(
            self._h, frames.data_ptr(), n, w, h, w * bpp, w * h * bpp, fmt, out_saddles.data_ptr(),
            out_saddles.shape[0], frame_table.data_ptr()))
        self._batch = None

    def saddles_batch_enqueue_ptr(self, dptr, n, w, h, row_stride, frame_stride, fmt, follow_torch_stream=True):
        """agx_saddles_batch_enqueue on a raw device address (any row / frame stride: a view cut out of a larger allocation).
        Like the tensor forms it is launched on torch's current stream of the detector's device, behind whatever torch has
        queued there to produce the frames (and torch events on that stream see the kernels); follow_torch_stream=False keeps
        the stream the detector is on (its own non-blocking one unless set_stream was called: the caller orders the work)."""
        if follow_torch_stream:
            self._follow_torch_stream()
        self._check(self._lib.agx_saddles_batch_enqueue(self._h, C.c_void_p(dptr), n, w, h, row_stride, frame_stride,
                                                        fmt))
        self._batch = (n, None)

    def saddles_batch_fetch(self, cap_per_frame=None, raise_on_overflow=True):
        """-> (list of SADDLE_DTYPE arrays, one per frame; per-frame status array).  cap_per_frame None:
        sized from the batch's longest list."""
        if self._batch is None:
            raise AgxError(_ffi.AGX_ERR_STATE, "no batch enqueued")
        n = self._batch[0]
        if cap_per_frame is None:
            counts = np.zeros(n, np.uint32)
            st = self._lib.agx_saddles_batch_fetch(self._h, None, 0, counts.ctypes.data, None)  # counts only
            if st not in (_ffi.AGX_OK, _ffi.AGX_ERR_CAPACITY):
                self._check(st)
            cap_per_frame = max(1, int(counts.max()))
        out = np.zeros((n, cap_per_frame), SADDLE_DTYPE)
        counts = np.zeros(n, np.uint32)
        status = np.zeros(n, np.int32)
        st = self._lib.agx_saddles_batch_fetch(self._h, out.ctypes.data, cap_per_frame, counts.ctypes.data,
                                               status.ctypes.data)
        if st != _ffi.AGX_OK and (raise_on_overflow or st != _ffi.AGX_ERR_CAPACITY):
            self._check(st)
        res = [out[i, : counts[i]].copy() if status[i] == 0 else out[i, :0].copy() for i in range(n)]
        return res, status

    def saddles_batch_fetch_into(self, out, counts, status):
        """agx_saddles_batch_fetch into caller-owned arrays (what a Rust / C caller does: no allocation per call): out
        SADDLE_DTYPE [n, cap], counts uint32 [n], status int32 [n].  Returns the call's status (0 or AGX_ERR_CAPACITY)."""
        if self._batch is None:
            raise AgxError(_ffi.AGX_ERR_STATE, "no batch enqueued")
        n = self._batch[0]
        assert out.dtype == SADDLE_DTYPE and out.shape[0] == n and out.flags.c_contiguous and counts.shape == (n,) and status.shape == (n,)
        st = self._lib.agx_saddles_batch_fetch(self._h, out.ctypes.data, out.shape[1], counts.ctypes.data, status.ctypes.data)
        if st not in (_ffi.AGX_OK, _ffi.AGX_ERR_CAPACITY):
            self._check(st)
        return st

    # ---- measurement / parity hooks ------------------------------------------------------
    def profile_enable(self, level=2):
        """0/False off, 1 = time the blur kernel only, 2/True = time every kernel."""
        level = 2 if level is True else (0 if level is False else int(level))
        self._check(self._lib.agx_profile_enable(self._h, level))

    def profile_reset(self):
        self._check(self._lib.agx_profile_reset(self._h))

    def profile_read(self):
        names = (C.c_char_p * _ffi.AGX_N_KERNELS)()
        ms = (C.c_double * _ffi.AGX_N_KERNELS)()
        cnt = (C.c_uint64 * _ffi.AGX_N_KERNELS)()
        self._check(self._lib.agx_profile_read(self._h, names, ms, cnt))
        return {names[i].decode(): (ms[i], int(cnt[i])) for i in range(_ffi.AGX_N_KERNELS) if names[i]}

    def constants(self):
        w = np.zeros(7, np.float32)
        cone = np.zeros(25, np.float32)
        pmat = np.zeros((25, 6), np.float32)
        self._check(self._lib.agx_detector_constants(self._h, w.ctypes.data, cone.ctypes.data, pmat.ctypes.data))
        return w, cone, pmat

    def debug_fetch(self, frame, what, shape=None):
        """Intermediate product of the last batch: 'blur', 'resp' (HxW f32; K1's in-register response,
        needs set_option("store_response", 1) before the batch), 'resp_recomputed', 'min' (f32),
        'centers' (cluster table sorted by first pixel), 'refined' (unfiltered saddles)."""
        code = {"blur": 0, "resp": 1, "min": 2, "centers": 3, "refined": 4, "counters": 5, "resp_recomputed": 6, "verify_stats": 7, "redzones": 8, "luma8": 9, "wave_times": 10}[what]
        n = C.c_size_t(0)
        if code in (0, 1, 6):
            assert shape is not None
            buf = np.empty(shape, np.float32)
        elif code == 2:
            buf = np.empty(1, np.float32)
        elif code == 5:
            buf = np.empty(8, np.uint32)
        elif code == 7:
            buf = np.empty(20, np.uint32)
        elif code == 8:
            buf = np.empty(6, np.uint32)
        elif code == 9:
            assert shape is not None
            buf = np.empty(shape, np.uint8)
        elif code == 10:  # `frame` selects the kernel (1 verify, 2 flood, 3 refine); shape = number of workgroups
            buf = np.zeros((int(shape), 2), np.uint64)
        elif code == 3:
            buf = np.empty(1 << 20, _CLUSTER_DTYPE)
        else:
            buf = np.empty(1 << 20, SADDLE_DTYPE)
        self._check(self._lib.agx_debug_fetch(self._h, frame, code, buf.ctypes.data, buf.nbytes, C.byref(n)))
        if code == 2:
            return buf[0]
        if code == 5:
            return dict(zip(["flags", "seeds", "big_seeds", "clusters", "generic_candidates", "generic_roots",
                             "refined", "saddles"], [int(v) for v in buf]))
        if code == 7:
            return buf
        if code == 8:  # AGX_REDZONE_BYTES set when the handle was created: guard bytes around the workspace buffers
            return {"buffers": int(buf[0]), "damaged_bytes": int(buf[1]), "first_buffer": int(np.int32(buf[2])),
                    "first_offset": int(buf[3:4].view(np.int32)[0]), "buffer0_address": int(buf[4]) | (int(buf[5]) << 32)}
        if code in (3, 4):
            return buf[: n.value].copy()
        return buf


class DetectorGroup:
    """Several GPUs of one node from ONE process over the C ABI's detector groups (agx_group_*):
    rank r = one TagDetector on devices[r] with its own stream; a batch shards by frame and the
    per-rank result slabs are gathered to devices[0] (transport "rccl": ncclSend / ncclRecv over
    xGMI; "peer": hipMemcpyPeerAsync -- also accepts the same device twice, for one-GPU boxes)."""

    def __init__(self, tag_family, devices, optional_detector_params=None, transport="rccl"):
        self._lib = _ffi.lib()
        self._g = C.c_void_p()
        fam = TagFamily.from_str(tag_family) if isinstance(tag_family, str) else TagFamily(tag_family)
        prm = optional_detector_params._c() if optional_detector_params is not None else None
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        tr = {"rccl": _ffi.AGX_GATHER_RCCL, "peer": _ffi.AGX_GATHER_PEER}[transport]
        st = self._lib.agx_group_create(int(fam), C.byref(prm) if prm is not None else None, devs, len(devices), tr,
                                        C.byref(self._g))
        if st != _ffi.AGX_OK:
            self._g = C.c_void_p()
            raise AgxError(st, "agx_group_create: %s" % self._lib.agx_group_last_error(None).decode())
        self.devices = list(devices)
        self._keep = None
        self._frames_per_rank = 0

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            self._lib.agx_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self._lib.agx_group_size(self._g)

    def _check(self, st):
        if st != _ffi.AGX_OK:
            raise AgxError(st, self._lib.agx_group_last_error(self._g).decode())

    def saddles_enqueue(self, frames_per_rank, records_per_frame=0):
        """frames_per_rank: one contiguous device tensor per rank ([F,H,W] u8 / int16 or [F,H,W,3] u8,
        same shape everywhere), rank r's on devices[r]; the caller has made sure they are ready
        (e.g. torch.cuda.synchronize): the ranks run on their detectors' own streams."""
        assert len(frames_per_rank) == len(self.devices)
        fmt, bpp = TagDetector._tensor_format(frames_per_rank[0])
        n, h, w = frames_per_rank[0].shape[:3]
        for t in frames_per_rank:
            assert tuple(t.shape) == tuple(frames_per_rank[0].shape) and t.is_contiguous() and t.is_cuda
        ptrs = (C.c_void_p * len(frames_per_rank))(*[t.data_ptr() for t in frames_per_rank])
        self._check(self._lib.agx_group_saddles_enqueue(self._g, ptrs, n, w, h, w * bpp, w * h * bpp, fmt,
                                                        records_per_frame))
        self._keep = list(frames_per_rank)
        self._frames_per_rank = n

    def saddles_fetch(self, cap_per_frame=2048, raise_on_overflow=True):
        """-> (list of SADDLE_DTYPE arrays, global frame r*F + f; status array)."""
        n = self._frames_per_rank * len(self.devices)
        out = np.zeros((n, cap_per_frame), SADDLE_DTYPE)
        counts = np.zeros(n, np.uint32)
        status = np.zeros(n, np.int32)
        st = self._lib.agx_group_saddles_fetch(self._g, out.ctypes.data, cap_per_frame, counts.ctypes.data,
                                               status.ctypes.data)
        if st != _ffi.AGX_OK and (raise_on_overflow or st != _ffi.AGX_ERR_CAPACITY):
            self._check(st)
        self._keep = None
        # a frame whose list is merely longer than cap_per_frame reports its length (status -3): fetch again with room
        self.last_counts = counts
        return [out[i, : counts[i]].copy() if status[i] == 0 else out[i, :0].copy() for i in range(n)], status
