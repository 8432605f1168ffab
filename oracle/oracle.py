"""ctypes binding of oracle/liborc.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import json
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liborc.so"

FMT_L8, FMT_L16, FMT_RGB8, FMT_LF32 = 0, 1, 2, 3


class Saddle(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("k", C.c_float), ("theta", C.c_float),
                ("phi", C.c_float)]


class Tag(C.Structure):
    _fields_ = [("id", C.c_uint32), ("xy", C.c_float * 8)]


class Params(C.Structure):
    _fields_ = [("tag_spacing_ratio", C.c_float), ("min_saddle_angle", C.c_float),
                ("max_saddle_angle", C.c_float), ("max_num_of_boards", C.c_int)]


class Debug(C.Structure):
    _fields_ = [("blur", C.c_void_p), ("resp", C.c_void_p), ("min_resp", C.c_void_p),
                ("n_clusters", C.c_void_p), ("centers", C.c_void_p), ("first_index", C.c_void_p),
                ("sizes", C.c_void_p), ("cap_clusters", C.c_int), ("n_refined", C.c_void_p),
                ("refined", C.c_void_p)]


SADDLE_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("k", "f4"), ("theta", "f4"), ("phi", "f4")])

# TagDetector::new family table, src/detector.rs:369-405: (edge, border, hamming, table)
FAMILIES = {
    "T16H5": (4, 2, 1, "T16H5"),
    "T25H7": (5, 2, 2, "T25H7"),
    "T25H9": (5, 2, 2, "T25H9"),
    "T36H11": (6, 2, 3, "T36H11"),
    "T36H11B1": (6, 1, 3, "T36H11"),
}


def build(force=False):
    src = _HERE / "agx_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_min_response.restype = C.c_float
        _lib.orc_theta_distance_degree.restype = C.c_float
        _lib.orc_theta_distance_degree.argtypes = [C.c_float] * 2
        for f in (_lib.orc_cross, _lib.orc_dot, _lib.orc_angle_degree):
            f.restype = C.c_float
            f.argtypes = [C.c_float] * 4
        _lib.orc_find_xy.argtypes = [C.c_float] * 6 + [C.c_void_p] * 2
        _lib.orc_rotate_bits.restype = C.c_uint64
        _lib.orc_rotate_bits.argtypes = [C.c_uint64, C.c_int]
        _lib.orc_gaussian_blur_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        _lib.orc_blur_weights.argtypes = [C.c_float, C.c_void_p, C.c_int]
        _lib.orc_pixel_bfs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.c_void_p, C.c_int]
        _lib.orc_cluster_centers.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_tag_affine.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        _lib.orc_refined_saddle_points.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int]
        _lib.orc_detect_tail.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                         C.c_int]
        _lib.orc_rochade_refine.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                            C.c_int, C.c_void_p]
        _lib.orc_luma_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p]
        _lib.orc_luma_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p]
        _lib.orc_hessian_response.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_min_response.argtypes = [C.c_void_p, C.c_size_t]
        _lib.orc_refine_pmat.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    return _lib


_codes_cache = None


def family_codes(name):
    """Code words from the committed data fixture tests/golden/tag_families.json."""
    global _codes_cache
    if _codes_cache is None:
        p = _HERE.parent / "tests" / "golden" / "tag_families.json"
        _codes_cache = json.loads(p.read_text())
    return np.asarray(_codes_cache[FAMILIES[name][3]], dtype=np.uint64)


def image_fmt(img):
    """(array, fmt, stride_bytes) for a numpy image: HxW u8, HxW u16 or HxWx3 u8."""
    a = np.ascontiguousarray(img)
    if a.ndim == 2 and a.dtype == np.uint8:
        return a, FMT_L8, a.shape[1]
    if a.ndim == 2 and a.dtype == np.uint16:
        return a, FMT_L16, a.shape[1] * 2
    if a.ndim == 3 and a.shape[2] == 3 and a.dtype == np.uint8:
        return a, FMT_RGB8, a.shape[1] * 3
    if a.ndim == 2 and a.dtype == np.float32:
        return a, FMT_LF32, a.shape[1] * 4
    raise ValueError("unsupported image: shape %s dtype %s" % (a.shape, a.dtype))


def default_params():
    return Params(0.3, 30.0, 60.0, 2)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def luma_f32(img):
    a, fmt, stride = image_fmt(img)
    h, w = a.shape[:2]
    out = np.empty((h, w), np.float32)
    assert lib().orc_luma_f32(_ptr(a), w, h, stride, fmt, _ptr(out)) == 0
    return out


def luma_u8(img):
    a, fmt, stride = image_fmt(img)
    h, w = a.shape[:2]
    out = np.empty((h, w), np.uint8)
    assert lib().orc_luma_u8(_ptr(a), w, h, stride, fmt, _ptr(out)) == 0
    return out


def blur_weights(sigma=1.5):
    w = np.zeros(64, np.float32)
    r = lib().orc_blur_weights(sigma, _ptr(w), 64)
    return w[: 2 * r + 1].copy()


def gaussian_blur_f32(img_f32, sigma=1.5):
    a = np.ascontiguousarray(img_f32, np.float32)
    out = np.empty_like(a)
    lib().orc_gaussian_blur_f32(_ptr(a), a.shape[1], a.shape[0], sigma, _ptr(out))
    return out


def hessian_response(img_f32):
    a = np.ascontiguousarray(img_f32, np.float32)
    out = np.empty_like(a)
    lib().orc_hessian_response(_ptr(a), a.shape[1], a.shape[0], _ptr(out))
    return out


def refine_constants(half=2):
    n = (2 * half + 1) ** 2
    p = np.zeros((n, 6), np.float32)
    k = np.zeros(n, np.float32)
    lib().orc_refine_pmat(half, _ptr(p), _ptr(k))
    return p, k


def refined_saddle_points(img, params=None, debug=False, cap=1 << 16):
    """TagDetector::refined_saddle_points.  Returns a structured array of saddles; with
    debug=True also a dict of the intermediate products."""
    a, fmt, stride = image_fmt(img)
    h, w = a.shape[:2]
    prm = params or default_params()
    out = np.zeros(cap, SADDLE_DTYPE)
    dbg_ptr = None
    d = {}
    if debug:
        capc = max(16, (w * h) // 8)
        d = dict(blur=np.empty((h, w), np.float32), resp=np.empty((h, w), np.float32),
                 min_resp=np.zeros(1, np.float32), n_clusters=np.zeros(1, np.int32),
                 centers=np.zeros((capc, 2), np.float32), first_index=np.zeros(capc, np.uint32),
                 sizes=np.zeros(capc, np.uint32), n_refined=np.zeros(1, np.int32),
                 refined=np.zeros(capc, SADDLE_DTYPE))
        dbg = Debug(_ptr(d["blur"]), _ptr(d["resp"]), _ptr(d["min_resp"]), _ptr(d["n_clusters"]),
                    _ptr(d["centers"]), _ptr(d["first_index"]), _ptr(d["sizes"]), capc,
                    _ptr(d["n_refined"]), _ptr(d["refined"]))
        dbg_ptr = C.addressof(dbg)
    n = lib().orc_refined_saddle_points(_ptr(a), w, h, stride, fmt, C.addressof(prm), _ptr(out), cap,
                                        dbg_ptr)
    if n < 0:
        raise RuntimeError("orc_refined_saddle_points failed: %d" % n)
    res = out[: min(n, cap)].copy()
    if debug:
        nc = int(d["n_clusters"][0])
        nr = int(d["n_refined"][0])
        d["centers"] = d["centers"][:nc]
        d["first_index"] = d["first_index"][:nc]
        d["sizes"] = d["sizes"][:nc]
        d["refined"] = d["refined"][:nr]
        d["min_resp"] = d["min_resp"][0]
        return res, d
    return res


def detect(img, family="T36H11", params=None, cap=1024):
    """TagDetector::detect.  Returns {tag_id: 4x2 float32 corners}."""
    a, fmt, stride = image_fmt(img)
    h, w = a.shape[:2]
    prm = params or default_params()
    edge, border, hamming, _ = FAMILIES[family]
    codes = family_codes(family)
    out = (Tag * cap)()
    n = lib().orc_detect(_ptr(a), w, h, stride, fmt, C.addressof(prm), border, edge, hamming,
                         _ptr(codes), len(codes), out, cap)
    if n < 0:
        raise RuntimeError("orc_detect failed: %d" % n)
    return {int(out[i].id): np.array(out[i].xy, np.float32).reshape(4, 2) for i in range(min(n, cap))}


def detect_tail(grey_u8, saddles, family="T36H11", params=None, cap=1024):
    """Host tail of detect() from a saddle list (structured array) and the u8 luma plane."""
    g = np.ascontiguousarray(grey_u8, np.uint8)
    h, w = g.shape
    prm = params or default_params()
    edge, border, hamming, _ = FAMILIES[family]
    codes = family_codes(family)
    s = np.ascontiguousarray(saddles, SADDLE_DTYPE).copy()
    out = (Tag * cap)()
    n = lib().orc_detect_tail(_ptr(g), w, h, _ptr(s), len(s), border, edge, hamming, _ptr(codes),
                              len(codes), prm.max_num_of_boards, out, cap)
    return {int(out[i].id): np.array(out[i].xy, np.float32).reshape(4, 2) for i in range(min(n, cap))}
