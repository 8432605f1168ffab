"""ctypes declarations of include/aprilgrid_amd.h (one-to-one)."""
import ctypes as C
import os
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libaprilgrid_amd.so")

AGX_OK = 0
AGX_ERR_ARG, AGX_ERR_FORMAT, AGX_ERR_CAPACITY, AGX_ERR_HIP = -1, -2, -3, -4
AGX_ERR_NO_DEVICE, AGX_ERR_FAMILY, AGX_ERR_STATE, AGX_ERR_NOMEM = -5, -6, -7, -8
AGX_L8, AGX_L16, AGX_RGB8, AGX_LF32 = 0, 1, 2, 3
AGX_GATHER_RCCL, AGX_GATHER_PEER = 0, 1
AGX_DBG_BLUR, AGX_DBG_RESP, AGX_DBG_MIN, AGX_DBG_CENTERS, AGX_DBG_REFINED = 0, 1, 2, 3, 4
AGX_N_KERNELS = 5


class Params(C.Structure):
    _fields_ = [("tag_spacing_ratio", C.c_float), ("min_saddle_angle", C.c_float),
                ("max_saddle_angle", C.c_float), ("max_num_of_boards", C.c_uint8)]


class SaddleC(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("k", C.c_float), ("theta", C.c_float),
                ("phi", C.c_float)]


class TagC(C.Structure):
    _fields_ = [("id", C.c_uint32), ("xy", C.c_float * 8)]


# every symbol include/aprilgrid_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "agx_abi_version": (C.c_int, []),
    "agx_status_string": (C.c_char_p, [C.c_int]),
    "agx_last_error": (C.c_char_p, [_P]),
    "agx_family_from_str": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "agx_default_params": (None, [C.POINTER(Params)]),
    "agx_detector_create": (C.c_int, [C.c_int, C.POINTER(Params), C.c_int, C.POINTER(_P)]),
    "agx_detector_destroy": (None, [_P]),
    "agx_detector_family_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int)]),
    "agx_detector_set_limits": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "agx_detector_set_stream": (C.c_int, [_P, _P, C.c_int]),
    "agx_detector_sync": (C.c_int, [_P]),
    "agx_detector_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "agx_detector_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int)]),
    "agx_refined_saddle_points": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_int, _P, C.c_uint32,
                                            C.POINTER(C.c_uint32)]),
    "agx_detect": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_int, _P, C.c_uint32,
                             C.POINTER(C.c_uint32)]),
    "agx_detect_planes": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, C.c_int, C.c_int, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "agx_saddles_batch_enqueue": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int]),
    "agx_saddles_batch_fetch": (C.c_int, [_P, _P, C.c_uint32, _P, _P]),
    "agx_saddles_batch_enqueue_to": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int,
                                               _P, C.c_uint32, _P]),
    "agx_host_parallelism": (C.c_int, []),
    "agx_debug_cgroup_cpu_quota": (C.c_int, [C.c_char_p, C.c_char_p]),
    "agx_detect_batch": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, _P, C.c_uint32,
                                   _P, _P, C.c_int]),
    "agx_group_create": (C.c_int, [C.c_int, C.POINTER(Params), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(_P)]),
    "agx_group_destroy": (None, [_P]),
    "agx_group_size": (C.c_int, [_P]),
    "agx_group_detector": (_P, [_P, C.c_int]),
    "agx_group_saddles_enqueue": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int,
                                            C.c_uint32]),
    "agx_group_saddles_fetch": (C.c_int, [_P, _P, C.c_uint32, _P, _P]),
    "agx_group_last_error": (C.c_char_p, [_P]),
    "agx_detect_from_saddles": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_int, C.c_int, C.c_size_t, _P, C.c_uint32,
                                          C.POINTER(C.c_uint32)]),
    "agx_detect_tail": (C.c_int, [C.c_int, C.POINTER(Params), _P, C.c_uint32, _P, C.c_int, C.c_int, C.c_size_t, _P,
                                  C.c_uint32, C.POINTER(C.c_uint32)]),
    "agx_detect_tail_threads": (C.c_int, [C.c_int, C.POINTER(Params), _P, C.c_uint32, _P, C.c_int, C.c_int, C.c_size_t, _P,
                                          C.c_uint32, C.POINTER(C.c_uint32), C.c_int]),
    "agx_luma8": (C.c_int, [_P, C.c_int, C.c_int, C.c_size_t, C.c_int, _P]),
    "agx_debug_angle_pairs": (C.c_int, [_P, C.c_size_t, _P, _P, _P]),
    "agx_debug_angle_pairs_coarse": (C.c_int, [_P, C.c_size_t, _P, _P]),
    "agx_debug_libm_atan2f_check": (C.c_int, [C.c_uint64, C.c_uint64, _P]),
    "agx_debug_white_block_angles": (C.c_int, [_P, C.c_size_t, _P, _P]),
    "agx_profile_enable": (C.c_int, [_P, C.c_int]),
    "agx_profile_reset": (C.c_int, [_P]),
    "agx_profile_read": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "agx_debug_fetch": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "agx_detector_constants": (C.c_int, [_P, _P, _P, _P]),
}


def build_library(force=False):
    """Compile libaprilgrid_amd.so in-tree with hipcc for gfx950 (make -C <package dir>)."""
    cmd = ["make", "-C", _PKG_DIR, "-s"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "aprilgrid-rs_amd: %s is missing. Build it with `make -C %s` (hipcc, gfx950) or "
                "__graft_entry__.build(). There is no CPU fallback." % (LIB_PATH, _PKG_DIR))
        # One HIP runtime per process: torch bundles its own libamdhip64 / libhsa-runtime64.  If
        # this library pulled in /opt/rocm's copy first and torch loaded its own afterwards, the
        # second runtime would find "no ROCm-capable device".  Import torch first when it is
        # installed so that libamdhip64.so.7 resolves to the copy already in the process.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # AGX_LIBRARY: another build of the same library (kernel A/B measurements on one GPU box)
        l = C.CDLL(os.environ.get("AGX_LIBRARY") or LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(l, name)  # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib
