"""aprilgrid-rs_amd -- MI355X-native AprilGrid saddle/tag detection path.

Host-side mirror of the Rust crate `aprilgrid` 0.8.0 (`aprilgrid::detector::TagDetector`,
`aprilgrid::TagFamily`, `DetectorParams`, `Saddle`) over the C ABI of
include/aprilgrid_amd.h.  All compute happens in libaprilgrid_amd.so (hand-written HIP for
gfx950 plus the C++ host tail); importing this package fails loudly if that library has not
been built -- there is no CPU fallback.
"""
from .detector import (DetectorGroup, DetectorParams, Saddle, TagDetector, TagFamily, AgxError, SADDLE_DTYPE,
                       build_library, library_path)

__all__ = ["DetectorGroup", "DetectorParams", "Saddle", "TagDetector", "TagFamily", "AgxError", "SADDLE_DTYPE",
           "build_library", "library_path"]
