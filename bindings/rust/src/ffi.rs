//! include/aprilgrid_amd.h, declaration for declaration.  Nothing here is written twice by hand without a check:
//! tests/test_rust_binding.py (CPU suite of the repository) parses this file and the header and compares every function's
//! name, argument count and argument / return widths, every `#[repr(C)]` struct's fields, and every constant.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

/// `agx_detector` / `agx_group`: opaque handles.
#[repr(C)]
pub struct agx_detector {
    _private: [u8; 0],
}
#[repr(C)]
pub struct agx_group {
    _private: [u8; 0],
}

/// `agx_params` = DetectorParams (reference src/detector.rs:25-30).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct agx_params {
    pub tag_spacing_ratio: c_float,
    pub min_saddle_angle: c_float,
    pub max_saddle_angle: c_float,
    pub max_num_of_boards: u8,
}
/// `agx_saddle` = Saddle (reference src/saddle.rs:3-9), 5 x f32.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct agx_saddle {
    pub x: c_float,
    pub y: c_float,
    pub k: c_float,
    pub theta: c_float,
    pub phi: c_float,
}
/// `agx_tag`: one entry of detect()'s map (reference src/detector.rs:520): id and four corners.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct agx_tag {
    pub id: u32,
    pub xy: [c_float; 8],
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct agx_frame_result {
    pub count: u32,
    pub offset: u32,
    pub status: u32,
    pub n_clusters: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct agx_cluster_info {
    pub first_index: u32,
    pub size: u32,
    pub cx: c_float,
    pub cy: c_float,
}

pub const AGX_ABI_VERSION: c_int = 1;
pub const AGX_N_KERNELS: c_int = 5;

// enum agx_status
pub const AGX_OK: c_int = 0;
pub const AGX_ERR_ARG: c_int = -1;
pub const AGX_ERR_FORMAT: c_int = -2;
pub const AGX_ERR_CAPACITY: c_int = -3;
pub const AGX_ERR_HIP: c_int = -4;
pub const AGX_ERR_NO_DEVICE: c_int = -5;
pub const AGX_ERR_FAMILY: c_int = -6;
pub const AGX_ERR_STATE: c_int = -7;
pub const AGX_ERR_NOMEM: c_int = -8;
// enum agx_family: the order of `enum TagFamily` (reference src/tag_families.rs:5-13)
pub const AGX_T16H5: c_int = 0;
pub const AGX_T25H7: c_int = 1;
pub const AGX_T25H9: c_int = 2;
pub const AGX_T36H11: c_int = 3;
pub const AGX_T36H11B1: c_int = 4;
// enum agx_format
pub const AGX_L8: c_int = 0;
pub const AGX_L16: c_int = 1;
pub const AGX_RGB8: c_int = 2;
pub const AGX_LF32: c_int = 3;
// per-frame status bits of the device frame table
pub const AGX_FRAME_CANDIDATE_OVERFLOW: c_int = 1;
pub const AGX_FRAME_CLUSTER_OVERFLOW: c_int = 2;
pub const AGX_FRAME_SADDLE_OVERFLOW: c_int = 4;
pub const AGX_FRAME_CENTROID_INEXACT: c_int = 8;
pub const AGX_FRAME_GENERIC_PATH: c_int = 16;
pub const AGX_FRAME_DENSE_THRESHOLD: c_int = 32;
pub const AGX_FRAME_LARGE_RESULT: c_int = 64;
// transports of a detector group
pub const AGX_GATHER_RCCL: c_int = 0;
pub const AGX_GATHER_PEER: c_int = 1;
// agx_debug_fetch items
pub const AGX_DBG_BLUR: c_int = 0;
pub const AGX_DBG_RESP: c_int = 1;
pub const AGX_DBG_MIN: c_int = 2;
pub const AGX_DBG_CENTERS: c_int = 3;
pub const AGX_DBG_REFINED: c_int = 4;
pub const AGX_DBG_COUNTERS: c_int = 5;
pub const AGX_DBG_RESP_RECOMPUTED: c_int = 6;
pub const AGX_DBG_VERIFY_STATS: c_int = 7;
pub const AGX_DBG_REDZONES: c_int = 8;
pub const AGX_DBG_LUMA8: c_int = 9;
pub const AGX_DBG_WAVE_TIMES: c_int = 10;
pub const AGX_DBG_TAIL_TABLE_ADDR: c_int = 11;

extern "C" {
    pub fn agx_abi_version() -> c_int;
    pub fn agx_status_string(status: c_int) -> *const c_char;
    pub fn agx_last_error(det: *const agx_detector) -> *const c_char;
    pub fn agx_family_from_str(name: *const c_char, family_out: *mut c_int) -> c_int;
    pub fn agx_default_params(out: *mut agx_params);

    pub fn agx_detector_create(family: c_int, params: *const agx_params, device: c_int, out: *mut *mut agx_detector) -> c_int;
    pub fn agx_detector_destroy(det: *mut agx_detector);
    pub fn agx_detector_family_info(det: *const agx_detector, edge_bits: *mut c_int, border_bits: *mut c_int,
                                    hamming_distance: *mut c_int, codes: *mut *const u64, n_codes: *mut c_int) -> c_int;
    pub fn agx_detector_set_limits(det: *mut agx_detector, max_candidates: u32, max_clusters: u32, max_saddles: u32) -> c_int;
    pub fn agx_detector_set_option(det: *mut agx_detector, name: *const c_char, value: c_int) -> c_int;
    pub fn agx_detector_get_option(det: *const agx_detector, name: *const c_char, value: *mut c_int) -> c_int;
    pub fn agx_detector_set_stream(det: *mut agx_detector, hip_stream: *mut c_void, external: c_int) -> c_int;
    pub fn agx_detector_sync(det: *mut agx_detector) -> c_int;
    pub fn agx_detector_constants(det: *const agx_detector, blur_w7: *mut c_float, cone25: *mut c_float, pmat150: *mut c_float) -> c_int;

    pub fn agx_refined_saddle_points(det: *mut agx_detector, pixels: *const c_void, width: c_int, height: c_int,
                                     row_stride_bytes: usize, format: c_int, out: *mut agx_saddle, cap: u32, n_out: *mut u32) -> c_int;
    pub fn agx_detect(det: *mut agx_detector, pixels: *const c_void, width: c_int, height: c_int, row_stride_bytes: usize,
                      format: c_int, out: *mut agx_tag, cap: u32, n_out: *mut u32) -> c_int;
    pub fn agx_detect_planes(det: *mut agx_detector, luma32f: *const c_float, stride32f_bytes: usize, luma8: *const u8,
                             stride8_bytes: usize, width: c_int, height: c_int, out: *mut agx_tag, cap: u32, n_out: *mut u32) -> c_int;
    pub fn agx_detect_batch(det: *mut agx_detector, frames: *const c_void, d_frames: *const c_void, n_frames: c_int, width: c_int,
                            height: c_int, row_stride_bytes: usize, frame_stride_bytes: usize, format: c_int, out: *mut agx_tag,
                            cap_per_frame: u32, counts: *mut u32, frame_status: *mut c_int, n_threads: c_int) -> c_int;
    pub fn agx_host_parallelism() -> c_int;
    pub fn agx_luma8(pixels: *const c_void, width: c_int, height: c_int, row_stride_bytes: usize, format: c_int, out: *mut u8) -> c_int;

    pub fn agx_saddles_batch_enqueue(det: *mut agx_detector, d_frames: *const c_void, n_frames: c_int, width: c_int, height: c_int,
                                     row_stride_bytes: usize, frame_stride_bytes: usize, format: c_int) -> c_int;
    pub fn agx_saddles_batch_enqueue_to(det: *mut agx_detector, d_frames: *const c_void, n_frames: c_int, width: c_int, height: c_int,
                                        row_stride_bytes: usize, frame_stride_bytes: usize, format: c_int, d_saddles: *mut c_void,
                                        saddle_capacity: u32, d_frame_table: *mut c_void) -> c_int;
    pub fn agx_saddles_batch_fetch(det: *mut agx_detector, out: *mut agx_saddle, cap_per_frame: u32, counts: *mut u32,
                                   frame_status: *mut c_int) -> c_int;

    pub fn agx_detect_from_saddles(det: *const agx_detector, saddles: *const agx_saddle, n_saddles: u32, luma8: *const u8, width: c_int,
                                   height: c_int, row_stride_bytes: usize, out: *mut agx_tag, cap: u32, n_out: *mut u32) -> c_int;
    pub fn agx_detect_tail(family: c_int, params: *const agx_params, saddles: *const agx_saddle, n_saddles: u32, luma8: *const u8,
                           width: c_int, height: c_int, row_stride_bytes: usize, out: *mut agx_tag, cap: u32, n_out: *mut u32) -> c_int;
    pub fn agx_detect_tail_threads(family: c_int, params: *const agx_params, saddles: *const agx_saddle, n_saddles: u32,
                                   luma8: *const u8, width: c_int, height: c_int, row_stride_bytes: usize, out: *mut agx_tag, cap: u32,
                                   n_out: *mut u32, n_threads: c_int) -> c_int;

    pub fn agx_group_create(family: c_int, params: *const agx_params, devices: *const c_int, n_devices: c_int, transport: c_int,
                            out: *mut *mut agx_group) -> c_int;
    pub fn agx_group_destroy(group: *mut agx_group);
    pub fn agx_group_size(group: *const agx_group) -> c_int;
    pub fn agx_group_detector(group: *mut agx_group, rank: c_int) -> *mut agx_detector;
    pub fn agx_group_saddles_enqueue(group: *mut agx_group, d_frames: *const *const c_void, frames_per_rank: c_int, width: c_int,
                                     height: c_int, row_stride_bytes: usize, frame_stride_bytes: usize, format: c_int,
                                     records_per_frame: u32) -> c_int;
    pub fn agx_group_saddles_fetch(group: *mut agx_group, out: *mut agx_saddle, cap_per_frame: u32, counts: *mut u32,
                                   frame_status: *mut c_int) -> c_int;
    pub fn agx_group_last_error(group: *const agx_group) -> *const c_char;

    pub fn agx_profile_enable(det: *mut agx_detector, on: c_int) -> c_int;
    pub fn agx_profile_reset(det: *mut agx_detector) -> c_int;
    pub fn agx_profile_read(det: *mut agx_detector, names: *mut *const c_char, ms_total: *mut c_double, launches: *mut u64) -> c_int;
    pub fn agx_debug_fetch(det: *mut agx_detector, frame: c_int, what: c_int, host_out: *mut c_void, cap_bytes: usize,
                           n_items: *mut usize) -> c_int;
    pub fn agx_debug_angle_pairs(vectors: *const c_float, n: usize, exact: *mut c_float, approx: *mut c_float, has_approx: *mut u8) -> c_int;
    pub fn agx_debug_angle_pairs_coarse(vectors: *const c_float, n: usize, coarse: *mut c_float, has_coarse: *mut u8) -> c_int;
    pub fn agx_debug_libm_atan2f_check(n: u64, seed: u64, mismatches: *mut u64) -> c_int;
    pub fn agx_debug_white_block_angles(triples: *const c_float, n: usize, reference: *mut c_float, binary64: *mut c_double) -> c_int;
    pub fn agx_debug_cgroup_cpu_quota(cgroup_root: *const c_char, proc_self_cgroup: *const c_char) -> c_int;
}
