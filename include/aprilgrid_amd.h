/*
 * aprilgrid_amd.h -- C ABI of the MI355X-native AprilGrid saddle/tag detection path.
 *
 * This is the drop-in boundary for the Rust crate `aprilgrid` 0.8.0: every entry point
 * below names the reference item (file:line under the crate root) it replaces, and a Rust
 * shim binds them 1:1.  Plain pointers and sizes only -- no C++, torch or
 * HIP types cross this line (a HIP stream is passed as an opaque void*).
 *
 * Conventions
 *   - every function returns AGX_OK (0) or a negative agx_status; nothing unwinds across
 *     the boundary: every entry point catches (std::bad_alloc, a failed std::thread ->
 *     AGX_ERR_NOMEM; tests/test_abi_cpu.py forces both) (the reference's panics -- detector.rs:500 "Only support u8c1 and u8c3",
 *     the height()-1 underflow at detector.rs:174 -- become AGX_ERR_FORMAT / AGX_ERR_ARG);
 *   - the caller owns every buffer it passes; results are copied into caller memory;
 *   - "nothing found" is AGX_OK with a zero count (reference: empty Vec / HashMap);
 *   - a detector handle owns one device, one stream and its scratch planes: it is NOT
 *     re-entrant.  The reference's `&self` methods are callable from many threads; a shim
 *     keeps that property by pooling handles (one per calling thread);
 *   - results never come from a CPU fallback: if the HIP runtime or a gfx950 device is
 *     missing, agx_detector_create fails with AGX_ERR_NO_DEVICE.
 */
#ifndef APRILGRID_AMD_H
#define APRILGRID_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_ABI_VERSION 1

typedef enum agx_status {
    AGX_OK = 0,
    AGX_ERR_ARG = -1,       /* null pointer, zero size, w or h < 2, bad stride            */
    AGX_ERR_FORMAT = -2,    /* pixel format / channel count the reference panics on; agx_detect /
                               agx_luma8 on AGX_LF32 (no u8 luma to derive: agx_detect_planes)     */
    AGX_ERR_CAPACITY = -3,  /* an output or internal list overflowed; nothing is truncated
                               silently -- raise the capacity (agx_detector_set_limits)    */
    AGX_ERR_HIP = -4,       /* a HIP runtime call failed; see agx_last_error              */
    AGX_ERR_NO_DEVICE = -5, /* no usable gfx950 device                                    */
    AGX_ERR_FAMILY = -6,    /* unknown tag family (TagFamily::from_str Err)               */
    AGX_ERR_STATE = -7,     /* call sequence error (e.g. fetch before enqueue); an unexpected internal error */
    AGX_ERR_NOMEM = -8      /* a host allocation or the creation of a worker thread failed: the WHOLE call
                               failed, no per-frame result of it is valid                  */
} agx_status;

/* tag_families::TagFamily -- src/tag_families.rs:5-13 */
typedef enum agx_family {
    AGX_T16H5 = 0,
    AGX_T25H7 = 1,
    AGX_T25H9 = 2,
    AGX_T36H11 = 3,
    AGX_T36H11B1 = 4 /* 1-bit border */
} agx_family;

/* The three DynamicImage variants the reference's tests, benches and detect_kornia feed the
 * path (src/detector.rs:409,507,478-503): ImageLuma8, ImageLuma16, ImageRgb8 (HWC) -- their luma
 * conversion (image crate to_luma32f) is fused into the blur kernel -- and, for every other
 * variant (La8, Rgba8, Rgb16, Rgba16, Rgb32F ...), the caller's own img.to_luma32f() plane. */
typedef enum agx_format {
    AGX_L8 = 0,   /* 1 byte / pixel                                    */
    AGX_L16 = 1,  /* 2 bytes / pixel, native endian                    */
    AGX_RGB8 = 2, /* 3 bytes / pixel, interleaved R,G,B (kornia Image<u8,3>) */
    AGX_LF32 = 3  /* 4 bytes / pixel: row-major f32 luma in [0,1] = DynamicImage::to_luma32f
                     (src/detector.rs:409) as the caller computed it; taken as is */
} agx_format;

/* detector::DetectorParams -- src/detector.rs:25-41 */
typedef struct agx_params {
    float tag_spacing_ratio; /* kept for layout parity; unused by the reference too (:621) */
    float min_saddle_angle;
    float max_saddle_angle;
    uint8_t max_num_of_boards;
} agx_params;

/* saddle::Saddle -- src/saddle.rs:3-9 ; repr(C) {p.0, p.1, k, theta, phi}, 20 bytes */
typedef struct agx_saddle {
    float x, y, k, theta, phi;
} agx_saddle;

/* one entry of detect()'s HashMap<u32, [(f32,f32);4]> -- src/detector.rs:505,520 */
typedef struct agx_tag {
    uint32_t id;
    float xy[8]; /* 4 corners (x,y), in the order the reference returns (:467-470) */
} agx_tag;

typedef struct agx_detector agx_detector; /* opaque; mirrors detector::TagDetector */

/* TagFamily::from_str -- src/tag_families.rs:15-28.  Accepts "t36h11"/"T36H11" etc. */
int agx_family_from_str(const char *name, int *family_out);

/* DetectorParams::default_params -- src/detector.rs:33-40 : (0.3, 30, 60, 2) */
void agx_default_params(agx_params *out);

/* TagDetector::new(tag_family, optional_detector_params) -- src/detector.rs:364-406.
 * params may be NULL (== None).  device = HIP device ordinal.  The constants the
 * reference recomputes on every call (blur weights image_util.rs:111-124, cone kernel and
 * pseudo-inverse detector.rs:208-254) are computed once here. */
int agx_detector_create(int family, const agx_params *params, int device, agx_detector **out);
void agx_detector_destroy(agx_detector *det);

/* Family facts of TagDetector::new's table (edge, border, hamming_distance, code_list). */
int agx_detector_family_info(const agx_detector *det, int *edge_bits, int *border_bits,
                             int *hamming_distance, const uint64_t **codes, int *n_codes);

/* Internal list capacities per frame (0 = keep default).  Defaults scale with the frame:
 * candidates W*H/2, clusters W*H/8, saddles W*H/64 (lists of up to 16384 saddles are ordered in
 * LDS, longer ones -- pure-noise frames of several megapixels -- in global memory).
 * Overflow of any of them
 * is reported as AGX_ERR_CAPACITY for that frame, never truncated. */
int agx_detector_set_limits(agx_detector *det, uint32_t max_candidates, uint32_t max_clusters,
                            uint32_t max_saddles);

/* Tuning / test switches (integers by name):
 *   "force_generic"        1 = cluster every frame with the generic union-find kernels instead
 *                          of the windowed flood fill (results are identical; test hook)
 *   "k1_rows_per_segment"  rows one wave of the blur kernel walks (0 = automatic)
 *   "sparse_path"          how a batch's sparse stages (verify, clusters, refinement, emission) are launched: 0 = by batch size
 *                          (default), 1 = three batch-wide launches, 2 = one 1024-thread workgroup per frame for all of it,
 *                          3 = the verify launch, then one workgroup per frame for the rest (what batches that fill the chip
 *                          take by themselves).  Results are identical on every path (tests/test_gpu_sparse_paths.py)
 *   "store_response"       1 = the blur kernel also stores the Hessian response it evaluates in
 *                          registers (parity tests: agx_debug_fetch AGX_DBG_RESP); slower
 *   "profile_stride"       with agx_profile_enable(det, 1): time the blur kernel of every n-th batch
 *                          only (an event pair costs the stream two ~5 us gaps around the kernel)
 *   "profile_kernel"       with agx_profile_enable(det, 1): which of the AGX_N_KERNELS launches is timed (0 = the blur
 *                          kernel, the default).  One kernel at a time keeps its neighbours back to back on the
 *                          stream, so the figure is the kernel's own duration; level 2 (all at once) opens a gap
 *                          in front of every kernel
 *   "reload_tuning_env"    (any value) the AGX_* measurement overrides of the environment (AGX_K1_ROWS, AGX_SPARSE_PATH, ...:
 *                          README.md "Tuning switches") are read once per process and kept; this forgets them, so that a
 *                          tool which changes its environment between runs of one process sees the change
 *   "tail_threads"         n > 1: agx_detect / agx_detect_planes / agx_detect_from_saddles search the boards of
 *                          ONE frame on n host threads (the up-to-30 seed saddles of try_find_best_board,
 *                          detector.rs:611-625, in waves of n, merged in the reference's order: same result;
 *                          latency of a single detect, at the price of host cores).  Default 1, like the
 *                          reference; agx_detect_batch parallelises over frames instead
 *   "device_tail"          agx_detect_batch's board search + decode (detector.rs:510-539) on the device, behind the chain
 *                          (csrc/tail_kernels.hip), instead of on the pool of host threads.  -1 (default): for calls of at least four
 *                          frames per host thread (below that the host tail is the faster one), where this process's atan2f is
 *                          glibc's own routine -- the kernel evaluates angle_degree (math_util.rs:31-33) by that routine,
 *                          restated --, else off; 1: on for every call, AGX_ERR_STATE where libm differs; 0: off.
 *                          The tags are the host tail's bit for bit either way: a frame the kernel cannot decide (an angle
 *                          within 1e-4 degrees of the white-block thresholds, saddle.rs:26-38) or hold (more than 1024 saddles,
 *                          128 board cells ...) is handed to the host tail.  Read back: "last_device_tail_frames",
 *                          "last_device_tail_fallbacks", "last_device_tail_uncertain" of the last agx_detect_batch call
 *   "tail_debug_band"      test hook of the hand-back path: n > 0 makes the device tail treat a white-block angle within n / 1000
 *                          degrees of 60 / 120 as undecided (the kernel's own band is 1e-4 degrees, which the suite's frames never
 *                          hit), so that a known share of frames takes the host tail; results unchanged
 *                          (tests/test_gpu_device_tail.py).  0 (default) = off
 *   "debug_ablation"       measurement switches of the kernels (tools/): bits 1 .. 1024 remove parts of the blur kernel's
 *                          work -- timing experiments, results are INVALID; bits 128 / 2048 / 8192 / 16384 collect
 *                          statistics and phase times (AGX_DBG_VERIFY_STATS), 4096 the start and end of every wave of
 *                          the sparse kernels (AGX_DBG_WAVE_TIMES), 32768 selects the blur kernel's former ascending
 *                          segment order, 65536 its former refresh interval, 131072 / 262144 stage and phase times of the
 *                          workgroup-per-frame kernel (tools/sparse_frame_phases.py, tools/flood_phases.py): results unchanged
 *                          (tests/test_gpu_parity.py) */
int agx_detector_set_option(agx_detector *det, const char *name, int value);
/* Read an option back; additionally the tiling the blur kernel used for the last enqueued batch:
 * "k1_rows_per_segment" (effective value), "k1_segments", "k1_strips", "k1_strip_columns", and
 * "last_sparse_path": how the last batch's sparse stages ran: 1 = the three launches, 2 = one workgroup per frame for all
 * of it, 3 = the verify launch, then a workgroup per frame (the values of option "sparse_path"). */
int agx_detector_get_option(const agx_detector *det, const char *name, int *value);

/* Stream selection.  external != 0: launch on the caller's stream `hip_stream` (hipStream_t as
 * void*; NULL is HIP's legacy default stream) so that the chain is stream-ordered behind the
 * producer of the frames and in front of consumers of the results (e.g. an RCCL gather), and
 * events the caller records bracket the kernels.  external == 0: back to the detector's own
 * non-blocking stream (hip_stream ignored). */
int agx_detector_set_stream(agx_detector *det, void *hip_stream, int external);

/* ---- single frame, host memory in / host memory out ---------------------------------- */

/* TagDetector::refined_saddle_points(&self, img) -> Vec<Saddle> -- src/detector.rs:408-446.
 * pixels: host pointer, row_stride_bytes between rows.  Writes min(*n_out, cap) saddles in
 * the reference's order (clusters by ascending first pixel, raster order); returns
 * AGX_ERR_CAPACITY (with *n_out = required) if cap is too small. */
int agx_refined_saddle_points(agx_detector *det, const void *pixels, int width, int height,
                              size_t row_stride_bytes, int format, agx_saddle *out, uint32_t cap,
                              uint32_t *n_out);

/* TagDetector::detect(&self, img) -> HashMap<u32,[(f32,f32);4]> -- src/detector.rs:505-540,
 * and detect_kornia (:478-503) through format = AGX_L8 / AGX_RGB8.  One entry per distinct
 * id (a later quad with the same id replaces the earlier one, as HashMap::insert). */
int agx_detect(agx_detector *det, const void *pixels, int width, int height,
               size_t row_stride_bytes, int format, agx_tag *out, uint32_t cap, uint32_t *n_out);

/* detect() for ANY DynamicImage variant: the caller hands over the two planes the reference
 * derives from the image itself -- img.to_luma32f() for the saddle chain (src/detector.rs:409) and
 * img.to_luma8() for the code decode (:507, :518) -- so the result does not depend on this
 * library restating the image crate's conversions for that variant. */
int agx_detect_planes(agx_detector *det, const float *luma32f, size_t stride32f_bytes, const uint8_t *luma8,
                      size_t stride8_bytes, int width, int height, agx_tag *out, uint32_t cap, uint32_t *n_out);

/* ---- batches of equally sized frames resident in device memory (HBM) ------------------ */

/* Enqueue the whole saddle chain (luma -> blur -> Hessian response -> min/threshold ->
 * clustering -> rochade_refine -> k/phi filter) for n_frames frames on the detector's
 * stream and return without waiting.  d_frames: DEVICE pointer; frame i starts at
 * d_frames + i*frame_stride_bytes.  Any row stride that covers a row is accepted (16-bit pixels
 * 2-byte aligned); rows and frames aligned to 4 bytes with width % 4 == 0 take the fast path.
 * HIP graphs: the call may be made while the detector's stream is being captured (after one eager
 * batch of the same geometry, so that the workspace exists); every captured batch then clears its
 * own counters, so a graph of any number of batches replays correctly any number of times. */
int agx_saddles_batch_enqueue(agx_detector *det, const void *d_frames, int n_frames, int width,
                              int height, size_t row_stride_bytes, size_t frame_stride_bytes,
                              int format);

/* Wait for the enqueued batch and copy the results out.  out: n_frames * cap_per_frame
 * saddles (frame i at out + i*cap_per_frame); counts[i] = saddles of frame i;
 * frame_status[i] (may be NULL) = AGX_OK or AGX_ERR_CAPACITY.  Returns the first non-OK
 * frame status, else AGX_OK. */
int agx_saddles_batch_fetch(agx_detector *det, agx_saddle *out, uint32_t cap_per_frame,
                            uint32_t *counts, int *frame_status);

/* Same chain, results left in CALLER-OWNED DEVICE memory (nothing is copied to the host), so
 * that they can be consumed or gathered on the device (RCCL):
 *   d_saddles      room for saddle_capacity agx_saddle records; the frames' lists are packed
 *                  back to back in completion order (each list itself in reference order);
 *   d_frame_table  n_frames agx_frame_result entries: where frame i's list starts, its length
 *                  and its status flags (0 = ok; AGX_FRAME_* bits otherwise).
 * A frame whose list does not fit (or whose internal lists overflowed) gets a non-zero
 * status and no records -- never a truncated list. */
typedef struct agx_frame_result {
    uint32_t count;   /* saddles of this frame                     */
    uint32_t offset;  /* index of its first record in d_saddles    */
    uint32_t status;  /* 0 or a combination of AGX_FRAME_* bits    */
    uint32_t n_clusters;
} agx_frame_result;
enum {
    AGX_FRAME_CANDIDATE_OVERFLOW = 1,
    AGX_FRAME_CLUSTER_OVERFLOW = 2,
    AGX_FRAME_SADDLE_OVERFLOW = 4,
    AGX_FRAME_CENTROID_INEXACT = 8, /* informational: a cluster's coordinate sum reached 2^24 */
    AGX_FRAME_GENERIC_PATH = 16,    /* informational: clustered by the generic kernels         */
    AGX_FRAME_DENSE_THRESHOLD = 32, /* (unused) */
    AGX_FRAME_LARGE_RESULT = 64     /* informational: more than 1024 refined candidates, emitted by the large-list path */
};
int agx_saddles_batch_enqueue_to(agx_detector *det, const void *d_frames, int n_frames, int width,
                                 int height, size_t row_stride_bytes, size_t frame_stride_bytes,
                                 int format, void *d_saddles, uint32_t saddle_capacity,
                                 void *d_frame_table);

/* Block until everything enqueued on the detector's stream has finished. */
int agx_detector_sync(agx_detector *det);

/* How many host threads this process can keep busy: the smaller of its CPU affinity mask and its
 * cgroup CPU quota (cpu.max / cpu.cfs_quota_us; a container given 16 CPUs of a 256-thread host reports
 * 16), at least 1.  The default thread count of agx_detect_batch: threads beyond a quota make every
 * thread of the process -- the one driving the device included -- sit out the rest of each scheduler
 * period.  (The reference never spawns a thread, src/detector.rs:505; batches are this library's.) */
int agx_host_parallelism(void);
/* Test hook: the CPU-quota half of that rule on a given cgroup mount point and stand-in for /proc/self/cgroup (v2 cpu.max
 * and v1 cpu.cfs_quota_us / cpu.cfs_period_us of the process's cgroup and every ancestor, the smallest, rounded up to whole
 * CPUs); 0 = no quota found.  Nothing is cached. */
int agx_debug_cgroup_cpu_quota(const char *cgroup_root, const char *proc_self_cgroup);

/* TagDetector::detect (src/detector.rs:505-540) over a batch of equally sized frames in HOST memory
 * (frame i at frames + i*frame_stride_bytes; formats AGX_L8 / AGX_L16 / AGX_RGB8).  With the device tail (option
 * "device_tail", the default where available) the saddle chain and the board search + decode of a chunk of up to 1024
 * frames run on the device while n_threads host threads (0 = agx_host_parallelism(); the pool lives as long as the
 * detector) upload the next chunk and take the frames the kernel hands back.  With the host tail the saddle
 * chain of a chunk of frames (about one frame per thread, 8 .. 64) runs on the device while the threads upload the next
 * chunks and run the board search + decode of the previous ones; no barrier between chunks.  Same tags either way.
 * d_frames: optional device copy of the same frames (skips the upload), else
 * NULL.  out: n_frames * cap_per_frame tags, frame i at out + i*cap_per_frame; counts[i] = tags of
 * frame i; frame_status[i] (may be NULL) = AGX_OK or AGX_ERR_CAPACITY.  Returns the first non-OK
 * frame status, else AGX_OK. */
int agx_detect_batch(agx_detector *det, const void *frames, const void *d_frames, int n_frames, int width,
                     int height, size_t row_stride_bytes, size_t frame_stride_bytes, int format, agx_tag *out,
                     uint32_t cap_per_frame, uint32_t *counts, int *frame_status, int n_threads);

/* ---- detector groups: several GPUs of one node driven from ONE process ----------------- */

/* The reference's detect(&self) is stateless, so a batch shards by frame:
 * rank r of a group owns one device, runs the whole chain for its own frames on its own stream,
 * and the only exchange step is the gather of the per-rank result slabs to the root device
 * (devices[0]).  A Rust host drives all GPUs of a node through these calls without torch. */
typedef struct agx_group agx_group;
enum {
    AGX_GATHER_RCCL = 0, /* ncclSend / ncclRecv in one group over xGMI (librccl opened on first use);
                            devices must be distinct */
    AGX_GATHER_PEER = 1  /* hipMemcpyPeerAsync + events over the same links; accepts a device more than
                            once (test configuration on one-GPU boxes) */
};
/* One detector (TagDetector::new, src/detector.rs:364-406) per entry of devices[] (NULL:
 * 0..n_devices-1). */
int agx_group_create(int family, const agx_params *params, const int *devices, int n_devices, int transport,
                     agx_group **out);
void agx_group_destroy(agx_group *group);
int agx_group_size(const agx_group *group);
/* Borrow rank r's detector (limits, options, profiling); owned by the group. */
agx_detector *agx_group_detector(agx_group *group, int rank);
/* refined_saddle_points (src/detector.rs:408-446) over n_devices * frames_per_rank frames: rank r's
 * frames are resident on ITS device at d_frames[r] (layout as agx_saddles_batch_enqueue).  Enqueues
 * every rank's chain and the gather and returns without waiting.  records_per_frame: average saddle
 * records per frame the per-rank result slab holds (0 = 512). */
int agx_group_saddles_enqueue(agx_group *group, const void *const *d_frames, int frames_per_rank, int width,
                              int height, size_t row_stride_bytes, size_t frame_stride_bytes, int format,
                              uint32_t records_per_frame);
/* Wait for the gather; frame f of rank r is global frame r*frames_per_rank + f of out / counts /
 * frame_status (sized n_devices*frames_per_rank, as agx_saddles_batch_fetch). */
int agx_group_saddles_fetch(agx_group *group, agx_saddle *out, uint32_t cap_per_frame, uint32_t *counts,
                            int *frame_status);
/* group == NULL: the reason of this thread's last failed agx_group_create. */
const char *agx_group_last_error(const agx_group *group);

/* Host tail only: TagDetector::detect's board search + decode (src/detector.rs:510-539)
 * from a saddle list and the u8 luma plane (to_luma8, :507), both in host memory.
 * saddles is not modified. */
int agx_detect_from_saddles(const agx_detector *det, const agx_saddle *saddles, uint32_t n_saddles,
                            const uint8_t *luma8, int width, int height, size_t row_stride_bytes,
                            agx_tag *out, uint32_t cap, uint32_t *n_out);

/* The same host tail without a detector handle (no device needed): family and params as in
 * agx_detector_create (params may be NULL). */
int agx_detect_tail(int family, const agx_params *params, const agx_saddle *saddles, uint32_t n_saddles,
                    const uint8_t *luma8, int width, int height, size_t row_stride_bytes, agx_tag *out,
                    uint32_t cap, uint32_t *n_out);

/* The same with the board search of the frame on n_threads host threads (created for this call; a
 * detector handle keeps its own: option "tail_threads").  Same result as n_threads = 1. */
int agx_detect_tail_threads(int family, const agx_params *params, const agx_saddle *saddles, uint32_t n_saddles,
                            const uint8_t *luma8, int width, int height, size_t row_stride_bytes, agx_tag *out,
                            uint32_t cap, uint32_t *n_out, int n_threads);

/* to_luma8 (src/detector.rs:507) of a host image into a tightly packed host plane. */
int agx_luma8(const void *pixels, int width, int height, size_t row_stride_bytes, int format,
              uint8_t *out);

/* ---- measurement and parity-test hooks ------------------------------------------------ */

/* Per-kernel device time of the chain, from hipEvents recorded on the detector's stream
 * around each launch while profiling is on.  names/ms/launches are arrays of
 * AGX_N_KERNELS entries (the chain has four launches since round 3: entries behind the last one
 * carry a NULL name and zeros); ms accumulates since the last reset. */
#define AGX_N_KERNELS 5
int agx_profile_enable(agx_detector *det, int on); /* 0 off, 1 = the blur kernel only (2 events per batch), 2 = every kernel */
int agx_profile_reset(agx_detector *det);
int agx_profile_read(agx_detector *det, const char **names, double *ms_total, uint64_t *launches);

/* Copy an intermediate product of frame `frame` of the last batch to host memory.
 * what: AGX_DBG_BLUR / AGX_DBG_RESP (width*height floats; AGX_DBG_RESP is the Hessian response
 * the blur kernel evaluated in its registers and needs option "store_response" = 1 set before
 * the batch was enqueued -- the chain itself never stores it), AGX_DBG_MIN (1 float),
 * AGX_DBG_CENTERS (n clusters * {u32 first_index, u32 size, f32 cx, f32 cy} sorted by
 * first_index), AGX_DBG_REFINED (unfiltered rochade_refine output, agx_saddle each, in
 * cluster order).  *n_items receives the element count; returns AGX_ERR_CAPACITY if
 * cap_bytes is too small. */
enum { AGX_DBG_BLUR = 0, AGX_DBG_RESP = 1, AGX_DBG_MIN = 2, AGX_DBG_CENTERS = 3, AGX_DBG_REFINED = 4,
       AGX_DBG_COUNTERS = 5, /* 8 x uint32: status flags (AGX_FRAME_*), flood seeds, second-tier seeds,
                                clusters, generic-path candidates, generic-path roots, refined, saddles */
       AGX_DBG_RESP_RECOMPUTED = 6, /* width*height floats: the response recomputed from the stored blur
                                       plane by a separate kernel (cross-check of AGX_DBG_RESP) */
       AGX_DBG_VERIFY_STATS = 7,    /* 20 x uint32: re-test statistics of K2 (debug_ablation bits 128 / 2048), phase times of
                                       K2 (8192) / of the flood + refine kernel (16384) in 10 ns ticks */
       AGX_DBG_LUMA8 = 9,   /* width*height bytes: to_luma8 as the device computed it for the last agx_detect on an
                               L16 / RGB8 image (agx_detect converts on the device; agx_luma8 is the host's) */
       AGX_DBG_WAVE_TIMES = 10, /* pairs of uint64 (start, end; 10 ns ticks of s_memrealtime) of every wave of one sparse
                                   kernel of the last batch; `frame` selects the kernel (1 verify, 2 flood + refine, 3 rare);
                                   needs debug_ablation & 4096 (tools/wave_timeline.py) */
       AGX_DBG_REDZONES = 8, /* 6 x uint32: guarded buffers, damaged guard bytes, first damaged buffer, its byte
                               offset from the payload start (int32), device address of buffer 0 (lo, hi: for the
                               check of the check).  Buffers = the chain's workspace, then the staging buffer, the luma
                               planes and the device tail's code list, tag rows and frame table (mapped pinned host memory)
                               as far as they exist.  Needs no enqueued batch.  Guard bytes exist only in handles created with
                               AGX_REDZONE_BYTES=<n> in the environment (memory-safety tests of the kernels) */
       AGX_DBG_TAIL_TABLE_ADDR = 11 /* 2 x uint64: host address and payload bytes of the device tail's frame table (the check
                                       of the check for a buffer in mapped host memory).  Needs no enqueued batch */ };
typedef struct agx_cluster_info {
    uint32_t first_index, size;
    float cx, cy;
} agx_cluster_info;
/* Test hook of the host tail: for n pairs of vectors (v0x, v0y, v1x, v1y) the reference's
 * angle(v0, v1) in degrees (math_util.rs:27-33) and the bounded approximation the board search uses
 * to decide threshold comparisons that are not close (csrc/host_tail.cpp, LazyAngle); has_approx[i] = 0
 * where the approximation is not used (zero / non-finite operands).  The CPU suite checks the bound. */
int agx_debug_angle_pairs(const float *vectors, size_t n, float *exact, float *approx, uint8_t *has_approx);
/* The coarser first-level approximation in front of it (a three-term polynomial in float, max error 0.04 degrees, guard band
 * 0.1): most of the search's ~16 000 angle comparisons per frame are decided from it.  Same hook, same check. */
int agx_debug_angle_pairs_coarse(const float *vectors, size_t n, float *coarse, uint8_t *has_coarse);
/* Test hook of the device tail (option "device_tail"): the kernel evaluates angle_degree's atan2f by glibc's own
 * single-precision routine, restated (csrc/libm_f32.h).  *mismatches = on how many of n pseudo-random operand pairs (any two
 * floats, cross / dot products of image-sized vectors, ratios at the ends of the routine's reduction intervals, plus the
 * special cases) this process's atan2f disagrees with the restatement; the option is refused unless that is 0. */
int agx_debug_libm_atan2f_check(uint64_t n, uint64_t seed, uint64_t *mismatches);
/* The one test of is_valid_quad the device tail cannot restate bit for bit is "filter white block" (saddle.rs:26-38: cosf, sinf):
 * for n triples (s0.theta, v02.x, v02.y) the angle as the reference evaluates it (binary32, this process's libm) and as the
 * kernel's decisive evaluation does (binary64).  The kernel decides 60 <= angle <= 120 from the latter only when it is farther
 * than 1e-4 degrees from both thresholds; the CPU suite checks that the two never differ by more than half of that. */
int agx_debug_white_block_angles(const float *triples, size_t n, float *reference, double *binary64);
int agx_debug_fetch(agx_detector *det, int frame, int what, void *host_out, size_t cap_bytes,
                    size_t *n_items);

/* Constants computed at create time (for parity tests): 7 blur weights, 25 cone taps,
 * 25x6 pseudo-inverse (row i, column j at [i*6+j]). */
int agx_detector_constants(const agx_detector *det, float *blur_w7, float *cone25, float *pmat150);

const char *agx_status_string(int status);
/* Message of the last failure on this detector (HIP error text etc.); never NULL.
 * det == NULL: the reason of the last failed agx_detector_create. */
const char *agx_last_error(const agx_detector *det);
int agx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* APRILGRID_AMD_H */
