// chain_kernels.h -- host-visible interface of the HIP saddle chain (chain_kernels.hip).
// Internal to the library; the public boundary is include/aprilgrid_amd.h.
#pragma once
#include <cstddef>
#include <cstdint>

namespace agx {

enum Kernel : int {
    K_BLUR_HESSIAN = 0,  // K1: luma -> 7-tap blur -> Hessian determinant -> per-frame min
    K_THRESHOLD = 1,     // K2: resp < 0.05*min  -> candidate list + slot plane
    K_UNION = 2,         // K3a: 4-connected union-find over candidates
    K_CENTROID = 3,      // K3b: root lookup, centroid sums, root list
    K_REFINE = 4,        // K4: rochade_refine per cluster
    K_FILTER_SORT = 5,   // K5: k/phi filter, reference-order emission
    K_COUNT = 6
};

// One record of K4's output list.
struct RefinedRec {
    uint32_t key;  // smallest linear pixel index of the cluster == reference emission rank
    float x, y, k, theta, phi;
};

// Per-frame counters, zeroed (hipMemsetAsync) before every batch.
struct FrameCounters {
    uint32_t min_key_inv;  // ~order_preserving(min response); atomicMax
    uint32_t n_cand;       // candidates appended by K2 (may exceed capacity -> flag)
    uint32_t n_roots;      // clusters
    uint32_t n_refined;    // rochade_refine survivors
    uint32_t max_k_bits;   // max k (k >= 0 so the raw bits order correctly)
    uint32_t flags;        // FLAG_* below
    uint32_t n_out;        // saddles after the filter
    uint32_t out_offset;   // start of this frame's saddles in the compact output array
};
enum : uint32_t {
    FLAG_CAND_OVERFLOW = 1u,
    FLAG_ROOT_OVERFLOW = 2u,
    FLAG_OUT_OVERFLOW = 4u,
    FLAG_CENTROID_INEXACT = 8u  // a cluster's coordinate sum reached 2^24 (f32 sums of the
                                // reference would round there)
};

struct RefineConsts {
    float cone[25];    // normalised 5x5 cone kernel, detector.rs:240-254
    float pmat[150];   // 25x6 pseudo-inverse, [i*6+j], detector.rs:208-237
};

// Device workspace + geometry of one batch.  All pointers are device pointers.
struct ChainArgs {
    // input
    const uint8_t *frames;
    long long frame_stride;  // bytes
    int row_stride;          // bytes, multiple of 4
    int fmt;                 // agx_format
    int W, H, n_frames;
    long long plane;  // W*H
    // K1 tiling (host-chosen, see plan_k1)
    int threads;        // workgroup size (multiple of 64)
    int n_strips;       // column strips per row
    int strip_cols;     // columns per strip (multiple of 4)
    int rows_per_seg;   // output rows per workgroup
    int n_segs;
    float w[7];  // blur taps
    // dense planes [n_frames][H][W]
    float *blur;
    float *resp;
    uint32_t *slot_plane;
    // per-frame
    FrameCounters *ctr;
    uint32_t *total_out;  // single counter: compact output allocation
    // candidate arrays [n_frames][cap_cand]
    uint32_t cap_cand, cap_roots, cap_out;
    uint32_t *cand;    // pixel index | left<<30 | up<<31
    uint32_t *parent;
    uint32_t *sumx, *sumy, *cnt, *minidx;
    uint32_t *roots;      // [n_frames][cap_roots]
    RefinedRec *refined;  // [n_frames][cap_roots]
    // output
    float *out;  // compact agx_saddle array (internal: n_frames*cap_out records, or caller-owned)
    uint32_t out_total_cap;  // records `out` can hold
    uint32_t *frame_table;   // optional caller-owned [n_frames][4]: count, offset, status, clusters
    float min_angle, max_angle;
};

// Choose K1's tiling for a frame size / batch size.  Returns false if unsupported.
bool plan_k1(ChainArgs &a, int override_rows_per_seg);

// Enqueue one kernel of the chain on `stream` (hipStream_t as void*).  Returns hipError_t.
int launch_kernel(int which, const ChainArgs &a, const RefineConsts &rc, void *stream);

size_t k5_lds_bytes(const ChainArgs &a);

}  // namespace agx
