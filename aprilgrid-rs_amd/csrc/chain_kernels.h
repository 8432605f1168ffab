// chain_kernels.h -- host-visible interface of the HIP saddle chain (chain_kernels.hip).
// Internal to the library; the public boundary is include/aprilgrid_amd.h.
#pragma once
#include <cstddef>
#include <cstdint>

namespace agx {

enum Kernel : int {
    K_BLUR_HESSIAN = 0,  // K1: luma -> 7-tap blur (stored) -> Hessian determinant -> per-frame min
    K_THRESHOLD = 1,     // K2: verify K1's candidate superset mask against the final threshold
                         //     (exact response recomputed from the blur plane at set bits only),
                         //     then mask scan -> flood seeds
    K_FLOOD_REFINE = 2,  // K3: bit-parallel flood fill per seed (32x32 window per lane) -> cluster record ->
                         //     rochade_refine of the cluster, by the same lane
    K_RARE = 3,          // K4: per frame: second flood tier (128x64 per wave), generic clustering fallback for frames
                         //     with larger components, k/phi filter and reference-order emission; clears the
                         //     next batch's counters
    K_SPARSE = 4,        // K2 + K3 + K4 of one frame in one 1024-thread workgroup (k_sparse_frame): batches that fill the chip
                         //     with one workgroup per frame run K1 + K_SPARSE, the others K1 .. K4
    K_COUNT = 5
};

// One record of K4's output list.
struct RefinedRec {
    uint32_t key;  // smallest linear pixel index of the cluster == reference emission rank
    float x, y, k, theta, phi;
};

// Per-frame counters, zeroed (hipMemsetAsync) before every batch.
struct FrameCounters {
    // its own 128-byte line: every wave of K1 publishes to / polls this word
    uint32_t min_key_inv;  // ~order_preserving(min response); atomicMax
    uint32_t pad0[31];
    uint32_t n_seeds;      // flood seeds
    uint32_t n_clusters;   // cluster records of the first flood tier (k_flood_refine) or of the generic path
    uint32_t n_refined;    // rochade_refine survivors
    uint32_t max_k_bits;   // max k (k >= 0 so the raw bits order correctly)
    uint32_t flags;        // FLAG_* below
    uint32_t n_out;        // saddles after the filter
    uint32_t out_offset;   // start of this frame's saddles in the compact output array
    uint32_t n_cand;       // generic path: candidate pixels
    uint32_t n_roots;      // generic path: union-find roots
    uint32_t n_big;        // seeds handed to the wave-wide second flood tier
    uint32_t refine_done;  // emit_large_split (k_rare with several workgroups per frame): lock + state of the frame's place in the
                           // compact output -- 0 free, 1 taken by the part that arrived first (CAS), 2 allocated and it fits,
                           // 3 allocated but over capacity; the frame's other parts spin on it until it is >= 2.  Must be 0 at
                           // batch start (the counters' clearing provides that).  Not touched on any other path.
    uint32_t n_clusters2;  // (unused since the second flood tier runs inside k_flood_refine; always 0)
    uint32_t stats[20];    // debug_ablation & 128: verify statistics by word row within a 128-row segment
};
static_assert(sizeof(FrameCounters) == 256, "FrameCounters is cleared as 64 dwords");
enum : uint32_t {
    FLAG_CAND_OVERFLOW = 1u,   // seed list (fast path) or candidate list (generic path) full
    FLAG_ROOT_OVERFLOW = 2u,   // cluster list full
    FLAG_OUT_OVERFLOW = 4u,
    FLAG_CENTROID_INEXACT = 8u,  // a cluster's coordinate sum reached 2^24 (f32 sums of the
                                 // reference would round there)
    FLAG_BIG_CLUSTER = 16u,      // a component left the flood windows: frame redone generically
    FLAG_LARGE_RESULT = 64u      // more than 1024 refined records: emitted by k_rare's large-list sort
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
    int row_stride;          // bytes
    int byte_rows;           // rows / frames are not 4-byte aligned: the blur kernel gathers bytes
    int fmt;                 // agx_format
    int W, H, n_frames;
    long long plane;  // W*H
    // K1 tiling (host-chosen, see plan_k1)
    int threads;        // workgroup size (multiple of 64)
    int n_strips;       // column strips per row
    int strip_cols;     // columns per strip (multiple of 4)
    int rows_per_seg;   // output rows per workgroup
    int n_segs;
    int k1_group;       // frames per dispatch group of K1 (0 or >= n_frames: the whole batch segment-major)
    int k1_async_poll;  // K1 polls the frame's running minimum by asynchronous vector loads (few waves) / awaited scalar loads
    float publish_factor;  // a wave publishes its running minimum m only if m < factor * the frame's known minimum
    float w[7];  // blur taps
    // dense planes [n_frames][H][W]
    float *blur;
    float *resp_dbg;       // parity tests only ("store_response"): K1's in-register response, [n_frames][H][W]; else null
    float *dummy;          // one row (W + 8 floats): target of K1's out-of-segment stores
    // weakest (largest) response among the candidates K1 admitted, per 4 columns x 32 rows:
    // [n_frames][mask_yb][mask_wpr / 4] (same indexing as the mask, x/4); -inf where none
    float *cand_max;
    uint32_t *slot_plane;  // generic path only (sparse-touched)
    // Candidate bit mask, TRANSPOSED: one word = 32 consecutive rows of one column.
    // mask[frame][yb][MASK_PAD_X + x] holds rows 32*yb .. 32*yb+31 of column x (bit = row & 31).
    // MASK_PAD_X zero words on both sides of a word row and the word rows past the image are
    // never written and stay zero from allocation (flood windows reach into them).
    uint32_t *mask;
    int mask_wpr;            // words per word-row: round_up4(W + 2*MASK_PAD_X)
    int mask_yb;             // word rows: H/32 + 4
    long long mask_plane;    // words per frame = mask_wpr * mask_yb
    int force_generic;       // test hook: treat every frame as FLAG_BIG_CLUSTER
    int sparse_after_verify; // k_sparse_frame runs behind k_verify_seeds (sparse path 3): it skips its own verify stage
    // debug_ablation & 4096: every wave of the sparse kernels records when it started and ended (100 MHz
    // constant clock, s_memrealtime) -- record (kernel - 1) * WAVE_TIMES_STRIDE + blockIdx.x of this array
    // (two 64-bit words each; the generic path's slot plane serves as storage).  Else null.
    unsigned long long *wave_times;
    int dbg;                 // timing ablations only (results invalid): 1 = K1 skips blur stores,
                             // 4 = K1 skips the Hessian/min, 8 = K1 stores into an L2-resident
                             // region, 16 = no shared-min refresh
    // per-frame
    FrameCounters *ctr;
    FrameCounters *ctr_next;  // the other counter set: k_rare clears records [0, n_frames] of it for the next batch (or null)
    uint32_t *total_out;  // single counter: compact output allocation
    uint32_t cap_cand, cap_roots, cap_out;
    uint32_t *seeds;      // [n_frames][cap_roots] pixel index of each flood seed
    // cluster records [n_frames][cap_roots]
    uint32_t *clu_key;    // smallest pixel index of the cluster
    uint32_t *clu_cnt;
    uint32_t *clu_sx, *clu_sy;  // integer coordinate sums (K4 leaves the f32 centroid bits here)
    // generic path: candidate arrays [n_frames][cap_cand]
    uint32_t *cand;    // pixel index | left<<30 | up<<31
    uint32_t *parent;
    unsigned long long *sumx, *sumy;
    uint32_t *cnt, *minidx;
    uint32_t *roots;      // [n_frames][cap_roots]
    RefinedRec *refined;  // [n_frames][cap_roots]
    // output
    float *out;  // compact agx_saddle array (internal: n_frames*cap_out records, or caller-owned)
    uint32_t out_total_cap;  // records `out` can hold
    uint32_t *frame_table;   // optional caller-owned [n_frames][4]: count, offset, status, clusters
    float min_angle, max_angle;
};

// Tuning overrides from the environment (AGX_K1_*, AGX_G_*, AGX_RARE_PARTS, AGX_SPARSE_PATH: measurement only).  A name is read
// from the environment once per process and kept; tuning_env_reload() forgets what was read (option "reload_tuning_env").
int tuning_env(const char *name, int dflt);
void tuning_env_reload();

// Choose K1's tiling for a frame size / batch size.  Returns false if unsupported.
bool plan_k1(ChainArgs &a, int override_rows_per_seg);

// Enqueue one kernel of the chain on `stream` (hipStream_t as void*).  Returns hipError_t.
int launch_kernel(int which, const ChainArgs &a, const RefineConsts &rc, void *stream);

size_t k5_lds_bytes(const ChainArgs &a);

constexpr size_t WAVE_TIMES_STRIDE = (size_t)1 << 20;  // records per kernel (debug_ablation & 4096)

constexpr int MASK_PAD_X = 64;  // zero columns on each side of the transposed mask (flood windows)  // flood window rows below the last image row

// Debug: recompute the Hessian response plane of `frame` from its blur plane into dst.
// Per-device kernel attributes (current device); hipError_t.
int init_device_kernels();

int launch_debug_resp(const ChainArgs &a, int frame, float *dst, void *stream);
// The batch's frame rows ([n_frames + 1][4]: count, offset, flags, clusters; the last row: total) and compact saddle array into
// mapped pinned host memory (device addresses h_table_dev / h_out_dev) by a kernel on `stream`.
int launch_publish(const ChainArgs &a, uint32_t h_out_records, uint32_t *h_table_dev, float *h_out_dev, void *stream);
// Zero n_records counter records on `stream` (used instead of a memset while the stream is being captured).
int launch_clear_counters(FrameCounters *ctr, size_t n_records, void *stream);
// u8 luma (to_luma8) of n_frames L16 (format 1) / RGB8 (format 2) frames in device memory: rows `pitch`
// bytes apart, frames `frame_stride` bytes apart -> tight [n_frames][H][W]
int launch_luma8(const void *src, size_t pitch, size_t frame_stride, int n_frames, int format, uint8_t *dst, int W, int H,
                 void *stream);

}  // namespace agx
