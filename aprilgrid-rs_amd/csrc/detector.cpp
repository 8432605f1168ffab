// detector.cpp -- implementation of the C ABI in include/aprilgrid_amd.h: the detector handle
// (mirror of aprilgrid::detector::TagDetector), device workspace, chain enqueue / fetch,
// profiling events and the parity-test hooks.  Compiled by hipcc; links only libamdhip64.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <system_error>
#include <vector>

#include "../../include/aprilgrid_amd.h"
#include "chain_kernels.h"
#include "detector_internal.h"
#include "host_tail.hpp"
#include "tail_kernels.h"

using namespace agx;

namespace {

const char *kKernelNames[K_COUNT] = {"k_blur_hessian", "k_verify_seeds", "k_flood_refine", "k_rare_emit", "k_sparse_frame"};

struct EventPair {
    hipEvent_t a, b;
    int kernel;
};

}  // namespace

struct agx_detector {
    int family = AGX_T36H11;
    FamilyInfo fam{};
    agx_params params{};
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    RefineConsts rc{};
    float blur_w[7]{};
    uint32_t lim_cand = 0, lim_roots = 0, lim_out = 0;
    int force_generic = 0;
    int k1_rows = 0;
    int sparse_path = 0;  // option "sparse_path": 0 = by batch size, 1 = K2 + K3 + K4 (three launches), 2 = k_sparse_frame (one)
    bool last_sparse_frame = false;  // the last batch ran K1 + K_SPARSE
    int last_sparse_path = 1;        // 1 three launches, 2 k_sparse_frame alone, 3 k_verify_seeds + k_sparse_frame
    int n_cus = 0;                   // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    int dbg = 0;
    float *d_dbg_resp = nullptr;  // lazily allocated plane for agx_debug_fetch(AGX_DBG_RESP_RECOMPUTED)
    long long dbg_resp_plane = 0;
    int store_resp = 0;           // option "store_response": K1's parity-test instantiation
    float *d_resp_store = nullptr;  // [n_frames][H][W] planes it writes
    size_t resp_store_floats = 0;
    bool resp_stored = false;     // the last batch ran with store_response
    int ws_W = 0, ws_H = 0;       // geometry the mask plane was last zeroed for

    // workspace (device)
    ChainArgs args{};
    size_t cap_frames = 0;        // frames the dense planes hold
    long long cap_plane = 0;      // pixels per frame the dense planes hold
    uint32_t alloc_cand = 0, alloc_roots = 0, alloc_out = 0;
    std::vector<void *> device_allocs;
    // AGX_REDZONE_BYTES (environment, read when the handle is created; tests only): every workspace
    // buffer gets this many guard bytes in front and behind, filled with 0xA5; agx_debug_fetch
    // (AGX_DBG_REDZONES) reports the guard bytes that no longer hold the pattern
    size_t redzone = 0;
    std::vector<size_t> alloc_bytes;  // payload bytes per entry of device_allocs
    // Buffers outside the chain's workspace (staging, luma planes, the device tail's code list and result tables): the same
    // guard bytes around each, device memory or mapped pinned host memory; AGX_DBG_REDZONES counts them after the workspace's
    struct SideBuf {
        void *base = nullptr;  // start of the front guard (what hipMalloc / hipHostMalloc returned)
        size_t bytes = 0;      // payload
        bool host = false;     // pinned host memory (mapped when dev != nullptr)
        void *dev = nullptr;   // device address of the payload of a mapped host buffer
    };
    enum { SB_STAGE, SB_LUMA_D, SB_LUMA_H, SB_CODES, SB_TAGS, SB_TAIL_TABLE, SB_COUNT };
    SideBuf side[SB_COUNT];
    // staging for the single-frame host API
    uint8_t *d_stage = nullptr;
    size_t stage_bytes = 0;
    // agx_detect on L16 / RGB8: the u8 luma of the staged frame, computed on the device
    uint8_t *d_luma = nullptr, *h_luma = nullptr;
    size_t luma_bytes = 0;
    // pinned host mirrors
    FrameCounters *h_ctr = nullptr;
    size_t h_ctr_frames = 0;
    uint32_t *h_total = nullptr;
    float *h_out = nullptr;
    size_t h_out_records = 0;
    float *d_out_internal = nullptr;  // workspace copy of args.out
    float *h_out_dev = nullptr;       // device address of h_out (mapped pinned memory): a single frame's list is written there directly
    bool out_in_host = false;         // last batch's compact output went straight to h_out
    uint32_t *h_table = nullptr, *h_table_dev = nullptr;  // mapped pinned [frames + 1][4]: per frame count, offset, status, clusters (k_publish / the single frame's k_rare)
    size_t h_table_rows = 0;
    size_t mask_words = 0;
    bool external_out = false;       // last batch wrote into caller-owned device memory

    // two counter sets used alternately: the last kernel of a batch clears the other set, so only a
    // batch that finds its set not known to be clear pays a memset
    FrameCounters *d_ctr[2] = {nullptr, nullptr};
    size_t ctr_cleared[2] = {0, 0};  // records of the set known to be zero (0 = in use / unknown)
    int ctr_cur = 0;

    bool enqueued = false;
    int profiling = 0;  // 0 off, 1 = K1 only, 2 = every kernel
    int prof_stride = 1;        // level 1: time the selected kernel of every prof_stride-th batch only
    int prof_kernel = K_BLUR_HESSIAN;  // level 1: which kernel (option "profile_kernel", default the blur kernel)
    uint64_t prof_batches = 0;  // batches enqueued while profiling
    std::vector<EventPair> pending_events;
    std::vector<hipEvent_t> free_events;
    double prof_ms[K_COUNT]{};
    uint64_t prof_launches[K_COUNT]{};

    // agx_detect_batch: the next chunk's upload runs on a stream of its own under the current chunk's chain and fetch
    hipStream_t upload_streams[AGX_UPLOAD_STREAMS] = {nullptr, nullptr, nullptr};  // agx_detect_batch: one per staging slot, all or none
    bool upload_streams_ready = false;
    TailWorkers *tail_workers = nullptr;  // option "tail_threads" > 1: one frame's board search on several threads
    int tail_threads = 1;
    void *pool = nullptr;  // agx_detect_batch: worker threads of the host tail
    int pool_threads = 0;
    std::vector<agx_saddle> scratch_saddles;  // host staging of agx_detect / agx_detect_planes (reused)

    // option "device_tail": agx_detect_batch's board search + decode on the device (tail_kernels.hip); frames the kernel
    // hands back (TAIL_UNCERTAIN / TAIL_CAPACITY) take the host tail
    int device_tail = -1;  // -1: by the batch's size, where this process's atan2f is the routine the kernel restates; 0 off; 1 on
    bool tail_ready = false;        // code list on the device and the kernel's attributes set for this device: all or nothing
    int tail_debug_band_mdeg = 0;   // option "tail_debug_band" (tests of the hand-back path), thousandths of a degree
    uint64_t *d_codes = nullptr;                                     // the family's code list
    agx_tag *h_tags = nullptr, *h_tags_dev = nullptr;                // mapped pinned [tail_frames][tail_tag_cap]
    uint32_t *h_tail_table = nullptr, *h_tail_table_dev = nullptr;   // mapped pinned [tail_frames][4]: count, status, ticks, saddles | seeds << 16
    size_t tail_frames = 0;
    uint32_t tail_tag_cap = 0;
    int last_tail_frames = 0, last_tail_fallbacks = 0, last_tail_uncertain = 0;  // of the last agx_detect_batch call

    std::string last_error;
};

namespace {

thread_local std::string g_create_error;  // reason of this thread's last failed agx_detector_create (det == NULL)

int fail(agx_detector *d, int status, const std::string &msg)
{
    if (d) d->last_error = msg;
    return status;
}

// Nothing unwinds across the C boundary (include/aprilgrid_amd.h, "Conventions"): every entry point runs its body through
// agx_guard.  A failed host allocation or thread creation becomes AGX_ERR_NOMEM, anything else AGX_ERR_STATE; the message
// goes to agx_last_error (best effort: storing it must not throw either).  Visible, like the reference's panic
// (src/detector.rs:500) -- and recoverable, unlike an exception that reaches a Rust frame.
void set_error_noexcept(agx_detector *d, const char *msg) noexcept
{
    try {
        if (d) d->last_error = msg;
        else g_create_error = msg;
    } catch (...) {
    }
}

template <typename F>
int agx_guard(const agx_detector *det_c, F &&body) noexcept
{
    agx_detector *det = const_cast<agx_detector *>(det_c);
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_error_noexcept(det, "out of host memory");
        return AGX_ERR_NOMEM;
    } catch (const std::system_error &e) {  // std::thread: no more threads / resources
        set_error_noexcept(det, e.what());
        return AGX_ERR_NOMEM;
    } catch (const std::exception &e) {
        set_error_noexcept(det, e.what());
        return AGX_ERR_STATE;
    } catch (...) {
        set_error_noexcept(det, "unknown exception");
        return AGX_ERR_STATE;
    }
}

#define HIP_TRY(det, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((det), AGX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Blur taps, reference src/image_util.rs:111-124 (sigma = 1.5 at the call site detector.rs:410)
void make_blur_weights(float sigma, float w[7])
{
    const int radius = (int)std::ceil(sigma * 2.0f);  // == 3
    const float two_sigma_sq = 2.0f * sigma * sigma;
    float sum = 0.0f;
    for (int i = 0; i < 2 * radius + 1; ++i) {
        const float x = (float)(i - radius);
        const float v = std::exp(-(x * x) / two_sigma_sq);
        w[i] = v;
        sum += v;
    }
    for (int i = 0; i < 2 * radius + 1; ++i) w[i] /= sum;
}

// Constants of rochade_refine for half_size_patch = 2 (the only value the reference passes,
// detector.rs:430): cone kernel (:240-254) and the 25x6 pseudo-inverse of the quadratic
// design matrix (:208-237).  The design's normal matrix is block diagonal on the symmetric
// 5x5 grid (odd moments vanish): {xy}, {x}, {y} decouple and {x^2, y^2, 1} is a 3x3 block,
// inverted here by its adjugate in binary64; one rounding to binary32.
void make_refine_consts(RefineConsts &rc)
{
    const int half = 2, ks = 5;
    double sx2 = 0, sx4 = 0;
    for (int c = 0; c < ks; ++c) {
        const double x = c - half;
        sx2 += x * x;
        sx4 += x * x * x * x;
    }
    const double n1 = ks;
    // moments over the grid
    const double m_x4 = n1 * sx4, m_x2y2 = sx2 * sx2, m_x2 = n1 * sx2, m_1 = n1 * n1;
    // block {x^2, y^2, 1}
    const double B[3][3] = {{m_x4, m_x2y2, m_x2}, {m_x2y2, m_x4, m_x2}, {m_x2, m_x2, m_1}};
    double adj[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int r0 = (j + 1) % 3, r1 = (j + 2) % 3, c0 = (i + 1) % 3, c1 = (i + 2) % 3;
            adj[i][j] = B[r0][c0] * B[r1][c1] - B[r0][c1] * B[r1][c0];
        }
    const double det = B[0][0] * adj[0][0] + B[0][1] * adj[1][0] + B[0][2] * adj[2][0];
    int i = 0;
    for (int r = 0; r < ks; ++r)
        for (int c = 0; c < ks; ++c, ++i) {
            const double x = c - half, y = r - half;
            const double q[3] = {x * x, y * y, 1.0};
            double blk[3];
            for (int a = 0; a < 3; ++a) blk[a] = (adj[a][0] * q[0] + adj[a][1] * q[1] + adj[a][2] * q[2]) / det;
            rc.pmat[i * 6 + 0] = (float)blk[0];          // x^2
            rc.pmat[i * 6 + 1] = (float)(x * y / m_x2y2 + 0.0); // xy (+0.0: no negative zeros)
            rc.pmat[i * 6 + 2] = (float)blk[1];          // y^2
            rc.pmat[i * 6 + 3] = (float)(x / m_x2 + 0.0);      // x
            rc.pmat[i * 6 + 4] = (float)(y / m_x2 + 0.0);      // y
            rc.pmat[i * 6 + 5] = (float)blk[2];          // 1
        }
    const float gamma = (float)half;
    float s = 0.0f;
    for (int a = 0; a < ks; ++a)
        for (int b = 0; b < ks; ++b) {
            const float da = gamma - (float)a, db = gamma - (float)b;
            rc.cone[a * ks + b] = std::max(0.0f, gamma + 1.0f - std::sqrt(da * da + db * db));
        }
    for (int a = 0; a < ks * ks; ++a) s += rc.cone[a];
    for (int a = 0; a < ks * ks; ++a) rc.cone[a] = rc.cone[a] / s;
}

template <typename T>
int dev_alloc(agx_detector *d, T *&ptr, size_t count)
{
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T), rz = d->redzone;
    HIP_TRY(d, hipMalloc(&p, bytes + 2 * rz));
    d->device_allocs.push_back(p);
    d->alloc_bytes.push_back(bytes);
    if (rz) {
        HIP_TRY(d, hipMemset(p, 0xA5, rz));
        HIP_TRY(d, hipMemset((char *)p + rz + bytes, 0xA5, rz));
    }
    ptr = (T *)((char *)p + rz);
    return AGX_OK;
}

// A buffer outside the workspace, with the handle's guard bytes around it.  kind: 0 device, 1 pinned host, 2 pinned host
// mapped into the device (b.dev = its device address).  The old buffer is freed first; on failure the slot is empty.
void side_free(agx_detector *d, int which)
{
    agx_detector::SideBuf &b = d->side[which];
    if (b.base) (void)(b.host ? hipHostFree(b.base) : hipFree(b.base));
    b = agx_detector::SideBuf();
}
void *side_alloc(agx_detector *d, int which, size_t bytes, int kind)
{
    side_free(d, which);
    agx_detector::SideBuf &b = d->side[which];
    const size_t rz = d->redzone, payload = std::max<size_t>(bytes, 1);
    void *p = nullptr;
    if (kind == 0) {
        if (hipMalloc(&p, payload + 2 * rz) != hipSuccess) return nullptr;
        if (rz && (hipMemset(p, 0xA5, rz) != hipSuccess || hipMemset((char *)p + rz + payload, 0xA5, rz) != hipSuccess)) {
            (void)hipFree(p);
            return nullptr;
        }
    } else {
        if (hipHostMalloc(&p, payload + 2 * rz, kind == 2 ? hipHostMallocMapped : hipHostMallocDefault) != hipSuccess) return nullptr;
        if (rz) {
            std::memset(p, 0xA5, rz);
            std::memset((char *)p + rz + payload, 0xA5, rz);
        }
        if (kind == 2) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) {
                (void)hipHostFree(p);
                return nullptr;
            }
            b.dev = (char *)dp + rz;
        }
    }
    b.base = p;
    b.bytes = payload;
    b.host = kind != 0;
    return (char *)p + rz;
}

void free_workspace(agx_detector *d)
{
    for (void *p : d->device_allocs) (void)hipFree(p);
    d->device_allocs.clear();
    d->alloc_bytes.clear();
    d->cap_frames = 0;
    d->cap_plane = 0;
    d->d_ctr[0] = d->d_ctr[1] = nullptr;
    d->ctr_cleared[0] = d->ctr_cleared[1] = 0;
    if (d->h_ctr) (void)hipHostFree(d->h_ctr);
    if (d->h_out) (void)hipHostFree(d->h_out);
    d->h_ctr = nullptr;
    d->h_out = nullptr;
    d->h_ctr_frames = 0;
    d->h_out_records = 0;
}

uint32_t clamp_u32(unsigned long long v, uint32_t lo, uint32_t hi)
{
    return (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>(v, lo), hi);
}

size_t mask_words_per_frame(int W, int H) { return (size_t)((W + 2 * MASK_PAD_X + 3) & ~3) * (size_t)(H / 32 + 4); }

// The mask's pad words / pad rows are never written by the kernels and must read as zero: the
// plane is cleared whenever the frame geometry (hence the mask layout) changes.
int set_mask_geometry(agx_detector *d, int W, int H)
{
    ChainArgs &a = d->args;
    a.mask_wpr = (W + 2 * MASK_PAD_X + 3) & ~3;
    a.mask_yb = H / 32 + 4;
    a.mask_plane = (long long)a.mask_wpr * a.mask_yb;
    if (d->ws_W != W || d->ws_H != H) {
        HIP_TRY(d, hipMemsetAsync(a.mask, 0, d->mask_words * sizeof(uint32_t), d->stream));
        d->ws_W = W;
        d->ws_H = H;
    }
    return AGX_OK;
}

// (Re)allocate the workspace for n_frames frames of W x H.  Never called inside a timed
// region once a configuration has been seen.
int ensure_workspace(agx_detector *d, int n_frames, int W, int H)
{
    const long long plane = (long long)W * H;
    const uint32_t cap_cand = d->lim_cand ? d->lim_cand : clamp_u32((unsigned long long)plane / 2, 4096, 1u << 28);
    const uint32_t cap_roots = d->lim_roots ? d->lim_roots : clamp_u32((unsigned long long)plane / 8, 1024, 1u << 26);
    const uint32_t cap_out = d->lim_out ? d->lim_out : clamp_u32((unsigned long long)plane / 64, 256, 1u << 24);
    ChainArgs &a = d->args;
    const bool fits = (size_t)n_frames <= d->cap_frames && plane <= d->cap_plane && cap_cand <= d->alloc_cand &&
                      cap_roots <= d->alloc_roots && cap_out <= d->alloc_out &&
                      (size_t)n_frames * mask_words_per_frame(W, H) <= d->mask_words;
    if (fits) {
        a.cap_cand = cap_cand;
        a.cap_roots = cap_roots;
        a.cap_out = cap_out;
        return set_mask_geometry(d, W, H);
    }
    HIP_TRY(d, hipStreamSynchronize(d->stream));
    free_workspace(d);
    const size_t F = (size_t)n_frames;
    int rc;
    const size_t mask_plane = mask_words_per_frame(W, H);
    if ((rc = dev_alloc(d, a.blur, F * plane + 16))) return rc;  // +16: aligned window loads may touch 3 floats past the end
    if ((rc = dev_alloc(d, a.dummy, (size_t)1 << 16))) return rc;  // any W < 65520
    if ((rc = dev_alloc(d, a.cand_max, F * mask_plane / 4 + 16))) return rc;
    if ((rc = dev_alloc(d, a.slot_plane, F * plane))) return rc;
    if ((rc = dev_alloc(d, a.mask, F * mask_plane))) return rc;
    d->mask_words = F * mask_plane;
    for (int p = 0; p < 2; ++p) {
        if ((rc = dev_alloc(d, d->d_ctr[p], F + 1))) return rc;  // + one extra record: its first word is total_out
        d->ctr_cleared[p] = 0;
    }
    a.ctr = d->d_ctr[0];
    a.total_out = &a.ctr[F].min_key_inv;
    if ((rc = dev_alloc(d, a.seeds, F * cap_roots))) return rc;
    if ((rc = dev_alloc(d, a.clu_key, F * cap_roots))) return rc;
    if ((rc = dev_alloc(d, a.clu_cnt, F * cap_roots))) return rc;
    if ((rc = dev_alloc(d, a.clu_sx, F * cap_roots))) return rc;
    if ((rc = dev_alloc(d, a.clu_sy, F * cap_roots))) return rc;
    if ((rc = dev_alloc(d, a.cand, F * cap_cand))) return rc;
    if ((rc = dev_alloc(d, a.parent, F * cap_cand))) return rc;
    if ((rc = dev_alloc(d, a.sumx, F * cap_cand))) return rc;
    if ((rc = dev_alloc(d, a.sumy, F * cap_cand))) return rc;
    if ((rc = dev_alloc(d, a.cnt, F * cap_cand))) return rc;
    if ((rc = dev_alloc(d, a.minidx, F * cap_cand))) return rc;
    if ((rc = dev_alloc(d, a.roots, F * cap_roots))) return rc;
    if ((rc = dev_alloc(d, a.refined, F * cap_roots))) return rc;
    if ((rc = dev_alloc(d, d->d_out_internal, F * cap_out * 5 + 4))) return rc;  // (+4: k_publish copies 16 bytes at a time)
    d->ws_W = d->ws_H = 0;  // forces the mask to be zeroed below
    HIP_TRY(d, hipHostMalloc((void **)&d->h_ctr, (F + 1) * sizeof(FrameCounters), hipHostMallocDefault));
    d->h_total = (uint32_t *)(d->h_ctr + F);
    d->h_ctr_frames = F;
    HIP_TRY(d, hipHostMalloc((void **)&d->h_out, (F * cap_out * 5 + 4) * sizeof(float), hipHostMallocMapped));
    d->h_out_dev = nullptr;
    if (hipHostGetDevicePointer((void **)&d->h_out_dev, d->h_out, 0) != hipSuccess) d->h_out_dev = nullptr;
    if (d->h_table_rows < F + 1) {
        if (d->h_table) (void)hipHostFree(d->h_table);
        d->h_table = d->h_table_dev = nullptr;
        d->h_table_rows = 0;
        HIP_TRY(d, hipHostMalloc((void **)&d->h_table, (F + 1) * 4 * sizeof(uint32_t), hipHostMallocMapped));
        if (hipHostGetDevicePointer((void **)&d->h_table_dev, d->h_table, 0) != hipSuccess) d->h_table_dev = nullptr;
        d->h_table_rows = F + 1;
    }
    d->h_out_records = F * cap_out;
    d->cap_frames = F;
    d->cap_plane = plane;
    d->alloc_cand = a.cap_cand = cap_cand;
    d->alloc_roots = a.cap_roots = cap_roots;
    d->alloc_out = a.cap_out = cap_out;
    return set_mask_geometry(d, W, H);
}

hipEvent_t get_event(agx_detector *d)
{
    if (!d->free_events.empty()) {
        hipEvent_t e = d->free_events.back();
        d->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Fold finished event pairs into the per-kernel totals (caller has synchronised the stream).
void harvest_events(agx_detector *d)
{
    for (EventPair &p : d->pending_events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            d->prof_ms[p.kernel] += ms;
            d->prof_launches[p.kernel] += 1;
        }
        d->free_events.push_back(p.a);
        d->free_events.push_back(p.b);
    }
    d->pending_events.clear();
}

// The chain of one chunk (frames [f0, f0+nf) of the batch) on stream `st`.
int enqueue_chunk(agx_detector *d, int f0, int nf, hipStream_t st)
{
    ChainArgs a = d->args;  // shifted copy
    const size_t F0 = (size_t)f0;
    a.n_frames = nf;
    a.frames += F0 * (size_t)a.frame_stride;
    a.blur += F0 * (size_t)a.plane;
    a.slot_plane += F0 * (size_t)a.plane;
    if (a.resp_dbg) a.resp_dbg += F0 * (size_t)a.plane;
    a.cand_max += F0 * (size_t)(a.mask_plane / 4);
    a.mask += F0 * (size_t)a.mask_plane;
    a.ctr += F0;
    if (a.ctr_next) a.ctr_next += F0;
    a.seeds += F0 * a.cap_roots;
    a.clu_key += F0 * a.cap_roots;
    a.clu_cnt += F0 * a.cap_roots;
    a.clu_sx += F0 * a.cap_roots;
    a.clu_sy += F0 * a.cap_roots;
    a.cand += F0 * a.cap_cand;
    a.parent += F0 * a.cap_cand;
    a.sumx += F0 * a.cap_cand;
    a.sumy += F0 * a.cap_cand;
    a.cnt += F0 * a.cap_cand;
    a.minidx += F0 * a.cap_cand;
    a.roots += F0 * a.cap_roots;
    a.refined += F0 * a.cap_roots;
    if (a.frame_table) a.frame_table += F0 * 4;
    // The sparse phase: one workgroup per frame doing all of it (k_sparse_frame) when the batch fills the chip that way,
    // else the three batch-wide launches.  (Timing ablations and the wave timeline instrument the three launches.)
    int path = d->sparse_path;
    {   // measurement override (read from the environment once per process, chain_kernels.h): 0 .. 3 or ignored
        const int forced = tuning_env("AGX_SPARSE_PATH", -1);
        if (forced >= 0 && forced <= 3) path = forced;
    }
    const int sparse_dbg = 32 | 64 | 128 | 256 | 2048 | 4096 | 8192 | 16384;  // debug_ablation bits that instrument K2 / K3 / K4
    // by batch size: one workgroup per frame pays when the frames fill the chip's 256 CUs in whole rounds (the last round at least
    // three quarters full); then k_verify_seeds keeps its launch (8 waves per SIMD, balanced over the whole batch: 40 us against the
    // 33 .. 75 us per frame of the verify stage inside k_sparse_frame) and flood + refine + emission share one (path 3)
    const int cus = d->n_cus > 0 ? d->n_cus : 256, nearly = cus - cus / 4;  // (MI355X: 256 CUs, 192)
    if (path == 0) path = (nf >= nearly && (nf % cus == 0 || nf % cus >= nearly)) ? 3 : 1;
    const bool fused = (path == 2 || path == 3) && !(a.dbg & sparse_dbg);
    d->last_sparse_frame = fused;
    d->last_sparse_path = fused ? path : 1;
    a.sparse_after_verify = fused && path == 3;
    const int plan_fused[] = {K_BLUR_HESSIAN, K_SPARSE}, plan_multi[] = {K_BLUR_HESSIAN, K_THRESHOLD, K_FLOOD_REFINE, K_RARE},
              plan_v_fe[] = {K_BLUR_HESSIAN, K_THRESHOLD, K_SPARSE};
    const int *plan = fused ? (path == 3 ? plan_v_fe : plan_fused) : plan_multi;
    const int n_plan = fused ? (path == 3 ? 3 : 2) : 4;
    for (int pi = 0; pi < n_plan; ++pi) {
        const int k = plan[pi];
        EventPair ev{nullptr, nullptr, k};
        // (an event pair costs the stream two ~5 us gaps around the kernel: level 1 can sample)
        const bool timed = d->profiling >= 2 || (d->profiling == 1 && k == d->prof_kernel && d->prof_batches % (uint64_t)d->prof_stride == (uint64_t)d->prof_stride - 1);  // the last of each group: never the first batch after an idle stream
        if (timed) {
            ev.a = get_event(d);
            ev.b = get_event(d);
            HIP_TRY(d, hipEventRecord(ev.a, st));
        }
        hipError_t e = (hipError_t)launch_kernel(k, a, d->rc, st);
        if (e != hipSuccess)
            return fail(d, AGX_ERR_HIP, std::string("launch ") + kKernelNames[k] + ": " + hipGetErrorString(e));
        if (timed) {
            HIP_TRY(d, hipEventRecord(ev.b, st));
            d->pending_events.push_back(ev);
        }
    }
    if (d->profiling) ++d->prof_batches;
    return AGX_OK;
}

int enqueue_chain(agx_detector *d)
{
    ChainArgs &a = d->args;
    // counters of the batch's frames and the compact-output cursor: the set the previous batch's last
    // kernel cleared, or -- first batch, larger batch, after an error -- one memset
    const int p = d->ctr_cur ^ 1;
    const size_t need = (size_t)a.n_frames + 1;
    // While the stream is being captured into a HIP graph the memset is always recorded: a replayed
    // graph must clear its own set every time (the clearing by the previous batch's last kernel is a fact
    // about the capture, not about the replays -- an odd number of captured batches would otherwise
    // accumulate counters from the second replay on), and nothing "known clear" survives the capture.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(d->stream, &cap);
    const bool capturing = cap == hipStreamCaptureStatusActive;
    if (capturing) {  // (a kernel node: see launch_clear_counters)
        HIP_TRY(d, (hipError_t)launch_clear_counters(d->d_ctr[p], need, d->stream));
    } else if (d->ctr_cleared[p] < need) {
        HIP_TRY(d, hipMemsetAsync(d->d_ctr[p], 0, need * sizeof(FrameCounters), d->stream));
    }
    d->ctr_cleared[p] = 0;
    a.ctr = d->d_ctr[p];
    a.ctr_next = d->d_ctr[p ^ 1];
    a.total_out = &a.ctr[a.n_frames].min_key_inv;
    d->ctr_cleared[p ^ 1] = 0;
    // (splitting a batch over several streams -- whole chain or sparse kernels only -- was measured
    // and lost every time: see DESIGN.md; batches in flight are separate detectors)
    const int rc = enqueue_chunk(d, 0, a.n_frames, d->stream);
    if (rc) return rc;
    d->ctr_cleared[p ^ 1] = capturing ? 0 : need;  // k_rare of this batch clears it, stream-ordered before the next batch
    d->ctr_cur = p;
    d->enqueued = true;
    return AGX_OK;
}

int frame_status_of(const FrameCounters &c, uint32_t cap_per_frame)
{
    if (c.flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) return AGX_ERR_CAPACITY;
    if (c.n_out > cap_per_frame) return AGX_ERR_CAPACITY;
    return AGX_OK;
}

bool valid_format(int f) { return f == AGX_L8 || f == AGX_L16 || f == AGX_RGB8 || f == AGX_LF32; }
int bytes_per_px(int f) { return f == AGX_L8 ? 1 : (f == AGX_L16 ? 2 : (f == AGX_RGB8 ? 3 : 4)); }
const char *kFormatMsg = "format must be AGX_L8, AGX_L16, AGX_RGB8 or AGX_LF32";

}  // namespace

extern "C" {

int agx_abi_version(void) { return AGX_ABI_VERSION; }

__attribute__((visibility("hidden"))) void *agx_internal_stream(agx_detector *det) { return det ? (void *)det->stream : nullptr; }
__attribute__((visibility("hidden"))) int agx_internal_device(const agx_detector *det) { return det ? det->device : -1; }
__attribute__((visibility("hidden"))) void *agx_internal_pool(agx_detector *det, int n_threads)
{
    if (!det || n_threads < 1) return nullptr;
    if (det->pool && det->pool_threads != n_threads) {
        destroy_worker_pool(det->pool);
        det->pool = nullptr;
    }
    if (!det->pool) {
        det->pool_threads = 0;
        det->pool = create_worker_pool(n_threads);  // (throws where threads cannot be created: agx_detect_batch catches)
        det->pool_threads = n_threads;
    }
    return det->pool;
}
// agx_detect_batch: u8 luma of a chunk of L16 / RGB8 frames (device pointers), computed on the device
// behind whatever is on the detector's stream and copied into pinned host memory: [n_frames][H][W] at
// *h_out (valid after the stream has been waited for -- the chunk's fetch does).  The staging is a ring of
// n_slots chunks; the caller reuses a slot when the host tails that read it are done.
__attribute__((visibility("hidden"))) int agx_internal_chunk_luma8(agx_detector *det, const void *d_frames, int n_frames,
                                                                   int width, int height, size_t row_stride,
                                                                   size_t frame_stride, int format, int slot, int n_slots,
                                                                   size_t chunk_capacity_frames, const uint8_t **h_out,
                                                                   const uint8_t **d_out)
{
    const size_t plane = (size_t)width * (size_t)height, one = plane * chunk_capacity_frames;
    if (slot < 0 || slot >= n_slots) return AGX_ERR_ARG;
    if ((size_t)n_slots * one > det->luma_bytes) {
        if (hipStreamSynchronize(det->stream) != hipSuccess) return AGX_ERR_HIP;
        det->luma_bytes = 0;
        det->d_luma = static_cast<uint8_t *>(side_alloc(det, agx_detector::SB_LUMA_D, (size_t)n_slots * one, 0));
        det->h_luma = static_cast<uint8_t *>(side_alloc(det, agx_detector::SB_LUMA_H, (size_t)n_slots * one, 1));
        if (!det->d_luma || !det->h_luma) return AGX_ERR_HIP;
        det->luma_bytes = (size_t)n_slots * one;
    }
    uint8_t *d = det->d_luma + (size_t)slot * one, *h = det->h_luma + (size_t)slot * one;
    if (launch_luma8(d_frames, row_stride, frame_stride, n_frames, format, d, width, height, det->stream) != 0) return AGX_ERR_HIP;
    if (d_out) *d_out = d;  // (the device tail reads it there; no copy to the host unless asked for)
    if (!h_out) return AGX_OK;
    if (hipMemcpyAsync(h, d, plane * (size_t)n_frames, hipMemcpyDeviceToHost, det->stream) != hipSuccess) return AGX_ERR_HIP;
    *h_out = h;
    return AGX_OK;
}
// agx_detect_batch: the upload streams, one per staging slot (created on first use, all of them or none: a failure
// half way destroys what exists, so that a later call starts over instead of finding a half-initialised set); hipError_t
__attribute__((visibility("hidden"))) int agx_internal_upload_streams(agx_detector *det, void **streams)
{
    if (!det->upload_streams_ready) {
        hipError_t e = hipSuccess;
        for (int i = 0; i < AGX_UPLOAD_STREAMS && e == hipSuccess; ++i)
            e = hipStreamCreateWithFlags(&det->upload_streams[i], hipStreamNonBlocking);
        if (e != hipSuccess) {
            for (int i = 0; i < AGX_UPLOAD_STREAMS; ++i) {
                if (det->upload_streams[i]) (void)hipStreamDestroy(det->upload_streams[i]);
                det->upload_streams[i] = nullptr;
            }
            return (int)e;
        }
        det->upload_streams_ready = true;
    }
    for (int i = 0; i < AGX_UPLOAD_STREAMS; ++i) streams[i] = det->upload_streams[i];
    return 0;
}
// agx_detect_batch: the last batch's results without the [frame][cap] layout of agx_saddles_batch_fetch: waits for the
// device, then *records = the batch's compact list in the detector's pinned host mirror (valid until the next enqueue;
// frame f's list = counts[f] records from offsets[f]) and status[f] = AGX_OK / AGX_ERR_CAPACITY (a device-side list of
// the frame overflowed: its count is 0).  Returns AGX_OK, or the HIP / state error of the fetch.
__attribute__((visibility("hidden"))) int agx_internal_fetch_compact(agx_detector *det, const agx_saddle **records, uint32_t *counts,
                                                                     uint32_t *offsets, int *status)
{
    const int n = det->enqueued ? det->args.n_frames : 0;
    // cap 0: counts and status only (a frame with any saddles reads "capacity" against cap 0: the real status is
    // derived from the counters below, with no limit on the list's length)
    int rc = agx_saddles_batch_fetch(det, nullptr, 0, counts, status);
    if (rc != AGX_OK && rc != AGX_ERR_CAPACITY) return rc;
    for (int f = 0; f < n; ++f) {
        const FrameCounters &c = det->h_ctr[f];
        status[f] = frame_status_of(c, 0xffffffffu);
        counts[f] = status[f] == AGX_OK ? c.n_out : 0;
        offsets[f] = c.out_offset;
    }
    *records = reinterpret_cast<const agx_saddle *>(det->h_out);
    det->last_error.clear();
    return AGX_OK;
}
// agx_detect_batch with option "device_tail": the board search + decode of the batch that was just enqueued, behind it on
// the detector's stream.  d_luma = the frames' u8 luma in device memory (L8 frames: the frames themselves).  Results go to
// mapped pinned host memory; agx_internal_fetch_tail waits for them.
// the device evaluates angle_degree's atan2f by glibc's routine (libm_f32.h): offered only where this process's atan2f IS that
// routine -- checked once per process on 2^20 operand pairs (more in tests/test_abi_cpu.py)
static uint64_t libm_check_once()
{
    // (AGX_DEBUG_LIBM_MISMATCH=1: tests of the refusal path pretend that one input differs)
    static const uint64_t mismatches = libm_atan2f_mismatches(1u << 20, 1) + (uint64_t)(tuning_env("AGX_DEBUG_LIBM_MISMATCH", 0) != 0);
    return mismatches;
}
// 0: the host tail; 1: the device tail (asked for); 2: the device tail is available and the call may choose by its size
__attribute__((visibility("hidden"))) int agx_internal_device_tail(agx_detector *det)
{
    if (det->device_tail < 0) return libm_check_once() == 0 ? 2 : 0;
    return det->device_tail;
}
__attribute__((visibility("hidden"))) void agx_internal_tail_stats(agx_detector *det, int frames, int fallbacks, int uncertain)
{
    det->last_tail_frames = frames;
    det->last_tail_fallbacks = fallbacks;
    det->last_tail_uncertain = uncertain;
}
// One-time set-up of the device tail on this handle's device: the family's code list in device memory and the kernel's
// attributes (155 KB of LDS).  All or nothing: a failure leaves nothing behind, and a later call starts over.
__attribute__((visibility("hidden"))) int agx_internal_tail_prepare(agx_detector *det)
{
    if (det->tail_ready) return AGX_OK;
    if (hipSetDevice(det->device) != hipSuccess) return AGX_ERR_HIP;
    det->d_codes = static_cast<uint64_t *>(side_alloc(det, agx_detector::SB_CODES, (size_t)det->fam.n_codes * sizeof(uint64_t), 0));
    if (!det->d_codes ||
        hipMemcpy(det->d_codes, det->fam.codes, (size_t)det->fam.n_codes * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess ||
        init_tail_kernels() != 0) {
        side_free(det, agx_detector::SB_CODES);
        det->d_codes = nullptr;
        return AGX_ERR_HIP;
    }
    det->tail_ready = true;
    return AGX_OK;
}
__attribute__((visibility("hidden"))) int agx_internal_tail_debug(const agx_detector *) { return tuning_env("AGX_TAIL_DEBUG", 0); }  // (read at create)
__attribute__((visibility("hidden"))) int agx_internal_enqueue_tail(agx_detector *det, const void *d_luma, size_t luma_row_stride,
                                                                    size_t luma_frame_stride, uint32_t tag_cap)
{
    if (!det->enqueued || det->external_out) return AGX_ERR_STATE;
    const ChainArgs &a = det->args;
    if (tag_cap == 0 || luma_row_stride > 0x7fffffffu) return AGX_ERR_ARG;
    if (tag_cap > 128u) tag_cap = 128u;  // (the kernel's own list of distinct ids; frames beyond it take the host tail)
    if (!det->tail_ready) {
        const int rc = agx_internal_tail_prepare(det);
        if (rc) return rc;
    }
    if ((size_t)a.n_frames > det->tail_frames || tag_cap > det->tail_tag_cap) {
        if (hipStreamSynchronize(det->stream) != hipSuccess) return AGX_ERR_HIP;
        const size_t F = std::max((size_t)a.n_frames, det->tail_frames), cap = std::max(tag_cap, det->tail_tag_cap);
        det->tail_frames = 0;  // (nothing usable until both tables exist)
        det->tail_tag_cap = 0;
        det->h_tags = static_cast<agx_tag *>(side_alloc(det, agx_detector::SB_TAGS, F * cap * sizeof(agx_tag), 2));
        det->h_tail_table = static_cast<uint32_t *>(side_alloc(det, agx_detector::SB_TAIL_TABLE, F * 4 * sizeof(uint32_t), 2));
        if (!det->h_tags || !det->h_tail_table) {
            side_free(det, agx_detector::SB_TAGS);
            side_free(det, agx_detector::SB_TAIL_TABLE);
            det->h_tags = det->h_tags_dev = nullptr;
            det->h_tail_table = det->h_tail_table_dev = nullptr;
            return AGX_ERR_HIP;
        }
        det->h_tags_dev = static_cast<agx_tag *>(det->side[agx_detector::SB_TAGS].dev);
        det->h_tail_table_dev = static_cast<uint32_t *>(det->side[agx_detector::SB_TAIL_TABLE].dev);
        det->tail_frames = F;
        det->tail_tag_cap = (uint32_t)cap;
    }
    TailArgs t{};
    t.saddles = a.out;
    t.ctr = a.ctr;
    t.n_frames = a.n_frames;
    t.luma = static_cast<const uint8_t *>(d_luma);
    t.luma_frame_stride = (long long)luma_frame_stride;
    t.luma_row_stride = (int)luma_row_stride;
    t.W = a.W;
    t.H = a.H;
    t.edge = det->fam.edge;
    t.border = det->fam.border;
    t.hamming = det->fam.hamming;
    t.n_codes = det->fam.n_codes;
    t.codes = det->d_codes;
    t.max_boards = det->params.max_num_of_boards;
    t.tags = det->h_tags_dev;
    t.table = det->h_tail_table_dev;
    t.tag_cap = tag_cap;  // this call's: a frame with more tags is handed back (TAIL_CAPACITY), whatever the table could hold
    t.tag_stride = det->tail_tag_cap;
    t.debug_band = (float)det->tail_debug_band_mdeg * 1e-3f;
    t.debug = tuning_env("AGX_TAIL_DEBUG", 0);
    t.debug_frame = tuning_env("AGX_TAIL_DEBUG_FRAME", 0);
    if (launch_board_tail(t, det->stream) != 0) return AGX_ERR_HIP;
    return AGX_OK;
}
__attribute__((visibility("hidden"))) int agx_internal_fetch_tail(agx_detector *det, const agx_tag **tags, const uint32_t **table, uint32_t *tag_cap)
{
    if (hipStreamSynchronize(det->stream) != hipSuccess) return AGX_ERR_HIP;
    *tags = det->h_tags;
    *table = det->h_tail_table;
    *tag_cap = det->tail_tag_cap;
    return AGX_OK;
}
__attribute__((visibility("hidden"))) void agx_internal_abandon_batch(agx_detector *det)
{
    if (!det) return;
    (void)hipStreamSynchronize(det->stream);
    harvest_events(det);
    det->enqueued = false;
}
__attribute__((visibility("hidden"))) const void *agx_internal_family(const agx_detector *det) { return &det->fam; }
__attribute__((visibility("hidden"))) int agx_internal_max_boards(const agx_detector *det) { return det->params.max_num_of_boards; }
__attribute__((visibility("hidden"))) void *agx_internal_stage(agx_detector *det, size_t bytes)
{
    if (bytes > det->stage_bytes) {
        if (hipSetDevice(det->device) != hipSuccess || hipStreamSynchronize(det->stream) != hipSuccess) return nullptr;
        det->stage_bytes = 0;
        det->d_stage = static_cast<uint8_t *>(side_alloc(det, agx_detector::SB_STAGE, bytes, 0));
        if (!det->d_stage) return nullptr;
        det->stage_bytes = bytes;
    }
    return det->d_stage;
}

const char *agx_status_string(int status)
{
    switch (status) {
    case AGX_OK: return "ok";
    case AGX_ERR_ARG: return "invalid argument";
    case AGX_ERR_FORMAT: return "unsupported pixel format";
    case AGX_ERR_CAPACITY: return "capacity exceeded";
    case AGX_ERR_HIP: return "HIP runtime error";
    case AGX_ERR_NO_DEVICE: return "no usable gfx950 device";
    case AGX_ERR_FAMILY: return "unknown tag family";
    case AGX_ERR_STATE: return "invalid call sequence";
    case AGX_ERR_NOMEM: return "out of host memory or threads";
    default: return "unknown status";
    }
}

const char *agx_last_error(const agx_detector *det) { return det ? det->last_error.c_str() : g_create_error.c_str(); }

int agx_family_from_str(const char *name, int *family_out)
{
    return agx_guard(nullptr, [&]() -> int {
    if (!name || !family_out) return AGX_ERR_ARG;
    static const struct { const char *lo, *up; int fam; } kNames[] = {
        {"t16h5", "T16H5", AGX_T16H5},     {"t25h7", "T25H7", AGX_T25H7},
        {"t25h9", "T25H9", AGX_T25H9},     {"t36h11", "T36H11", AGX_T36H11},
        {"t36h11b1", "T36H11B1", AGX_T36H11B1}};
    for (const auto &n : kNames)
        if (!std::strcmp(name, n.lo) || !std::strcmp(name, n.up)) {
            *family_out = n.fam;
            return AGX_OK;
        }
    return AGX_ERR_FAMILY;
    });
}

void agx_default_params(agx_params *out)
{
    if (!out) return;
    out->tag_spacing_ratio = 0.3f;
    out->min_saddle_angle = 30.0f;
    out->max_saddle_angle = 60.0f;
    out->max_num_of_boards = 2;
}

int agx_detector_create(int family, const agx_params *params, int device, agx_detector **out)
{
    return agx_guard(nullptr, [&]() -> int {
    if (!out) return AGX_ERR_ARG;
    *out = nullptr;
    g_create_error.clear();
    FamilyInfo fam;
    if (!family_info(family, fam)) return AGX_ERR_FAMILY;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
        g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e) + ", devices=" +
                         std::to_string(n_dev) + ", requested " + std::to_string(device);
        return AGX_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return AGX_ERR_NO_DEVICE;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {  // the code object is gfx950 only
        g_create_error = std::string("device arch is '") + prop.gcnArchName + "', need gfx950";
        return AGX_ERR_NO_DEVICE;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return AGX_ERR_NO_DEVICE;
    }
    struct Unwind {  // (an exception or an early return behind the stream's creation)
        void operator()(agx_detector *p) const
        {
            if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
            delete p;
        }
    };
    tuning_env_reload();  // the AGX_* measurement overrides are read when a detector is created (and kept: no getenv per launch)
    std::unique_ptr<agx_detector, Unwind> d(new agx_detector());
    d->family = family;
    d->fam = fam;
    if (params) d->params = *params;
    else agx_default_params(&d->params);
    d->device = device;
    d->n_cus = prop.multiProcessorCount;
    if (const char *rz = std::getenv("AGX_REDZONE_BYTES")) {  // tests: guard bytes around every workspace buffer
        const long v = std::atol(rz);
        if (v > 0 && v <= (1 << 24)) d->redzone = ((size_t)v + 255) & ~(size_t)255;  // keeps the buffers' alignment
    }
    e = hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_create_error = std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e);
        return AGX_ERR_HIP;
    }
    d->stream = d->own_stream;
    e = (hipError_t)init_device_kernels();  // function attributes are per device
    if (e != hipSuccess) {
        g_create_error = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e);
        return AGX_ERR_HIP;
    }
    make_blur_weights(1.5f, d->blur_w);
    make_refine_consts(d->rc);
    *out = d.release();
    return AGX_OK;
    });
}

void agx_detector_destroy(agx_detector *det)
{
    if (!det) return;
    try {
    (void)hipSetDevice(det->device);
    (void)hipStreamSynchronize(det->stream);
    harvest_events(det);
    for (hipEvent_t e : det->free_events) (void)hipEventDestroy(e);
    free_workspace(det);
    if (det->pool) destroy_worker_pool(det->pool);
    if (det->tail_workers) destroy_tail_workers(det->tail_workers);
    for (int i = 0; i < agx_detector::SB_COUNT; ++i) side_free(det, i);  // staging, luma planes, the device tail's buffers
    if (det->h_table) (void)hipHostFree(det->h_table);
    if (det->d_dbg_resp) (void)hipFree(det->d_dbg_resp);
    if (det->d_resp_store) (void)hipFree(det->d_resp_store);
    for (int i = 0; i < AGX_UPLOAD_STREAMS; ++i)
        if (det->upload_streams[i]) (void)hipStreamDestroy(det->upload_streams[i]);
    if (det->own_stream) (void)hipStreamDestroy(det->own_stream);
    } catch (...) {  // (joining worker threads can throw std::system_error)
    }
    delete det;
}

int agx_detector_family_info(const agx_detector *det, int *edge_bits, int *border_bits, int *hamming_distance,
                             const uint64_t **codes, int *n_codes)
{
    return agx_guard(det, [&]() -> int {
    if (!det) return AGX_ERR_ARG;
    if (edge_bits) *edge_bits = det->fam.edge;
    if (border_bits) *border_bits = det->fam.border;
    if (hamming_distance) *hamming_distance = det->fam.hamming;
    if (codes) *codes = det->fam.codes;
    if (n_codes) *n_codes = det->fam.n_codes;
    return AGX_OK;
    });
}

int agx_detector_set_limits(agx_detector *det, uint32_t max_candidates, uint32_t max_clusters, uint32_t max_saddles)
{
    return agx_guard(det, [&]() -> int {
    if (!det) return AGX_ERR_ARG;
    if (max_candidates >= (1u << 30) || max_saddles > (1u << 24)) return AGX_ERR_ARG;
    det->lim_cand = max_candidates;
    det->lim_roots = max_clusters;
    det->lim_out = max_saddles;
    return AGX_OK;
    });
}

int agx_detector_set_stream(agx_detector *det, void *hip_stream, int external)
{
    return agx_guard(det, [&]() -> int {
    if (!det) return AGX_ERR_ARG;
    HIP_TRY(det, hipSetDevice(det->device));
    HIP_TRY(det, hipStreamSynchronize(det->stream));
    harvest_events(det);
    det->stream = external ? (hipStream_t)hip_stream : det->own_stream;
    return AGX_OK;
    });
}

int agx_detector_set_option(agx_detector *det, const char *name, int value)
{
    return agx_guard(det, [&]() -> int {
    if (!det || !name) return AGX_ERR_ARG;
    if (!std::strcmp(name, "force_generic")) det->force_generic = value != 0;
    else if (!std::strcmp(name, "k1_rows_per_segment")) det->k1_rows = value > 0 ? value : 0;
    else if (!std::strcmp(name, "sparse_path")) det->sparse_path = value >= 0 && value <= 3 ? value : 0;
    else if (!std::strcmp(name, "debug_ablation")) det->dbg = value;  // timing only, results invalid
    else if (!std::strcmp(name, "store_response")) det->store_resp = value != 0;
    else if (!std::strcmp(name, "profile_stride")) det->prof_stride = value > 1 ? value : 1;
    else if (!std::strcmp(name, "profile_kernel")) det->prof_kernel = value >= 0 && value < K_COUNT ? value : K_BLUR_HESSIAN;
    else if (!std::strcmp(name, "device_tail")) {  // 1 on (refused where libm differs), 0 off, -1 back to the default (on where possible)
        if (value > 0) {
            const uint64_t mismatches = libm_check_once();
            if (mismatches) {
                det->device_tail = 0;
                return fail(det, AGX_ERR_STATE, "device_tail: this C library's atan2f is not the routine the device tail restates (" +
                                                    std::to_string(mismatches) + " of 2^20 inputs differ); the host tail stays in use");
            }
        }
        det->device_tail = value > 0 ? 1 : (value < 0 ? -1 : 0);
    }
    else if (!std::strcmp(name, "tail_debug_band")) det->tail_debug_band_mdeg = value > 0 ? std::min(value, 30000) : 0;  // tests: thousandths of a degree
    else if (!std::strcmp(name, "reload_tuning_env")) tuning_env_reload();  // (process-wide: the AGX_* overrides are read again)
    else if (!std::strcmp(name, "tail_threads")) {
        const int n = value < 1 ? 1 : (value > 64 ? 64 : value);
        if (n != det->tail_threads) {
            if (det->tail_workers) destroy_tail_workers(det->tail_workers);
            det->tail_workers = nullptr;  // (creating the new ones can fail: AGX_ERR_NOMEM, the option back at one thread)
            det->tail_threads = 1;
            det->tail_workers = create_tail_workers(n);
            det->tail_threads = n;
        }
    } else return fail(det, AGX_ERR_ARG, std::string("unknown option ") + name);
    return AGX_OK;
    });
}

int agx_detector_get_option(const agx_detector *det, const char *name, int *value)
{
    return agx_guard(det, [&]() -> int {
    if (!det || !name || !value) return AGX_ERR_ARG;
    const ChainArgs &a = det->args;
    if (!std::strcmp(name, "force_generic")) *value = det->force_generic;
    else if (!std::strcmp(name, "store_response")) *value = det->store_resp;
    else if (!std::strcmp(name, "debug_ablation")) *value = det->dbg;
    else if (!std::strcmp(name, "tail_threads")) *value = det->tail_threads;
    else if (!std::strcmp(name, "device_tail")) *value = det->device_tail;  // (-1: by the batch's size, where available)
    else if (!std::strcmp(name, "tail_debug_band")) *value = det->tail_debug_band_mdeg;
    else if (!std::strcmp(name, "last_device_tail_frames")) *value = det->last_tail_frames;
    else if (!std::strcmp(name, "last_device_tail_fallbacks")) *value = det->last_tail_fallbacks;
    else if (!std::strcmp(name, "last_device_tail_uncertain")) *value = det->last_tail_uncertain;  // (of them: an angle inside its guard band)
    // the blur kernel's tiling of the last enqueued batch (0 before the first one)
    else if (!std::strcmp(name, "k1_rows_per_segment")) *value = a.rows_per_seg;
    else if (!std::strcmp(name, "sparse_path")) *value = det->sparse_path;
    else if (!std::strcmp(name, "last_sparse_path")) *value = det->last_sparse_path;
    else if (!std::strcmp(name, "k1_segments")) *value = a.n_segs;
    else if (!std::strcmp(name, "k1_strips")) *value = a.n_strips;
    else if (!std::strcmp(name, "k1_strip_columns")) *value = a.strip_cols;
    else return AGX_ERR_ARG;
    return AGX_OK;
    });
}

int agx_detector_sync(agx_detector *det)
{
    return agx_guard(det, [&]() -> int {
    if (!det) return AGX_ERR_ARG;
    HIP_TRY(det, hipSetDevice(det->device));
    HIP_TRY(det, hipStreamSynchronize(det->stream));
    return AGX_OK;
    });
}

static int batch_enqueue_impl(agx_detector *det, const void *d_frames, int n_frames, int width, int height,
                              size_t row_stride_bytes, size_t frame_stride_bytes, int format, void *d_saddles,
                              uint32_t saddle_capacity, void *d_frame_table)
{
    if (!det || !d_frames || n_frames <= 0) return fail(det, AGX_ERR_ARG, "null frames or n_frames <= 0");
    if (!valid_format(format)) return fail(det, AGX_ERR_FORMAT, kFormatMsg);
    if (width < 2 || height < 2) return fail(det, AGX_ERR_ARG, "width and height must be >= 2");
    if ((long long)width * height >= (1ll << 30) || width > 65000) return fail(det, AGX_ERR_ARG, "frame too large (>= 2^30 px or wider than 65000)");
    const size_t px_bytes = (size_t)bytes_per_px(format);
    if (row_stride_bytes < (size_t)width * px_bytes || row_stride_bytes > 0x7fffffffu ||
        (n_frames > 1 && frame_stride_bytes < row_stride_bytes * (size_t)height))
        return fail(det, AGX_ERR_ARG, "strides must cover a row / a frame");
    if (format == AGX_L16 && ((row_stride_bytes | frame_stride_bytes | (uintptr_t)d_frames) & 1))
        return fail(det, AGX_ERR_ARG, "16-bit pixels must be 2-byte aligned");
    if (format == AGX_LF32 && ((row_stride_bytes | frame_stride_bytes | (uintptr_t)d_frames) & 3))
        return fail(det, AGX_ERR_ARG, "f32 pixels must be 4-byte aligned");
    // rows that are not 4-byte aligned (tightly packed L8 / RGB8 of a width that is not a multiple
    // of 4, odd-width L16): the blur kernel gathers bytes instead of loading dwords
    const bool byte_rows = ((row_stride_bytes | (uintptr_t)d_frames | (n_frames > 1 ? frame_stride_bytes : 0)) & 3) != 0;
    if (n_frames > 65535) return fail(det, AGX_ERR_ARG, "at most 65535 frames per batch");
    HIP_TRY(det, hipSetDevice(det->device));
    int rc = ensure_workspace(det, n_frames, width, height);
    if (rc) return rc;
    ChainArgs &a = det->args;
    a.frames = (const uint8_t *)d_frames;
    a.frame_stride = (long long)frame_stride_bytes;
    a.row_stride = (int)row_stride_bytes;
    a.byte_rows = byte_rows ? 1 : 0;
    a.fmt = format;
    a.W = width;
    a.H = height;
    a.n_frames = n_frames;
    a.plane = (long long)width * height;
    std::memcpy(a.w, det->blur_w, sizeof(a.w));
    a.min_angle = det->params.min_saddle_angle;
    a.max_angle = det->params.max_saddle_angle;
    if (d_saddles) {
        a.out = (float *)d_saddles;
        a.out_total_cap = saddle_capacity;
        a.frame_table = (uint32_t *)d_frame_table;
        det->external_out = true;
    } else {
        // One frame (the reference's own use): the list goes straight to the pinned host mirror over PCIe -- a few KB of
        // posted writes -- and the fetch needs no second copy behind the counters' (one wait less per call).
        det->out_in_host = n_frames == 1 && det->h_out_dev != nullptr && det->h_table_dev != nullptr;
        a.out = det->out_in_host ? det->h_out_dev : det->d_out_internal;
        a.out_total_cap = (uint32_t)std::min<size_t>((size_t)n_frames * a.cap_out, 0xffffffffu);
        a.frame_table = det->out_in_host ? det->h_table_dev : nullptr;  // ... and so do its count, offset, status
        det->external_out = false;
    }
    a.force_generic = det->force_generic;
    a.dbg = det->dbg;
    // debug_ablation & 4096: wave start / end times of the sparse kernels into the slot plane (generic path's, sparsely used)
    a.wave_times = ((det->dbg & 4096) && (size_t)n_frames * (size_t)a.plane * 4 >= 3 * WAVE_TIMES_STRIDE * 16)
                       ? reinterpret_cast<unsigned long long *>(a.slot_plane) : nullptr;
    a.resp_dbg = nullptr;
    det->resp_stored = false;
    if (det->store_resp) {  // parity tests: K1 also stores the response it evaluates in registers
        const size_t need = (size_t)n_frames * (size_t)a.plane;
        if (need > det->resp_store_floats) {
            HIP_TRY(det, hipStreamSynchronize(det->stream));
            if (det->d_resp_store) (void)hipFree(det->d_resp_store);
            det->d_resp_store = nullptr;
            det->resp_store_floats = 0;
            HIP_TRY(det, hipMalloc((void **)&det->d_resp_store, need * sizeof(float)));
            det->resp_store_floats = need;
        }
        HIP_TRY(det, hipMemsetAsync(det->d_resp_store, 0, need * sizeof(float), det->stream));  // the border ring is 0
        a.resp_dbg = det->d_resp_store;
        det->resp_stored = true;
    }
    if (!plan_k1(a, det->k1_rows)) return fail(det, AGX_ERR_ARG, "unsupported frame geometry");
    return enqueue_chain(det);
}

int agx_saddles_batch_enqueue(agx_detector *det, const void *d_frames, int n_frames, int width, int height,
                              size_t row_stride_bytes, size_t frame_stride_bytes, int format)
{
    return agx_guard(det, [&]() -> int {
    return batch_enqueue_impl(det, d_frames, n_frames, width, height, row_stride_bytes, frame_stride_bytes, format,
                              nullptr, 0, nullptr);
    });
}

int agx_saddles_batch_enqueue_to(agx_detector *det, const void *d_frames, int n_frames, int width, int height,
                                 size_t row_stride_bytes, size_t frame_stride_bytes, int format, void *d_saddles,
                                 uint32_t saddle_capacity, void *d_frame_table)
{
    return agx_guard(det, [&]() -> int {
    if (!d_saddles || !d_frame_table || ((uintptr_t)d_saddles & 3) || ((uintptr_t)d_frame_table & 3))
        return fail(det, AGX_ERR_ARG, "null or misaligned output buffers");
    return batch_enqueue_impl(det, d_frames, n_frames, width, height, row_stride_bytes, frame_stride_bytes, format,
                              d_saddles, saddle_capacity, d_frame_table);
    });
}

int agx_saddles_batch_fetch(agx_detector *det, agx_saddle *out, uint32_t cap_per_frame, uint32_t *counts,
                            int *frame_status)
{
    return agx_guard(det, [&]() -> int {
    if (!det || !counts || (!out && cap_per_frame)) return fail(det, AGX_ERR_ARG, "null output");
    if (!det->enqueued) return fail(det, AGX_ERR_STATE, "no batch enqueued");
    if (det->external_out) return fail(det, AGX_ERR_STATE, "last batch wrote to caller-owned device buffers");
    HIP_TRY(det, hipSetDevice(det->device));
    const ChainArgs &a = det->args;
    const size_t F = (size_t)a.n_frames;
    bool have_counters = false;
    if (det->out_in_host) {  // one frame: list and table row are in host memory when the stream is through -- no copy at all
        HIP_TRY(det, hipStreamSynchronize(det->stream));
        const uint32_t *row = det->h_table;
        if (!(row[2] & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW))) {
            FrameCounters &c = det->h_ctr[0];
            std::memset(&c, 0, sizeof c);
            c.n_out = row[0];
            c.out_offset = row[1];
            c.flags = row[2];
            c.n_clusters = row[3];
            *det->h_total = row[0] + row[1];
            have_counters = true;
        }  // else: the full record below (sizes of the overflowing lists for the error message)
    }
    bool out_fetched = det->out_in_host;
    if (!have_counters && !det->out_in_host && det->h_table_dev && det->h_out_dev && F + 1 <= det->h_table_rows) {
        // a batch: frame rows and the compact array come over by a kernel (k_publish), one wait
        hipError_t e = (hipError_t)launch_publish(a, (uint32_t)std::min<size_t>(det->h_out_records, 0xffffffffu), det->h_table_dev, det->h_out_dev, det->stream);
        if (e != hipSuccess) return fail(det, AGX_ERR_HIP, std::string("k_publish: ") + hipGetErrorString(e));
        HIP_TRY(det, hipStreamSynchronize(det->stream));
        bool bad = false;
        for (size_t f = 0; f < F; ++f) {
            const uint32_t *row = det->h_table + 4 * f;
            FrameCounters &c = det->h_ctr[f];
            std::memset(&c, 0, sizeof c);
            c.n_out = row[0];
            c.out_offset = row[1];
            c.flags = row[2];
            c.n_clusters = row[3];
            bad = bad || (row[2] & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW));
        }
        *det->h_total = det->h_table[4 * F];
        have_counters = !bad;  // (an overflowing frame: the full records below, for the sizes in the error message)
        out_fetched = true;
    }
    if (!have_counters) {
        HIP_TRY(det, hipMemcpyAsync(det->h_ctr, a.ctr, F * sizeof(FrameCounters), hipMemcpyDeviceToHost, det->stream));
        HIP_TRY(det, hipMemcpyAsync(det->h_total, a.total_out, sizeof(uint32_t), hipMemcpyDeviceToHost, det->stream));
        HIP_TRY(det, hipStreamSynchronize(det->stream));
    }
    harvest_events(det);
    const uint32_t total = *det->h_total;
    if (total > det->h_out_records) return fail(det, AGX_ERR_HIP, "compact output counter out of range");
    if (total && !out_fetched) {
        HIP_TRY(det, hipMemcpyAsync(det->h_out, a.out, (size_t)total * 5 * sizeof(float), hipMemcpyDeviceToHost,
                                    det->stream));
        HIP_TRY(det, hipStreamSynchronize(det->stream));
    }
    int first_bad = AGX_OK;
    for (size_t f = 0; f < F; ++f) {
        const FrameCounters &c = det->h_ctr[f];
        const int st = frame_status_of(c, cap_per_frame);
        counts[f] = c.n_out;
        if (frame_status) frame_status[f] = st;
        if (st != AGX_OK) {
            if (first_bad == AGX_OK) {
                first_bad = st;
                char buf[160];
                std::snprintf(buf, sizeof buf,
                              "frame %zu: capacity exceeded (flags=0x%x seeds=%u clusters=%u candidates=%u saddles=%u)", f,
                              c.flags, c.n_seeds, c.n_clusters + c.n_clusters2, c.n_cand, c.n_out);
                det->last_error = buf;
            }
            continue;
        }
        if (c.n_out)
            std::memcpy(out + f * (size_t)cap_per_frame, det->h_out + (size_t)c.out_offset * 5,
                        (size_t)c.n_out * sizeof(agx_saddle));
    }
    return first_bad;
    });
}

}  // extern "C"

namespace {
// agx_refined_saddle_points; with want_luma8 (L16 / RGB8 only) the u8 luma of the frame is computed on the
// device behind the chain and is in det->h_luma (tight [H][W]) when this returns.
int refined_saddle_points_impl(agx_detector *det, const void *pixels, int width, int height, size_t row_stride_bytes,
                               int format, agx_saddle *out, uint32_t cap, uint32_t *n_out, bool want_luma8)
{
    if (!det || !pixels || !n_out) return fail(det, AGX_ERR_ARG, "null argument");
    if (!valid_format(format)) return fail(det, AGX_ERR_FORMAT, kFormatMsg);
    if (width < 2 || height < 2) return fail(det, AGX_ERR_ARG, "width and height must be >= 2");
    const size_t row_bytes = (size_t)width * bytes_per_px(format);
    if (row_stride_bytes < row_bytes) return fail(det, AGX_ERR_ARG, "row stride smaller than a row");
    HIP_TRY(det, hipSetDevice(det->device));
    const size_t pitch = (row_bytes + 3) & ~(size_t)3;
    const size_t need = pitch * (size_t)height;
    if (need > det->stage_bytes) {
        HIP_TRY(det, hipStreamSynchronize(det->stream));
        det->stage_bytes = 0;
        det->d_stage = static_cast<uint8_t *>(side_alloc(det, agx_detector::SB_STAGE, need, 0));
        if (!det->d_stage) return fail(det, AGX_ERR_HIP, "hipMalloc: staging buffer");
        det->stage_bytes = need;
    }
    HIP_TRY(det, hipMemcpy2DAsync(det->d_stage, pitch, pixels, row_stride_bytes, row_bytes, (size_t)height,
                                  hipMemcpyHostToDevice, det->stream));
    int rc = agx_saddles_batch_enqueue(det, det->d_stage, 1, width, height, pitch, need, format);
    if (rc) return rc;
    if (want_luma8) {  // stream-ordered behind the chain; the fetch below waits for the stream
        const size_t lb = (size_t)width * (size_t)height;
        if (lb > det->luma_bytes) {
            HIP_TRY(det, hipStreamSynchronize(det->stream));
            det->luma_bytes = 0;
            det->d_luma = static_cast<uint8_t *>(side_alloc(det, agx_detector::SB_LUMA_D, lb, 0));
            det->h_luma = static_cast<uint8_t *>(side_alloc(det, agx_detector::SB_LUMA_H, lb, 1));
            if (!det->d_luma || !det->h_luma) return fail(det, AGX_ERR_HIP, "hipMalloc / hipHostMalloc: luma planes");
            det->luma_bytes = lb;
        }
        hipError_t e = (hipError_t)launch_luma8(det->d_stage, pitch, need, 1, format, det->d_luma, width, height, det->stream);
        if (e != hipSuccess) return fail(det, AGX_ERR_HIP, std::string("k_luma8: ") + hipGetErrorString(e));
        HIP_TRY(det, hipMemcpyAsync(det->h_luma, det->d_luma, lb, hipMemcpyDeviceToHost, det->stream));
    }
    // one frame: the pinned host mirror of the batch fetch holds the list; a too-small `cap` reports
    // the needed size
    uint32_t count = 0;
    int st = AGX_OK;
    rc = agx_saddles_batch_fetch(det, nullptr, 0, &count, &st);  // cap 0: counts and status only
    if (rc && rc != AGX_ERR_CAPACITY) return rc;
    const FrameCounters &c = det->h_ctr[0];
    if (c.flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) return AGX_ERR_CAPACITY;  // last_error set by fetch
    *n_out = c.n_out;
    if (c.n_out > cap) return fail(det, AGX_ERR_CAPACITY, "output capacity too small");
    if (c.n_out) std::memcpy(out, det->h_out + (size_t)c.out_offset * 5, (size_t)c.n_out * sizeof(agx_saddle));
    det->last_error.clear();
    return AGX_OK;
}
}  // namespace

extern "C" {

int agx_refined_saddle_points(agx_detector *det, const void *pixels, int width, int height, size_t row_stride_bytes,
                              int format, agx_saddle *out, uint32_t cap, uint32_t *n_out)
{
    return agx_guard(det, [&]() -> int {
    return refined_saddle_points_impl(det, pixels, width, height, row_stride_bytes, format, out, cap, n_out, false);
    });
}

// The single frame of the last agx_refined_saddle_points call again, into a larger buffer (its
// list is still in the pinned host mirror).
static int refetch_single(agx_detector *det, agx_saddle *out, uint32_t cap, uint32_t *n_out)
{
    const FrameCounters &c = det->h_ctr[0];
    *n_out = c.n_out;
    if (c.n_out > cap) return AGX_ERR_CAPACITY;
    if (c.n_out) std::memcpy(out, det->h_out + (size_t)c.out_offset * 5, (size_t)c.n_out * sizeof(agx_saddle));
    det->last_error.clear();
    return AGX_OK;
}

int agx_debug_angle_pairs(const float *vectors, size_t n, float *exact, float *approx, uint8_t *has_approx)
{
    return agx_guard(nullptr, [&]() -> int {
    if (!vectors || !exact || !approx || !has_approx) return AGX_ERR_ARG;
    debug_angle_pairs(vectors, n, exact, approx, has_approx);
    return AGX_OK;
    });
}

int agx_debug_angle_pairs_coarse(const float *vectors, size_t n, float *coarse, uint8_t *has_coarse)
{
    return agx_guard(nullptr, [&]() -> int {
    if (!vectors || !coarse || !has_coarse) return AGX_ERR_ARG;
    debug_angle_pairs(vectors, n, nullptr, nullptr, nullptr, coarse, has_coarse);
    return AGX_OK;
    });
}

int agx_debug_white_block_angles(const float *triples, size_t n, float *reference, double *binary64)
{
    return agx_guard(nullptr, [&]() -> int {
    if (!triples || !reference || !binary64) return AGX_ERR_ARG;
    debug_white_block_angles(triples, n, reference, binary64);
    return AGX_OK;
    });
}

int agx_debug_libm_atan2f_check(uint64_t n, uint64_t seed, uint64_t *mismatches)
{
    return agx_guard(nullptr, [&]() -> int {
    if (!mismatches) return AGX_ERR_ARG;
    *mismatches = libm_atan2f_mismatches(n, seed);
    return AGX_OK;
    });
}

int agx_luma8(const void *pixels, int width, int height, size_t row_stride_bytes, int format, uint8_t *out)
{
    return agx_guard(nullptr, [&]() -> int {
    if (!pixels || !out || width <= 0 || height <= 0) return AGX_ERR_ARG;
    return luma8(pixels, width, height, row_stride_bytes, format, out);
    });
}

int agx_detect_from_saddles(const agx_detector *det, const agx_saddle *saddles, uint32_t n_saddles,
                            const uint8_t *luma, int width, int height, size_t row_stride_bytes, agx_tag *out,
                            uint32_t cap, uint32_t *n_out)
{
    return agx_guard(det, [&]() -> int {
    if (!det || !luma || !n_out || (!saddles && n_saddles) || (!out && cap)) return AGX_ERR_ARG;
    std::vector<agx_saddle> refined(saddles, saddles + n_saddles);
    std::vector<agx_tag> tags;
    detect_tail(det->fam, det->params.max_num_of_boards, std::move(refined), luma, width, height, row_stride_bytes,
                tags, det->tail_workers);
    *n_out = (uint32_t)tags.size();
    if (tags.size() > cap) return AGX_ERR_CAPACITY;
    if (!tags.empty()) std::memcpy(out, tags.data(), tags.size() * sizeof(agx_tag));
    return AGX_OK;
    });
}

int agx_detect_tail(int family, const agx_params *params, const agx_saddle *saddles, uint32_t n_saddles,
                    const uint8_t *luma, int width, int height, size_t row_stride_bytes, agx_tag *out, uint32_t cap,
                    uint32_t *n_out)
{
    return agx_guard(nullptr, [&]() -> int {
    return agx_detect_tail_threads(family, params, saddles, n_saddles, luma, width, height, row_stride_bytes, out, cap,
                                   n_out, 1);
    });
}

int agx_detect_tail_threads(int family, const agx_params *params, const agx_saddle *saddles, uint32_t n_saddles,
                            const uint8_t *luma, int width, int height, size_t row_stride_bytes, agx_tag *out,
                            uint32_t cap, uint32_t *n_out, int n_threads)
{
    return agx_guard(nullptr, [&]() -> int {
    if (!luma || !n_out || (!saddles && n_saddles) || (!out && cap) || width < 1 || height < 1) return AGX_ERR_ARG;
    FamilyInfo fam;
    if (!family_info(family, fam)) return AGX_ERR_FAMILY;
    agx_params prm;
    if (params) prm = *params;
    else agx_default_params(&prm);
    std::vector<agx_saddle> refined(saddles, saddles + n_saddles);
    std::vector<agx_tag> tags;
    struct Del {
        void operator()(TailWorkers *w) const { destroy_tail_workers(w); }
    };
    std::unique_ptr<TailWorkers, Del> workers(create_tail_workers(n_threads));  // threads of this call only (nullptr for <= 1)
    detect_tail(fam, prm.max_num_of_boards, std::move(refined), luma, width, height, row_stride_bytes, tags, workers.get());
    *n_out = (uint32_t)tags.size();
    if (tags.size() > cap) return AGX_ERR_CAPACITY;
    if (!tags.empty()) std::memcpy(out, tags.data(), tags.size() * sizeof(agx_tag));
    return AGX_OK;
    });
}

int agx_detect(agx_detector *det, const void *pixels, int width, int height, size_t row_stride_bytes, int format,
               agx_tag *out, uint32_t cap, uint32_t *n_out)
{
    return agx_guard(det, [&]() -> int {
    if (!det || !pixels || !n_out) return fail(det, AGX_ERR_ARG, "null argument");
    if (!valid_format(format)) return fail(det, AGX_ERR_FORMAT, kFormatMsg);
    if (width < 2 || height < 2) return fail(det, AGX_ERR_ARG, "width and height must be >= 2");
    if (format == AGX_LF32) return fail(det, AGX_ERR_FORMAT, "an f32 luma plane carries no u8 luma for the decode: use agx_detect_planes");
    // detector.rs:507-508: u8 luma for the decode, saddle chain on the device.  L8 is its own luma; for
    // L16 / RGB8 the conversion (the integer formulas of luma8()) runs on the device behind the chain -- the
    // frame is there anyway -- and comes back with the saddles
    const bool device_luma = format != AGX_L8;
    std::vector<agx_saddle> &saddles = det->scratch_saddles;
    if (saddles.size() < 16384) saddles.resize(16384);
    uint32_t ns = 0;
    int rc = refined_saddle_points_impl(det, pixels, width, height, row_stride_bytes, format, saddles.data(),
                                        (uint32_t)saddles.size(), &ns, device_luma);
    if (rc == AGX_ERR_CAPACITY && ns > saddles.size()) {  // longer list than ever before: the batch is still fetchable
        saddles.resize(ns);
        rc = refetch_single(det, saddles.data(), (uint32_t)saddles.size(), &ns);
    }
    if (rc) return rc;
    const uint8_t *grey = device_luma ? det->h_luma : (const uint8_t *)pixels;
    const size_t grey_stride = device_luma ? (size_t)width : row_stride_bytes;
    return agx_detect_from_saddles(det, saddles.data(), ns, grey, width, height, grey_stride, out, cap, n_out);
    });
}

int agx_detect_planes(agx_detector *det, const float *luma32f, size_t stride32f_bytes, const uint8_t *luma8,
                      size_t stride8_bytes, int width, int height, agx_tag *out, uint32_t cap, uint32_t *n_out)
{
    return agx_guard(det, [&]() -> int {
    if (!det || !luma32f || !luma8 || !n_out) return fail(det, AGX_ERR_ARG, "null argument");
    if (width < 2 || height < 2) return fail(det, AGX_ERR_ARG, "width and height must be >= 2");
    if (stride8_bytes < (size_t)width) return fail(det, AGX_ERR_ARG, "row stride smaller than a row");
    std::vector<agx_saddle> &saddles = det->scratch_saddles;
    if (saddles.size() < 16384) saddles.resize(16384);
    uint32_t ns = 0;
    int rc = agx_refined_saddle_points(det, luma32f, width, height, stride32f_bytes, AGX_LF32, saddles.data(),
                                       (uint32_t)saddles.size(), &ns);
    if (rc == AGX_ERR_CAPACITY && ns > saddles.size()) {  // longer list than ever before: the batch is still fetchable
        saddles.resize(ns);
        rc = refetch_single(det, saddles.data(), (uint32_t)saddles.size(), &ns);
    }
    if (rc) return rc;
    return agx_detect_from_saddles(det, saddles.data(), ns, luma8, width, height, stride8_bytes, out, cap, n_out);
    });
}

int agx_profile_enable(agx_detector *det, int on)
{
    if (!det) return AGX_ERR_ARG;
    det->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    return AGX_OK;
}

int agx_profile_reset(agx_detector *det)
{
    return agx_guard(det, [&]() -> int {
    if (!det) return AGX_ERR_ARG;
    HIP_TRY(det, hipSetDevice(det->device));
    HIP_TRY(det, hipStreamSynchronize(det->stream));
    harvest_events(det);
    for (int k = 0; k < K_COUNT; ++k) {
        det->prof_ms[k] = 0.0;
        det->prof_launches[k] = 0;
    }
    det->prof_batches = 0;
    return AGX_OK;
    });
}

int agx_profile_read(agx_detector *det, const char **names, double *ms_total, uint64_t *launches)
{
    return agx_guard(det, [&]() -> int {
    if (!det) return AGX_ERR_ARG;
    HIP_TRY(det, hipSetDevice(det->device));
    HIP_TRY(det, hipStreamSynchronize(det->stream));
    harvest_events(det);
    for (int k = 0; k < AGX_N_KERNELS; ++k) {  // the chain has K_COUNT launches; the entries behind them stay empty
        if (names) names[k] = k < K_COUNT ? kKernelNames[k] : nullptr;
        if (ms_total) ms_total[k] = k < K_COUNT ? det->prof_ms[k] : 0.0;
        if (launches) launches[k] = k < K_COUNT ? det->prof_launches[k] : 0;
    }
    return AGX_OK;
    });
}

int agx_detector_constants(const agx_detector *det, float *blur_w7, float *cone25, float *pmat150)
{
    return agx_guard(det, [&]() -> int {
    // det == NULL: compute them afresh (they do not depend on the family or the device)
    float w[7];
    RefineConsts rc;
    if (det) {
        std::memcpy(w, det->blur_w, sizeof w);
        rc = det->rc;
    } else {
        make_blur_weights(1.5f, w);
        make_refine_consts(rc);
    }
    if (blur_w7) std::memcpy(blur_w7, w, sizeof w);
    if (cone25) std::memcpy(cone25, rc.cone, sizeof rc.cone);
    if (pmat150) std::memcpy(pmat150, rc.pmat, sizeof rc.pmat);
    return AGX_OK;
    });
}

namespace {
// AGX_DBG_REDZONES: {buffers, damaged guard bytes, first damaged buffer, its offset (from the payload start, as int32),
// device address of buffer 0's payload (lo, hi)}.  Buffers = the chain's workspace in allocation order, then the side
// buffers that exist (staging, luma planes, the device tail's code list / tag rows / frame table) -- the last three in
// mapped pinned HOST memory, whose guards are read in place.
int fetch_redzones(agx_detector *det, void *host_out, size_t cap_bytes, size_t *n_items)
{
    *n_items = 6;
    if (cap_bytes < 6 * sizeof(uint32_t)) return AGX_ERR_CAPACITY;
    HIP_TRY(det, hipSetDevice(det->device));
    HIP_TRY(det, hipStreamSynchronize(det->stream));
    const size_t rz = det->redzone;
    const uint64_t first_payload = det->device_allocs.empty() ? 0ull : (uint64_t)(uintptr_t)((char *)det->device_allocs[0] + rz);
    uint32_t v[6] = {0u, 0u, 0xffffffffu, 0u, (uint32_t)first_payload, (uint32_t)(first_payload >> 32)};
    std::vector<uint8_t> zone(rz);
    auto check = [&](const void *base, size_t bytes, bool host) -> int {
        const uint32_t index = v[0]++;
        for (int side = 0; rz && side < 2; ++side) {
            const char *src = (const char *)base + (side ? rz + bytes : 0);
            if (host) std::memcpy(zone.data(), src, rz);
            else HIP_TRY(det, hipMemcpy(zone.data(), src, rz, hipMemcpyDeviceToHost));
            for (size_t b = 0; b < rz; ++b)
                if (zone[b] != 0xA5 && v[1]++ == 0) {
                    v[2] = index;
                    v[3] = (uint32_t)(int32_t)(side ? (long long)(bytes + b) : (long long)b - (long long)rz);
                }
        }
        return AGX_OK;
    };
    for (size_t i = 0; i < det->device_allocs.size(); ++i)
        if (int rc = check(det->device_allocs[i], det->alloc_bytes[i], false)) return rc;
    for (const agx_detector::SideBuf &b : det->side)
        if (b.base)
            if (int rc = check(b.base, b.bytes, b.host)) return rc;
    std::memcpy(host_out, v, sizeof v);
    return AGX_OK;
}
}  // namespace

int agx_debug_fetch(agx_detector *det, int frame, int what, void *host_out, size_t cap_bytes, size_t *n_items)
{
    return agx_guard(det, [&]() -> int {
    if (!det || !host_out || !n_items) return fail(det, AGX_ERR_ARG, "null argument");
    if (what == AGX_DBG_REDZONES) return fetch_redzones(det, host_out, cap_bytes, n_items);  // (no batch needed: agx_detect_batch leaves none)
    if (what == AGX_DBG_TAIL_TABLE_ADDR) {  // tests of the guard check: {host address, payload bytes} of the device tail's frame table
        *n_items = 2;
        if (cap_bytes < 2 * sizeof(uint64_t)) return AGX_ERR_CAPACITY;
        if (!det->h_tail_table) return fail(det, AGX_ERR_STATE, "no device tail has run on this handle");
        const uint64_t v[2] = {(uint64_t)(uintptr_t)det->h_tail_table, (uint64_t)det->side[agx_detector::SB_TAIL_TABLE].bytes};
        std::memcpy(host_out, v, sizeof v);
        return AGX_OK;
    }
    if (!det->enqueued) return fail(det, AGX_ERR_STATE, "no batch enqueued");
    const ChainArgs &a = det->args;
    if (frame < 0 || frame >= a.n_frames) return fail(det, AGX_ERR_ARG, "frame out of range");
    HIP_TRY(det, hipSetDevice(det->device));
    HIP_TRY(det, hipStreamSynchronize(det->stream));
    const size_t plane = (size_t)a.plane;
    FrameCounters c;
    HIP_TRY(det, hipMemcpy(&c, a.ctr + frame, sizeof c, hipMemcpyDeviceToHost));
    switch (what) {
    case AGX_DBG_BLUR: {
        *n_items = plane;
        if (cap_bytes < plane * sizeof(float)) return AGX_ERR_CAPACITY;
        HIP_TRY(det, hipMemcpy(host_out, a.blur + (size_t)frame * plane, plane * sizeof(float), hipMemcpyDeviceToHost));
        return AGX_OK;
    }
    case AGX_DBG_RESP: {
        // the response K1 evaluated in registers, stored by its "store_response" instantiation
        *n_items = plane;
        if (cap_bytes < plane * sizeof(float)) return AGX_ERR_CAPACITY;
        if (!det->resp_stored)
            return fail(det, AGX_ERR_STATE, "AGX_DBG_RESP needs option store_response=1 before the batch is enqueued");
        HIP_TRY(det, hipMemcpy(host_out, det->d_resp_store + (size_t)frame * plane, plane * sizeof(float), hipMemcpyDeviceToHost));
        return AGX_OK;
    }
    case AGX_DBG_RESP_RECOMPUTED: {
        // cross-check: the response recomputed from the stored blur plane by a separate kernel
        *n_items = plane;
        if (cap_bytes < plane * sizeof(float)) return AGX_ERR_CAPACITY;
        if (det->dbg_resp_plane < (long long)plane) {
            if (det->d_dbg_resp) (void)hipFree(det->d_dbg_resp);
            det->d_dbg_resp = nullptr;
            det->dbg_resp_plane = 0;
            HIP_TRY(det, hipMalloc((void **)&det->d_dbg_resp, plane * sizeof(float)));
            det->dbg_resp_plane = (long long)plane;
        }
        hipError_t e = (hipError_t)launch_debug_resp(a, frame, det->d_dbg_resp, det->stream);
        if (e != hipSuccess) return fail(det, AGX_ERR_HIP, std::string("k_debug_resp: ") + hipGetErrorString(e));
        HIP_TRY(det, hipStreamSynchronize(det->stream));
        HIP_TRY(det, hipMemcpy(host_out, det->d_dbg_resp, plane * sizeof(float), hipMemcpyDeviceToHost));
        return AGX_OK;
    }
    case AGX_DBG_COUNTERS: {
        // flags, seeds, second-tier seeds, clusters, generic candidates, generic roots, refined, out
        *n_items = 8;
        if (cap_bytes < 8 * sizeof(uint32_t)) return AGX_ERR_CAPACITY;
        const uint32_t v[8] = {c.flags, c.n_seeds, c.n_big, c.n_clusters + c.n_clusters2, c.n_cand, c.n_roots, c.n_refined, c.n_out};
        std::memcpy(host_out, v, sizeof v);
        return AGX_OK;
    }
    case 9: {  // AGX_DBG_LUMA8: the u8 luma the device computed for the last agx_detect on an L16 / RGB8 image
        const size_t lb = (size_t)a.W * (size_t)a.H;
        *n_items = lb;
        if (!det->h_luma || det->luma_bytes < lb) return fail(det, AGX_ERR_STATE, "no device luma: agx_detect on an L16 / RGB8 image first");
        if (cap_bytes < lb) return AGX_ERR_CAPACITY;
        std::memcpy(host_out, det->h_luma, lb);
        return AGX_OK;
    }
    case 10: {  // wave timeline (debug_ablation & 4096): `frame` selects the kernel (1 = verify, 2 = flood, 3 = refine);
                // cap_bytes / 16 records of {start, end} (100 MHz ticks), one per workgroup in launch order
        if (!a.wave_times || frame < 1 || frame > 3) return fail(det, AGX_ERR_STATE, "wave timeline needs debug_ablation & 4096 on the last batch");
        const size_t n = std::min<size_t>(cap_bytes / 16, WAVE_TIMES_STRIDE);
        *n_items = n;
        HIP_TRY(det, hipMemcpy(host_out, a.wave_times + 2 * (size_t)(frame - 1) * WAVE_TIMES_STRIDE, n * 16, hipMemcpyDeviceToHost));
        return AGX_OK;
    }
    case 7: {  // verify statistics (debug_ablation & 128), 20 x uint32
        *n_items = 20;
        if (cap_bytes < 20 * sizeof(uint32_t)) return AGX_ERR_CAPACITY;
        std::memcpy(host_out, c.stats, sizeof c.stats);
        return AGX_OK;
    }
    case AGX_DBG_MIN: {
        *n_items = 1;
        if (cap_bytes < sizeof(float)) return AGX_ERR_CAPACITY;
        uint32_t key = ~c.min_key_inv;
        uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
        std::memcpy(host_out, &u, sizeof u);
        return AGX_OK;
    }
    case AGX_DBG_CENTERS: {
        const uint32_t n = std::min(c.n_clusters + c.n_clusters2, a.cap_roots);
        *n_items = n;
        if (cap_bytes < (size_t)n * sizeof(agx_cluster_info)) return AGX_ERR_CAPACITY;
        std::vector<uint32_t> key(n), cnt(n), sx(n), sy(n);
        const size_t cb = (size_t)frame * a.cap_roots;
        HIP_TRY(det, hipMemcpy(key.data(), a.clu_key + cb, (size_t)n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(det, hipMemcpy(cnt.data(), a.clu_cnt + cb, (size_t)n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(det, hipMemcpy(sx.data(), a.clu_sx + cb, (size_t)n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(det, hipMemcpy(sy.data(), a.clu_sy + cb, (size_t)n * 4, hipMemcpyDeviceToHost));
        std::vector<agx_cluster_info> info(n);
        for (uint32_t i = 0; i < n; ++i) {
            info[i].first_index = key[i];
            info[i].size = cnt[i];
            std::memcpy(&info[i].cx, &sx[i], 4);  // K4 leaves the f32 centroid here
            std::memcpy(&info[i].cy, &sy[i], 4);
        }
        std::sort(info.begin(), info.end(),
                  [](const agx_cluster_info &p, const agx_cluster_info &q) { return p.first_index < q.first_index; });
        std::memcpy(host_out, info.data(), (size_t)n * sizeof(agx_cluster_info));
        return AGX_OK;
    }
    case AGX_DBG_REFINED: {
        const uint32_t n = c.n_refined;
        *n_items = n;
        if (cap_bytes < (size_t)n * sizeof(agx_saddle)) return AGX_ERR_CAPACITY;
        std::vector<RefinedRec> rec(n);
        HIP_TRY(det, hipMemcpy(rec.data(), a.refined + (size_t)frame * a.cap_roots, (size_t)n * sizeof(RefinedRec),
                               hipMemcpyDeviceToHost));
        std::sort(rec.begin(), rec.end(), [](const RefinedRec &p, const RefinedRec &q) { return p.key < q.key; });
        agx_saddle *o = (agx_saddle *)host_out;
        for (uint32_t i = 0; i < n; ++i) o[i] = {rec[i].x, rec[i].y, rec[i].k, rec[i].theta, rec[i].phi};
        return AGX_OK;
    }
    default: return fail(det, AGX_ERR_ARG, "unknown debug item");
    }
    });
}

}  // extern "C"
