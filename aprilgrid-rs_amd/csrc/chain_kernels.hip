// chain_kernels.hip -- the per-pixel saddle chain of aprilgrid's
// TagDetector::refined_saddle_points (reference src/detector.rs:408-446) as hand-written
// HIP for gfx950 (MI355X, wave64).
//
// Bit-exactness contract: every f32 operation below is performed in the reference's order
// with one rounding per operation.  This file MUST be compiled with -ffp-contract=off and
// without fast-math; the only fused operations are the explicit __builtin_fmaf calls in
// div_const(), which are an exact (exhaustively tested) replacement for a division.
//
//   K1 k_blur_hessian   luma->f32 (image crate to_luma32f), 7-tap separable Gaussian
//                       (image_util.rs:110-206), Hessian determinant (image_util.rs:72-109),
//                       per-frame min (detector.rs:414-417).  Every WAVE is autonomous: it owns a
//                       strip of <= 248 columns (4 pixels per lane, neighbours by DPP) and
//                       marches down a segment of rows with the 7-row vertical window and the
//                       3-row Hessian window in registers.  No LDS exchange, no barriers in the
//                       row loop.  The response is NOT stored.  K1 also writes a candidate
//                       SUPERSET as a transposed 1 bit / pixel mask: resp < 0.05*m for a running
//                       minimum m >= min_frame.
//   K2 k_verify_seeds   the exact threshold resp < 0.05*min_frame (detector.rs:418,177) at the set
//                       bits that can still fail (response recomputed from the blur plane) and the
//                       flood seeds, in one pass over the mask
//   K3 k_flood_refine   4-connected components (image_util.rs:208-236) + centroid sums
//                       (detector.rs:421-429): one component per lane, bit-parallel flood fill
//                       of a 32x32 window of the mask held in registers; oversized components
//                       by the whole wave in a 128x64 window; then rochade_refine
//                       (detector.rs:194-361) of the cluster by the lane that flooded it
//   K4 k_rare           per frame (1024 threads): the k/phi filter (detector.rs:436-445) and the
//                       emission of the frame's saddles in reference order; before that, for frames
//                       where a component leaves the flood windows, the generic clustering fallback
//                       (mask -> candidate list -> lock-free union-find -> sums) and its refinement;
//                       the large-list sort (> 1024 refined records per frame)
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>

#include "chain_kernels.h"

namespace agx {

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_order_key(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t key)
{
    uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
    return __uint_as_float(u);
}

// v / D for an integer-valued v in [0, D], D = 255 or 65535: q = RN(v*r) refined by two
// fused steps is the correctly rounded quotient for every such v (tests/test_gpu_parity.py
// checks all 256 / 65536 inputs against the oracle's true division).
template <int D>
__device__ __forceinline__ float div_const(float v)
{
    const float r = 1.0f / (float)D;  // constant-folded, correctly rounded
    float q = v * r;
    float e = __builtin_fmaf(-q, (float)D, v);
    return __builtin_fmaf(e, r, q);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroups go to the 8 XCDs round-robin by their linear id, and the sparse kernels use only the
// first few slots of a frame (as many as it has work for).  A (slots, frames) grid with an x
// extent that is a multiple of 8 would park the idle slots on the same XCDs for every frame
// (refine: 0.096 ms at 16 slots per frame, 0.072 ms at 15).  These kernels therefore take a 1-D
// grid in slot-major order: linear id -> (slot = id / n_frames, frame = id % n_frames).  Slot s of
// all frames is dispatched before slot s+1, a frame stays on one XCD, and the slots without work
// come last and exit at once.
struct FrameSlot {
    int frame;
    uint32_t slot, n_slots;
};
__device__ __forceinline__ FrameSlot frame_slot(int n_frames, bool newest_first)
{
    FrameSlot fs;
    fs.slot = blockIdx.x / (uint32_t)n_frames;
    const int f = (int)(blockIdx.x - fs.slot * (uint32_t)n_frames);
    fs.frame = newest_first ? n_frames - 1 - f : f;
    fs.n_slots = gridDim.x / (uint32_t)n_frames;
    return fs;
}

// Neighbour-lane exchange by DPP wave shifts (one VALU op, no LDS): lane l receives the value
// of lane l-1 (from_left) or l+1 (from_right); lane 0 / lane 63 receive 0 (bound_ctrl: no
// destination to initialise).
__device__ __forceinline__ float from_left(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ uint32_t from_left_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t from_right_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

// The lane id recomputed where it is used (2 VALU): values that are needed once every few rows are derived
// from it on the spot instead of living in registers across K1's row loop (the kernel sits exactly at
// the 5-waves-per-SIMD register limit, and the compiler hoists anything loop-invariant).
// OR (minimum) of a value over the wave, wave-uniform result: four row shifts, two row broadcasts and a readlane on the
// vector ALU.  (The __shfl_xor butterfly is six ds_bpermute round trips through LDS, ~100 cycles each for a wave that has
// nothing else to do meanwhile: k_verify_seeds reduces up to five words per tile.)
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1 (lanes without a source: 0)
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8  -> lane 15 of a row: the row
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1, 3
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);  // row_bcast:31 into rows 2, 3 -> lane 63: the wave
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Inclusive prefix sum over the wave's lanes (same six DPP steps).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // inclusive within each row of 16
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // rows 1, 3 += the row before
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);  // rows 2, 3 += rows 0 + 1
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_min_step(float f)
{
    const float inf = __builtin_inff();  // lanes without a source, rows outside the mask: their own value stands
    return fminf(f, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, inf), __builtin_bit_cast(int, f), CTRL, ROW_MASK, 0xf, false)));
}
__device__ __forceinline__ float wave_min_f32(float f)
{
    f = dpp_min_step<0x111, 0xf>(f);
    f = dpp_min_step<0x112, 0xf>(f);
    f = dpp_min_step<0x114, 0xf>(f);
    f = dpp_min_step<0x118, 0xf>(f);
    f = dpp_min_step<0x142, 0xa>(f);
    f = dpp_min_step<0x143, 0xc>(f);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), 63));
}

__device__ __forceinline__ int lane_id_here()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Build option (-DAGX_K1_DEFER_ROWS=10; OFF by default): K1 defers the threshold decision of the first rows of
// every segment.  A wave that starts with nothing published admits every negative response of its first rows
// (its running threshold is 0.05 x the minimum of one or two rows), and K2 re-tests all of those from the blur
// plane.  With the option the responses of those rows are kept in LDS as bf16 (truncated: for a negative value an
// upper bound; with the low half set, a lower bound) and the rows' mask bits are decided again when the segment's
// first TWO word rows are complete, still as a superset: a bit stays if the lower bound is below the running
// threshold, and the block's weakest admitted candidate (cand_max) takes the upper bounds of the bits that stay.
// Measured (round 3, 256 x 1280x800, 8 alternating runs per build on one box): K2's re-tests 5 760 -> 2 870 per
// frame, K2 64 -> 61 us -- but K1 300 -> 307 us (29 KB of LDS per workgroup and the extra branches in the row
// loop), a net loss; the parity suite passes with it on.
#ifndef AGX_K1_SYNC_GAP_MAX
#define AGX_K1_SYNC_GAP_MAX 16  // rows between two refreshes of the running threshold at most (128: K2 +5 us, frames whose strips see no corner for long: 1.5x the re-tests)
#endif
#ifndef AGX_K1_SYNC_GAP_LATE
#define AGX_K1_SYNC_GAP_LATE 64  // ... for waves that find a published minimum when they start
#endif
#ifndef AGX_K1_WPE_MAX
#define AGX_K1_WPE_MAX 8  // A/B builds: -DAGX_K1_WPE_MAX=4 holds K1 at four waves per SIMD
#endif
#ifndef AGX_K1_WPE
#define AGX_K1_WPE 1  // A/B builds: -DAGX_K1_WPE=0 leaves the register budget of K1 to the compiler
#endif
#ifndef AGX_K1_DEFER_ROWS
#define AGX_K1_DEFER_ROWS 0  // off by default (measured: see the note at K1_DEFER_ROWS); A/B builds: -DAGX_K1_DEFER_ROWS=10
#endif
constexpr int K1_DEFER_ROWS = AGX_K1_DEFER_ROWS > 0 ? AGX_K1_DEFER_ROWS : 1;

// debug_ablation & 4096: start / end time of a wave (see ChainArgs::wave_times); the destructor runs on
// every exit path of the kernel.
struct WaveTimer {
    unsigned long long *rec;
    unsigned long long t0;
    __device__ __forceinline__ WaveTimer(const ChainArgs &a, int kernel, bool on = true)
        : rec(on && a.wave_times ? a.wave_times + 2 * ((size_t)(kernel - 1) * WAVE_TIMES_STRIDE + (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) : nullptr), t0(0)
    {
        if (rec) t0 = wall_clock64();
    }
    __device__ __forceinline__ ~WaveTimer()
    {
        if (rec && (threadIdx.x & 63) == 0 && (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) < WAVE_TIMES_STRIDE) {
            rec[0] = t0;
            rec[1] = wall_clock64();
        }
    }
};

// Raw pixel words of 4 consecutive pixels: L8 1 dword, L16 2 dwords, RGB8 3 dwords, LF32 4 dwords.
template <int FMT>
struct RawPx {
    static constexpr int BPP = FMT == 0 ? 1 : (FMT == 1 ? 2 : (FMT == 2 ? 3 : 4));
    uint32_t d[BPP];
};

template <int FMT, bool A4>
__device__ __forceinline__ RawPx<FMT> load_raw(const uint8_t *__restrict__ rowp, int c0, int W, bool byte_rows)
{
    constexpr int BPP = RawPx<FMT>::BPP;
    RawPx<FMT> r;
    if (FMT == 0 && A4) {  // A4: W is a multiple of 4
        // always exactly one dword load: lanes left / right of the image read the first / last
        // dword of the row and replicate its edge pixel (clamp-to-edge)
        const int cc = c0 < 0 ? 0 : (c0 > W - 4 ? W - 4 : c0);
        const uint32_t d = *reinterpret_cast<const uint32_t *>(rowp + cc);
        const uint32_t first = (d & 0xffu) * 0x01010101u, last = (d >> 24) * 0x01010101u;
        r.d[0] = c0 < 0 ? first : (c0 >= W ? last : d);
        return r;
    }
    if (c0 >= 0 && c0 + 3 < W && !byte_rows) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(rowp + (size_t)c0 * BPP);
#pragma unroll
        for (int i = 0; i < BPP; ++i) r.d[i] = p[i];
    } else {
        // lanes reaching over the left / right image edge: clamp-to-edge, byte gather
#pragma unroll
        for (int i = 0; i < BPP; ++i) r.d[i] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = c0 + j;
            c = c < W - 1 ? c : W - 1;
            c = c > 0 ? c : 0;
#pragma unroll
            for (int b = 0; b < BPP; ++b) {
                uint32_t byte = rowp[(size_t)c * BPP + b];
                int pos = j * BPP + b;
                r.d[pos >> 2] |= byte << ((pos & 3) * 8);
            }
        }
    }
    return r;
}

// image 0.25.9 to_luma32f (reference call site src/detector.rs:409), SURVEY.md App. B
template <int FMT>
__device__ __forceinline__ void convert_px(const RawPx<FMT> &r, float m[4])
{
    if (FMT == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = div_const<255>((float)((r.d[0] >> (8 * j)) & 0xffu));
    } else if (FMT == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            m[j] = div_const<65535>((float)((r.d[j >> 1] >> (16 * (j & 1))) & 0xffffu));
    } else if (FMT == 3) {  // the caller's own to_luma32f plane
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = __uint_as_float(r.d[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t c[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                int pos = j * 3 + b;
                c[b] = (r.d[pos >> 2] >> ((pos & 3) * 8)) & 0xffu;
            }
            uint32_t l = (2126u * c[0] + 7152u * c[1] + 722u * c[2]) / 10000u;
            m[j] = div_const<255>((float)l);
        }
    }
}

// 8-bit luma of pixel j of the raw words (L8: the byte; RGB8: the integer luma of the image crate)
template <int FMT>
__device__ __forceinline__ uint32_t luma_byte(const RawPx<FMT> &r, int j)
{
    if (FMT != 2) return (r.d[0] >> (8 * j)) & 0xffu;
    uint32_t c[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int pos = j * 3 + b;
        c[b] = (r.d[pos >> 2] >> ((pos & 3) * 8)) & 0xffu;
    }
    return (2126u * c[0] + 7152u * c[1] + 722u * c[2]) / 10000u;
}

// ------------------------------------------------------------------------------------------
// K1: every wave is autonomous.  A wave owns a strip of up to 248 columns: lane l holds the 4
// pixels of columns xs-4+4l .. xs-1+4l of the current row, lanes 0 and n+1 are halo lanes that
// only feed their neighbours.  The wave marches down its row segment; per row it converts 4
// pixels, pulls the 3+3 neighbouring pixels from the adjacent lanes by DPP, does the 7-tap
// horizontal pass, pushes the result into a 7-row register window for the vertical pass,
// stores the blur row, and evaluates the Hessian determinant of the previous row (its left /
// right blur neighbours again by DPP) for the per-frame minimum.  No LDS exchange and no barriers in
// the row loop; 8-bit formats read their tap products from an LDS table built once per workgroup.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int H_full_segs(const ChainArgs &a) { return a.H / a.rows_per_seg; }

// RESP: debug instantiation (agx_detector_set_option "store_response") that also stores the
// determinant this kernel evaluates in registers, for the parity tests (AGX_DBG_RESP).
#ifndef AGX_K1_VALU_MASK
#define AGX_K1_VALU_MASK 1  // the row's compare / mask / weakest-candidate block on the vector ALU alone (0: through the scalar unit)
#endif
#ifndef AGX_BLUR_STORE_AUX
#define AGX_BLUR_STORE_AUX 2  // cache policy bits of the blur plane's buffer stores: 2 = nt (the plane is written once and read
// sparsely two launches later; kept out of the caches' way, the sparse kernels' misses do not have to evict it first)
#endif
#ifndef AGX_IN_LOAD_AUX
#define AGX_IN_LOAD_AUX 0  // cache policy bits of the frame's buffer loads
#endif
// UF (round 5; with A4 false): frames whose width is not a multiple of 4 or whose rows / base are not 4-byte aligned keep the
// aligned form's loads and tap table -- one (unaligned) 4 / 8 / 12 / 16-byte load per lane and row through the buffer resource
// (gfx9 in the HSA runtime's unaligned access mode), the lane's four pixels picked out of it by per-lane selectors that also
// replicate the edge pixel (clamp-to-edge) -- instead of gathering bytes row by row: the generic form ran 3.4 - 4.8 x slower per
// pixel than the aligned one (profiles/r5_k1_unaligned_*).  The generic form remains for widths below 4 and frames beyond the
// 32-bit buffer offsets.
template <int FMT, bool A4, bool RESP = false, bool UF = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(((A4 || UF) && !RESP && AGX_K1_WPE) ? ((FMT == 0 || (FMT == 1 && !UF)) ? (AGX_K1_WPE_MAX < 5 ? AGX_K1_WPE_MAX : 5) : 4) : 1, AGX_K1_WPE_MAX))) k_blur_hessian(ChainArgs a)
{
    static_assert(!UF || !A4, "UF is the aligned form's data path for frames that are not aligned");
    const int lane = threadIdx.x & 63;
    // u8 -> the pixel's four distinct tap products of the horizontal pass: entry v holds
    // (v/255)*w0 .. (v/255)*w3 (true division and the same multiplications the pass would do, once
    // per entry; w4..w6 equal w2..w0 bit for bit).  One 16-byte LDS read per pixel replaces the
    // conversion AND the pass's 7 multiplications per output pixel; the LDS pipe is otherwise idle.
    __shared__ float4 s_lut4[256];
    // deferred rows (see K1_DEFER_ROWS): per wave, the rows' responses as 4 x bf16 per lane, and the first word
    // row of the segment (4 mask words + the block's cand_max) until it is decided.  Arbitrary f32 planes
    // (FMT 3) may hold infinities, whose truncated bounds are not ordered: they keep the direct decision.
    constexpr bool DEFER = FMT != 3 && AGX_K1_DEFER_ROWS > 0;
    __shared__ uint2 s_def[DEFER ? 4 : 1][K1_DEFER_ROWS][64];
    __shared__ uint32_t s_blk0[DEFER ? 4 : 1][5][64];
    if (FMT == 0 || FMT == 2) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) {
            const float f = (float)i / 255.0f;
            s_lut4[i] = make_float4(f * a.w[0], f * a.w[1], f * a.w[2], f * a.w[3]);
        }
        __syncthreads();
    }
    // wave-uniform quantities are made scalar explicitly (the compiler cannot prove it)
    // Work units (frame, strip, segment) in dispatch order: first every full-height segment of
    // every frame, then the short last segments (H not a multiple of the segment height) -- the
    // short ones fill the slots that free up while the last full ones are still running, instead
    // of leaving a thin extra round at the end.
    const int u = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int n_full = H_full_segs(a);  // segments of full height per frame
    const int per_frame_full = a.n_strips * n_full;
    const int total_full = per_frame_full * a.n_frames;
    int frame, strip, seg;
    if (a.k1_group > 0 && a.k1_group < a.n_frames) {
        // Frame groups: the batch's frames in groups of k1_group, group after group; inside a group segment-major,
        // the full segments from the middle outwards, the short last segment behind them.  The frames of a group are
        // complete when the group's waves are, long before the launch ends: the sparse stages of those frames can run
        // under the blur of the groups that follow.
        const int per_frame = a.n_strips * a.n_segs;
        const int per_group = per_frame * a.k1_group;
        if (u >= per_frame * a.n_frames) return;  // whole wave: padding of the last workgroup
        const int g = u / per_group;
        const int gbase = g * a.k1_group;
        const int ng = min(a.k1_group, a.n_frames - gbase);
        const int ur = u - g * per_group;
        const int per_seg = a.n_strips * ng;
        const int k = ur / per_seg;
        const int r = ur - k * per_seg;
        const int mid = (n_full - 1) >> 1, dist = (k + 1) >> 1;
        seg = k >= n_full ? k : ((k & 1) ? mid + dist : mid - dist);
        frame = gbase + r / a.n_strips;
        strip = r - (r / a.n_strips) * a.n_strips;
    } else if (u < total_full) {
        if (a.dbg & 1024) {  // A/B: the former frame-major order
            frame = u / per_frame_full;
            const int r = u - frame * per_frame_full;
            seg = r / a.n_strips;
            strip = r - seg * a.n_strips;
        } else {
            // segment-major over the whole batch: segment s of every frame before segment s+1 of any.
            // The chip holds about half of the batch's waves at a time, so the later segments of a
            // frame start when its earlier ones have published their minima: they threshold against a
            // nearly final value from their first row on (a wave that starts with nothing known admits
            // every noise-level negative response of its first rows, all of which K2 has to re-test).
            const int per_seg = a.n_strips * a.n_frames;
            const int k = u / per_seg;
            const int r = u - k * per_seg;
            // ... and the segments from the middle of the frame outwards (3, 4, 2, 5, 1, 6, 0 for seven): the waves
            // of the first round all start with nothing known, and the rows most likely to show a strong corner
            // early -- whose minima then tighten everyone's threshold -- are not the frame's top rows
            const int mid = (n_full - 1) >> 1, dist = (k + 1) >> 1;
            seg = (a.dbg & 32768) ? k : ((k & 1) ? mid + dist : mid - dist);  // (32768: A/B, ascending)
            frame = r / a.n_strips;
            strip = r - frame * a.n_strips;
        }
    } else {
        const int v = u - total_full;
        if (n_full == a.n_segs || v >= a.n_strips * a.n_frames) return;  // whole wave: padding of the last workgroup
        frame = v / a.n_strips;
        strip = v - frame * a.n_strips;
        seg = n_full;
    }
    const int W = a.W, H = a.H;
    const int xs = strip * a.strip_cols;
    const int xe = min(W, xs + a.strip_cols);
    const int c0 = xs - 4 + 4 * lane;
    const bool lane_valid = (lane >= 1) && (c0 < xe);
    const int ys = seg * a.rows_per_seg;
    const int ye = min(H, ys + a.rows_per_seg);

    const uint8_t *fbase = a.frames + (size_t)frame * (size_t)a.frame_stride;
    float *blur_f = a.blur + (size_t)frame * (size_t)a.plane;
    const int wv = DEFER ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;  // this wave's part of s_def / s_blk0
    const float w0 = a.w[0], w1 = a.w[1], w2 = a.w[2], w3 = a.w[3], w4 = a.w[4], w5 = a.w[5],
                w6 = a.w[6];

    // Vertical pass as 7 running sums: acc[s] is the partial sum of one blur row in flight.  An
    // arriving horizontally-blurred row h is tap 0 of blur row r+3, tap 1 of row r+2, ... tap 6
    // of row r-3, so every sum receives its taps in index order exactly as image_util.rs:192-201
    // adds them; w[i] == w[6-i] bit for bit (image_util.rs:116-124 evaluates exp(-(x*x)/..) for
    // x = -3..3), so the 7 products per pixel are only 4 distinct ones.  The row loop is
    // unrolled by 7 so that the slot of every sum is a compile-time register.
    float acc[7][4];
    // blur rows b-2, b-1 (and b below) with their edge columns.  The lane's own four values are
    // one vector value, so that the row store can take them from the register quad they live in.
    // The two values beyond the lane's four columns are not kept: they are fetched from the
    // neighbouring lane where they are used (each is used once, so the DPP shift folds into the
    // consuming subtraction / addition).
    struct Row6 {
        f32x4 c;
        __device__ __forceinline__ float operator[](int i) const
        {
            return i == 0 ? from_left(c[3]) : (i == 5 ? from_right(c[0]) : c[i - 1]);
        }
    };
    Row6 up, mid;
#pragma unroll
    for (int s7 = 0; s7 < 7; ++s7)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[s7][j] = 0.0f;
    up.c = mid.c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // Running minimum per pixel column of the lane (a column either always or never takes part in
    // the frame's minimum, so validity is applied when the columns are combined, not per row).
    // A4: columns 1 and 2 share rmin[1].  The frame always contains its zero border ring -> 0.
    float rmin[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // Candidate superset: a pixel can only end up below the final threshold 0.05*min_frame if it
    // is below 0.05*m for every m >= min_frame; m = the smallest response this frame has shown so
    // far (own rows + what other waves published in ctr.min_key_inv).  thr_run only ever moves
    // down towards the final threshold, so nothing is lost; the list is filtered again in K2.
    FrameCounters &ctr = a.ctr[frame];
    uint32_t *mask_f = a.mask + (size_t)frame * (size_t)a.mask_plane;
    // wave-uniform; kept as bit patterns so that they live in scalar registers
    int thr_run_bits = 0, published_bits = 0;  // 0.0f
    // VALU_MASK: the row's candidate bits without the scalar unit in the chain (see the row's compare block): the running
    // threshold per column of the lane, -inf for columns that are not its own (halo lanes, the image's border ring)
    constexpr bool VALU_MASK = AGX_K1_VALU_MASK != 0 && !(FMT != 3 && AGX_K1_DEFER_ROWS > 0);
    float thr_v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int rows_to_sync = 0, sync_gap = 1, gap_cap = AGX_K1_SYNC_GAP_MAX;
    // ctr.min_key_inv as fetched at the previous sync point.  The first fetch is issued right here, at
    // the start of the wave, and awaited at the first sync point seven warm-up rows later: a wave that
    // starts when other waves of its frame have already published (the later segments under the
    // segment-major dispatch order) thresholds against the frame's running minimum from its first row
    // on instead of against the minimum of that one row.
    // The poll is a vector load (every lane the same word, one request) whose register is read at the NEXT sync point:
    // the compiler's own counted wait sits in front of that read, rows later, and the wave computes in between.  (As a
    // scalar load it had to be awaited where it was issued -- an in-flight SGPR cannot be kept from the compiler -- and
    // with one frame on the chip every wave of the frame polls and publishes to the same L2 line at the same moment:
    // ~2.5 us per poll, six polls per 32-row segment, 15 of K1's 45 us at 1280x800.)
    // Batches that fill the chip keep the scalar load (a.k1_async_poll == 0, plan_k1): four other waves per SIMD cover the
    // wait, and the vector load's counted wait would drain the blur stores at every sync point (K1 +4 % at 256 frames).
    auto poll_min = [&]() { return __hip_atomic_load(&ctr.min_key_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const bool async_poll = a.k1_async_poll != 0;  // wave-uniform
    uint32_t polled_v = 0u;
    uint32_t polled = 0u;  // the last value polled (wave-uniform)
    if (async_poll) polled_v = poll_min();
    else asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(polled) : "s"(&ctr.min_key_inv) : "memory");
    bool first_sync = true;
    float cmax = -__builtin_inff();  // weakest candidate response of this lane in the current 32-row block
    uint32_t mw[4] = {0u, 0u, 0u, 0u};  // this lane's 4 mask words (4 columns x 32 rows) in progress
    int y_pushed = 0;                    // last row whose bits were shifted into mw

    // per-pixel "takes part in the min" (interior column of this lane's strip)
    bool min_ok[4];
    uint64_t ok_mask[4];  // the same as wave-wide lane masks
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        min_ok[j] = lane_valid && (c0 + j > 0) && (c0 + j < W - 1);
        ok_mask[j] = __ballot(min_ok[j]);
    }
    if (VALU_MASK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) thr_v[j] = min_ok[j] ? 0.0f : -__builtin_inff();
    }
    auto lane_min = [&]() {
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(A4 && j == 2)) v = fminf(v, min_ok[j] ? rmin[j] : 0.0f);
        return v;
    };
    const bool store_ok = lane_valid && !(a.dbg & 1);

    const int r0 = ys - 4, r1 = ye + 3;
    auto rowptr = [&](int r) {
        int rr = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
        return fbase + (size_t)rr * (size_t)a.row_stride;
    };
    // Input prefetch.  L8 frames with W % 4 == 0 (FAST): the 7 input dwords of a 7-row body are
    // loaded together one body ahead.  At the top of a body they are copied into working
    // registers -- the only place the wave waits for memory -- and the next body's 7 loads are
    // issued at once.  gfx9 counts loads and stores in one in-order vmcnt and the compiler waits
    // with vmcnt(0) here, i.e. it also drains the blur stores issued so far; batching makes that
    // happen once per 7 rows, on loads that are a whole body old, instead of before every row.
    // FAST: 8-bit luma (L8, RGB8) and W % 4 == 0.  RGB8 rows arrive as 3 dwords per lane and are
    // reduced to the packed luma dword of the image crate's integer formula when the group starts;
    // from there on the two formats share everything.
    constexpr bool FAST = (FMT == 0 || FMT == 2) && (A4 || UF);
    // BUF: W % 4 == 0, any format -- the rows of a group are fetched together through the buffer
    // resource one group ahead (L16 keeps its raw 2 dwords per row and converts row by row).
    constexpr bool BUF = A4 || UF;
    constexpr int RW = FMT == 2 ? 3 : (FMT == 1 ? 2 : (FMT == 3 ? 4 : 1));  // input dwords per lane and row
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    uint32_t ring[7][RW];
    RawPx<FMT> raw_a, raw_b, raw_c, raw_d;
    const int cc = c0 < 0 ? 0 : (c0 > W - 4 ? W - 4 : c0);  // FAST: clamped column of this lane's dword
    // byte selector of v_perm_b32: identity, or the first / last byte of the dword four times
    // (UF: W is any width >= 4 -- pixel j of the lane is column clamp(c0 + j, 0, W - 1), byte (that column - cc) of the dword at cc;
    // the lane that straddles the right edge repeats the last pixel: e.g. W = 1282, c0 = 1280 -> dword at 1278, selector 0x03030302)
    uint32_t edge_sel = c0 < 0 ? 0x00000000u : (c0 >= W ? 0x03030303u : 0x03020100u);
    int src_px[4] = {0, 1, 2, 3};  // UF: which of the four loaded pixels (columns cc .. cc + 3) the lane's pixel j is
    if (UF) {
        edge_sel = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int col = c0 + j;
            col = col < 0 ? 0 : (col > W - 1 ? W - 1 : col);
            src_px[j] = col - cc;
            edge_sel |= (uint32_t)(col - cc) << (8 * j);
        }
    }
    // A4 kernels address the frame and its blur plane through raw buffer resources: the base is
    // wave-uniform (SGPRs), the lane's column offset is one constant VGPR and the row offset an
    // SGPR, so a row's load / store costs no address arithmetic on the vector ALU.  Range rule
    // (checked by tools/ubench/buffer_oob_test.hip): an access with voffset >= num_records - soffset
    // is dropped -- a store with soffset == num_records goes nowhere.
    constexpr uint32_t RSRC_WORD3 = 0x00020000u;  // raw 32-bit buffer, gfx9
    const uint32_t blur_bytes = (uint32_t)a.plane * 4u;
    __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)fbase, 0, (int)((uint32_t)H * (uint32_t)a.row_stride), RSRC_WORD3);
    __amdgpu_buffer_rsrc_t rs_blur = __builtin_amdgcn_make_buffer_rsrc((void *)blur_f, 0, (int)blur_bytes, RSRC_WORD3);
    auto issue_load = [&](int r, uint32_t (&dst)[RW]) {
        const int rr = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
        if (FMT == 3) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, cc * 4, rr * a.row_stride, AGX_IN_LOAD_AUX);
            dst[0] = v.x;
            dst[RW > 1 ? 1 : 0] = v.y;
            dst[RW > 2 ? 2 : 0] = v.z;
            dst[RW > 3 ? 3 : 0] = v.w;
        } else if (FMT == 2) {
            const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs_in, cc * 3, rr * a.row_stride, AGX_IN_LOAD_AUX);
            dst[0] = v.x;
            dst[RW > 1 ? 1 : 0] = v.y;
            dst[RW > 2 ? 2 : 0] = v.z;
        } else if (FMT == 1) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_in, cc * 2, rr * a.row_stride, AGX_IN_LOAD_AUX);
            dst[0] = v.x;
            dst[RW > 1 ? 1 : 0] = v.y;
        } else {
            dst[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_in, cc, rr * a.row_stride, AGX_IN_LOAD_AUX);
        }
    };
    // L16 edge lanes: 16-bit selectors of v_perm_b32 over (dword1:dword0) -- identity, or the first /
    // last pixel of the 8 bytes in both halves
    uint32_t sel16_lo = c0 < 0 ? 0x01000100u : (c0 >= W ? 0x07060706u : 0x03020100u);
    uint32_t sel16_hi = c0 < 0 ? 0x01000100u : (c0 >= W ? 0x07060706u : 0x07060504u);
    if (UF) {  // pixel j = halfword src_px[j] of the eight bytes
        sel16_lo = (uint32_t)(2 * src_px[0]) | ((uint32_t)(2 * src_px[0] + 1) << 8) | ((uint32_t)(2 * src_px[1]) << 16) | ((uint32_t)(2 * src_px[1] + 1) << 24);
        sel16_hi = (uint32_t)(2 * src_px[2]) | ((uint32_t)(2 * src_px[2] + 1) << 8) | ((uint32_t)(2 * src_px[3]) << 16) | ((uint32_t)(2 * src_px[3] + 1) << 24);
    }
    if (BUF) {
#pragma unroll
        for (int k = 0; k < 7; ++k) issue_load(r0 + k, ring[k]);
    } else {
        raw_a = load_raw<FMT, A4>(rowptr(r0), c0, W, a.byte_rows);
        raw_b = load_raw<FMT, A4>(rowptr(r0 + 1), c0, W, a.byte_rows);
        raw_c = load_raw<FMT, A4>(rowptr(r0 + 2), c0, W, a.byte_rows);
        raw_d = load_raw<FMT, A4>(rowptr(r0 + 3), c0, W, a.byte_rows);
    }

#pragma unroll 1
    for (int rbase = r0; rbase <= r1; rbase += 7) {
        uint32_t cur[7];  // packed 8-bit luma of the lane's four pixels, per row of the group
        uint32_t got[7][RW];
        if (BUF) {
#pragma unroll
            for (int k = 0; k < 7; ++k)
#pragma unroll
                for (int q = 0; q < RW; ++q) got[k][q] = ring[k][q];
#pragma unroll
            for (int k = 0; k < 7; ++k) issue_load(rbase + 7 + k, ring[k]);  // rows past the end clamp to H-1
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (FMT == 2) {
                    RawPx<FMT> px;
#pragma unroll
                    for (int q = 0; q < RW; ++q) px.d[q] = got[k][q];
                    cur[k] = luma_byte<FMT>(px, 0) | (luma_byte<FMT>(px, 1) << 8) | (luma_byte<FMT>(px, 2) << 16) |
                             (luma_byte<FMT>(px, 3) << 24);
                } else if (FMT == 0) {
                    cur[k] = got[k][0];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int r = rbase + k;
            if (r > r1) break;  // wave-uniform
            float m[4];
            float4 P[4];  // FAST: tap products of the lane's own four pixels
            if (FAST) {
                // lanes left / right of the image replicate the edge pixel (clamp-to-edge)
                const uint32_t dd = __builtin_amdgcn_perm(cur[k], cur[k], edge_sel);
#pragma unroll
                for (int j = 0; j < 4; ++j) P[j] = s_lut4[(dd >> (8 * j)) & 0xffu];
            } else if (BUF && FMT == 3) {  // LF32, aligned: lanes left / right of the image replicate the edge pixel
                const uint32_t first = got[k][0], last = got[k][RW > 3 ? 3 : 0];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (UF) {
                        const int q = src_px[j];
                        m[j] = __uint_as_float(q == 0 ? got[k][0] : (q == 1 ? got[k][RW > 1 ? 1 : 0] : (q == 2 ? got[k][RW > 2 ? 2 : 0] : got[k][RW > 3 ? 3 : 0])));
                    } else {
                        m[j] = __uint_as_float(c0 < 0 ? first : (c0 >= W ? last : got[k][RW > j ? j : 0]));
                    }
                }
            } else if (BUF) {  // L16, aligned: 4 x u16 in two dwords, edge lanes replicate by permute
                const uint32_t d0 = got[k][0], d1 = got[k][RW > 1 ? 1 : 0];
                const uint32_t e0 = __builtin_amdgcn_perm(d1, d0, sel16_lo), e1 = __builtin_amdgcn_perm(d1, d0, sel16_hi);
                m[0] = div_const<65535>((float)(e0 & 0xffffu));
                m[1] = div_const<65535>((float)(e0 >> 16));
                m[2] = div_const<65535>((float)(e1 & 0xffffu));
                m[3] = div_const<65535>((float)(e1 >> 16));
            } else {
                convert_px<FMT>(raw_a, m);
                raw_a = raw_b;
                raw_b = raw_c;
                raw_c = raw_d;
                raw_d = load_raw<FMT, A4>(rowptr(r + 4), c0, W, a.byte_rows);
            }

            // horizontal pass, image_util.rs:137-185: taps in index order, mul then add
            // SHARE (round 5: every format on the aligned / unaligned data paths): a pixel's four distinct tap products (w[i] ==
            // w[6 - i] bit for bit) are formed ONCE, by the lane that owns the pixel, and the three pixels on either side are taken
            // as products from the neighbouring lane -- 16 multiplications per lane and row instead of 28 and no separate
            // neighbour moves (the DPP shift folds into the add).  The 8-bit formats read the products from the LDS table (FAST),
            // L16 / LF32 multiply.  The same products either way: x * w_i rounds identically in whichever lane evaluates it.
            constexpr bool SHARE = FAST || BUF;
            if (SHARE && !FAST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) P[j] = make_float4(m[j] * w0, m[j] * w1, m[j] * w2, m[j] * w3);
            }
            float x[10];
            if (!SHARE) {
                x[0] = from_left(m[1]); x[1] = from_left(m[2]); x[2] = from_left(m[3]);
                x[3] = m[0]; x[4] = m[1]; x[5] = m[2]; x[6] = m[3];
                x[7] = from_right(m[0]); x[8] = from_right(m[1]); x[9] = from_right(m[2]);
            }
            // SHARE: x[n] * w_i comes out of the table -- of the own pixels directly, of the three
            // pixels on either side through the neighbouring lane (the DPP shift folds into the add)
            auto prod = [&](int n, int i) -> float {
                const int t = i <= 3 ? i : 6 - i;
                const float4 &q = P[n < 3 ? n + 1 : (n > 6 ? n - 7 : n - 3)];
                const float e = t == 0 ? q.x : (t == 1 ? q.y : (t == 2 ? q.z : q.w));
                return n < 3 ? from_left(e) : (n > 6 ? from_right(e) : e);
            };
            f32x4 bc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v;
                if (SHARE) {
                    // (LF32: the reference folds from 0.0 -- image_util.rs:146,160,178 -- and 0.0 + (-0.0) is +0.0, which only an
                    // arbitrary f32 plane can produce)
                    v = FMT == 3 ? 0.0f + prod(j, 0) : prod(j, 0);
                    v = v + prod(j + 1, 1);
                    v = v + prod(j + 2, 2);
                    v = v + prod(j + 3, 3);
                    v = v + prod(j + 4, 4);
                    v = v + prod(j + 5, 5);
                    v = v + prod(j + 6, 6);
                } else {
                    // the reference folds from 0.0 (image_util.rs:146,160,178: `let mut sum = 0.0`): 0.0 + (-0.0) is
                    // +0.0, which only an arbitrary f32 plane can produce (integer luma is >= 0, the taps > 0)
                    v = FMT == 3 ? 0.0f + x[j] * w0 : x[j] * w0;
                    v = v + x[j + 1] * w1;
                    v = v + x[j + 2] * w2;
                    v = v + x[j + 3] * w3;
                    v = v + x[j + 4] * w4;
                    v = v + x[j + 5] * w5;
                    v = v + x[j + 6] * w6;
                }
                // vertical pass: blur row b = r-3 completes, rows r-2 .. r+3 advance
                const float p0 = v * w0, p1 = v * w1, p2 = v * w2, p3 = v * w3;
                bc[j] = acc[k][j] + p0;                              // tap 6 of row r-3
                acc[(k + 1) % 7][j] = acc[(k + 1) % 7][j] + p1;      // tap 5 of row r-2
                acc[(k + 2) % 7][j] = acc[(k + 2) % 7][j] + p2;      // tap 4 of row r-1
                acc[(k + 3) % 7][j] = acc[(k + 3) % 7][j] + p3;      // tap 3 of row r
                acc[(k + 4) % 7][j] = acc[(k + 4) % 7][j] + p2;      // tap 2 of row r+1
                acc[(k + 5) % 7][j] = acc[(k + 5) % 7][j] + p1;      // tap 1 of row r+2
                acc[(k + 6) % 7][j] = FMT == 3 ? 0.0f + p0 : p0;     // tap 0 of row r+3 (free slot; image_util.rs:190-201 starts from 0.0, see above)
            }
            const int b = r - 3;
            // the store is issued for every row (rows outside the segment go to a dummy row) so
            // that the number of memory operations per row is fixed and the compiler can wait
            // with a counted vmcnt(N) instead of draining the stores before every row
            if (store_ok) {
                if (A4 || UF) {  // valid lanes hold 4 in-image pixels (UF: the last one of a row may hold fewer); rows outside the segment are dropped
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v4 = __builtin_bit_cast(u32x4, bc);
                    const uint32_t row_off = (b >= ys && b < ye) ? (uint32_t)((a.dbg & 8) ? (b & 7) : b) * (uint32_t)W * 4u : blur_bytes;
                    if (A4 || c0 + 3 < W) {  // (UF: rows of the plane are only 4-byte aligned; a 16-byte store needs no more)
                        __builtin_amdgcn_raw_buffer_store_b128(v4, rs_blur, c0 * 4, (int)row_off, AGX_BLUR_STORE_AUX);
                    } else {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            if (c0 + j < W) __builtin_amdgcn_raw_buffer_store_b32(v4[j], rs_blur, (c0 + j) * 4, (int)row_off, AGX_BLUR_STORE_AUX);
                    }
                } else {
                    float *dst = (b >= ys && b < ye) ? blur_f + (size_t)((a.dbg & 8) ? (b & 7) : b) * W + c0 : a.dummy + c0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < W) __builtin_nontemporal_store(bc[j], &dst[j]);
                }
            }
            // Hessian determinant of row y = b-1 (rows b-2, b-1, b), image_util.rs:88-106
            Row6 dn;
            dn.c = bc;
            const int y = b - 1;
            if (y >= ys && y < ye && !(a.dbg & 4)) {  // wave-uniform
                if (y > 0 && y < H - 1) {
                    float dv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v11 = up[j], v12 = up[j + 1], v13 = up[j + 2];
                        const float v21 = mid[j], v22 = mid[j + 1], v23 = mid[j + 2];
                        const float v31 = dn[j], v32 = dn[j + 1], v33 = dn[j + 2];
                        // columns outside the lane's share (halo lanes, the image's border ring) hold a
                        // meaningless value here; they are masked where the columns are combined
                        if (FMT == 3) {  // arbitrary f32 planes: image_util.rs:100-104 literally
                            const float t22 = v22 * 2.0f;
                            const float lxx = (v21 - t22) + v23;
                            const float lyy = (v12 - t22) + v32;
                            const float lxy = (((v13 - v11) + v31) - v33) * 0.25f;
                            dv[j] = lxx * lyy - lxy * lxy;
                        } else {
                            // the same with three roundings removed that cannot happen: 2*v22 is exact, so
                            // (v21 - 2*v22) rounds once either way; lxy = s*0.25 and lxy*lxy = RN(s*s)/16 are
                            // exact scalings (luma from 8 / 16-bit integers: s is 0 or far above the subnormal
                            // range), so lxx*lyy - lxy*lxy = RN(RN(lxx*lyy) - RN(s*s)/16), which is what the
                            // last fma evaluates.  Checked bit for bit against the oracle through the
                            // stored-response instantiation (AGX_DBG_RESP).
                            const float lxx = __builtin_fmaf(v22, -2.0f, v21) + v23;
                            const float lyy = __builtin_fmaf(v22, -2.0f, v12) + v32;
                            const float sxy = ((v13 - v11) + v31) - v33;
                            dv[j] = __builtin_fmaf(sxy * sxy, -0.0625f, lxx * lyy);
                        }
                    }
                    if (RESP) {  // the in-register response itself (border ring stays zero)
                        float *rrow = a.resp_dbg + (size_t)frame * (size_t)a.plane + (size_t)y * W + c0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (min_ok[j]) rrow[j] = dv[j];
                    }
                    if (A4) {
                        // (as instructions: fminf() makes the compiler canonicalise the loop-carried operand first -- a v_max_f32 per
                        // minimum; the operands are results of arithmetic or of these minima, v_min_f32 returns the other operand for a NaN)
                        asm("v_min_f32 %0, %0, %1" : "+v"(rmin[0]) : "v"(dv[0]));
                        asm("v_min3_f32 %0, %0, %1, %2" : "+v"(rmin[1]) : "v"(dv[1]), "v"(dv[2]));
                        asm("v_min_f32 %0, %0, %1" : "+v"(rmin[3]) : "v"(dv[3]));
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) rmin[j] = fminf(rmin[j], dv[j]);
                    }
                    // refresh the running threshold with exponential back-off (rows 0,1,3,7,15,31,63,..).
                    // The poll of the frame's published minimum is asynchronous: a sync point uses
                    // the value fetched at the previous one and issues the next fetch without
                    // waiting for it (a staler threshold is only a slightly larger superset).
                    if (rows_to_sync <= 0 && !(a.dbg & 16)) {
                        float wmin = wave_min_f32(lane_min());
                        const int wmin_bits = __builtin_bit_cast(int, wmin);
                        const float published = __builtin_bit_cast(float, published_bits);
                        if (async_poll) polled = __builtin_amdgcn_readfirstlane(polled_v);
                        if (first_sync) {  // the fetch issued at the start of the wave
                            if (!async_poll) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(polled) : : "memory");
                            first_sync = false;
                            // a wave that starts when the frame has already published a minimum (the later dispatch rounds)
                            // refreshes less often: what it would learn it mostly knows
                            if (polled && !(a.dbg & 65536)) gap_cap = AGX_K1_SYNC_GAP_LATE;
                        }
                        const float gmin = polled ? f32_from_order_key(~polled) : 0.0f;  // 0 = nothing seen yet
                        // Every wave of a frame publishes to ONE word, and atomics on one L2 line serialise
                        // (~45 ns each: with one 1920x1080 frame on the chip, 272 waves x 6 sync points made
                        // K1 106 us instead of 33).  Frames cut into many waves (few frames per batch:
                        // plan_k1 sets publish_factor 1.5, else 1) publish only what improves on the frame's
                        // known minimum by more than half.  A staler published value only widens the
                        // candidate superset; the exact minimum is published unconditionally at the wave's end.
                        if (wmin < published && wmin < gmin * a.publish_factor) {
                            if (lane == 0) atomicMax(&ctr.min_key_inv, ~f32_order_key(wmin));
                            published_bits = wmin_bits;
                        }
                        // (asm: the builtin would be commuted with the multiply and leave a VGPR.  The
                        // assembler inserts no wait states inside asm: on gfx950 a readlane needs one
                        // after the VALU write of its source, a VALU read of the SGPR two after this)
                        asm("s_nop 0\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(thr_run_bits) : "v"(fminf(wmin, gmin) * 0.05f));
                        if (VALU_MASK) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) thr_v[j] = min_ok[j] ? __builtin_bit_cast(float, thr_run_bits) : -__builtin_inff();
                        }
                        if (async_poll) polled_v = poll_min();  // read at the next sync point
                        else  // scalar load past the scalar cache, awaited here
                            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(polled) : "s"(&ctr.min_key_inv) : "memory");
                        rows_to_sync = sync_gap;
                        sync_gap = min(sync_gap * 2, gap_cap);
                    }
                    --rows_to_sync;
                    // Candidate bit of this row: compare -> lane mask (SGPR pair), restricted to the
                    // lane's own columns, shifted into the word by an add-with-carry (mw = 2*mw + bit:
                    // rows enter at bit 0, the word is bit-reversed when it is stored).  One block with
                    // the four columns interleaved: a VALU-written SGPR needs two wait states before
                    // a VALU reads it on gfx950, which the assembler does not insert inside asm.
                    if constexpr (VALU_MASK) {
                        // All on the vector ALU: compare against the column's own threshold (lane mask in an SGPR pair),
                        // add-with-carry into the word, and the block's weakest candidate kept unconditionally (the candidate's
                        // response, or the current maximum itself).  The former block went VALU -> SALU (s_and with the
                        // column masks, s_or for "any candidate in this row") -> VALU and branched on the s_or: with one
                        // wave per SIMD the round trips through the scalar unit were the longest stretch of the row.
                        uint64_t cj[4];
                        float sel[4];
                        asm("v_cmp_lt_f32_e64 %[c0], %[d0], %[t0]\n\t"
                            "v_cmp_lt_f32_e64 %[c1], %[d1], %[t1]\n\t"
                            "v_cmp_lt_f32_e64 %[c2], %[d2], %[t2]\n\t"
                            "v_cmp_lt_f32_e64 %[c3], %[d3], %[t3]\n\t"
                            "v_addc_co_u32_e64 %[m0], vcc, %[m0], %[m0], %[c0]\n\t"
                            "v_addc_co_u32_e64 %[m1], vcc, %[m1], %[m1], %[c1]\n\t"
                            "v_addc_co_u32_e64 %[m2], vcc, %[m2], %[m2], %[c2]\n\t"
                            "v_addc_co_u32_e64 %[m3], vcc, %[m3], %[m3], %[c3]\n\t"
                            "v_cndmask_b32_e64 %[s0], %[cm], %[d0], %[c0]\n\t"
                            "v_cndmask_b32_e64 %[s1], %[cm], %[d1], %[c1]\n\t"
                            "v_cndmask_b32_e64 %[s2], %[cm], %[d2], %[c2]\n\t"
                            "v_cndmask_b32_e64 %[s3], %[cm], %[d3], %[c3]\n\t"
                            "v_max3_f32 %[cm], %[cm], %[s0], %[s1]\n\t"
                            "v_max3_f32 %[cm], %[cm], %[s2], %[s3]"
                            : [m0] "+v"(mw[0]), [m1] "+v"(mw[1]), [m2] "+v"(mw[2]), [m3] "+v"(mw[3]), [cm] "+v"(cmax),
                              [c0] "=&s"(cj[0]), [c1] "=&s"(cj[1]), [c2] "=&s"(cj[2]), [c3] "=&s"(cj[3]),
                              [s0] "=&v"(sel[0]), [s1] "=&v"(sel[1]), [s2] "=&v"(sel[2]), [s3] "=&v"(sel[3])
                            : [d0] "v"(dv[0]), [d1] "v"(dv[1]), [d2] "v"(dv[2]), [d3] "v"(dv[3]),
                              [t0] "v"(thr_v[0]), [t1] "v"(thr_v[1]), [t2] "v"(thr_v[2]), [t3] "v"(thr_v[3])
                            : "vcc");
                    } else {
                        uint64_t cj[4], cj_any;  // cj_any: some lane has a candidate in this row
                        asm("v_cmp_lt_f32_e64 %[c0], %[d0], %[thr]\n\t"
                            "v_cmp_lt_f32_e64 %[c1], %[d1], %[thr]\n\t"
                            "v_cmp_lt_f32_e64 %[c2], %[d2], %[thr]\n\t"
                            "v_cmp_lt_f32_e64 %[c3], %[d3], %[thr]\n\t"
                            "s_and_b64 %[c0], %[c0], %[k0]\n\t"
                            "s_and_b64 %[c1], %[c1], %[k1]\n\t"
                            "s_and_b64 %[c2], %[c2], %[k2]\n\t"
                            "s_and_b64 %[c3], %[c3], %[k3]\n\t"
                            "v_addc_co_u32_e64 %[m0], vcc, %[m0], %[m0], %[c0]\n\t"
                            "v_addc_co_u32_e64 %[m1], vcc, %[m1], %[m1], %[c1]\n\t"
                            "v_addc_co_u32_e64 %[m2], vcc, %[m2], %[m2], %[c2]\n\t"
                            "v_addc_co_u32_e64 %[m3], vcc, %[m3], %[m3], %[c3]\n\t"
                            "s_or_b64 %[any], %[c0], %[c1]\n\t"
                            "s_or_b64 %[any], %[any], %[c2]\n\t"
                            "s_or_b64 %[any], %[any], %[c3]"
                            : [m0] "+v"(mw[0]), [m1] "+v"(mw[1]), [m2] "+v"(mw[2]), [m3] "+v"(mw[3]),
                              [c0] "=&s"(cj[0]), [c1] "=&s"(cj[1]), [c2] "=&s"(cj[2]), [c3] "=&s"(cj[3]), [any] "=&s"(cj_any)
                            : [d0] "v"(dv[0]), [d1] "v"(dv[1]), [d2] "v"(dv[2]), [d3] "v"(dv[3]), [thr] "s"(thr_run_bits),
                              [k0] "s"(ok_mask[0]), [k1] "s"(ok_mask[1]), [k2] "s"(ok_mask[2]), [k3] "s"(ok_mask[3])
                            : "vcc", "scc");  // s_and_b64 writes SCC
                        const bool defer_row = DEFER && __builtin_expect((y - ys) < K1_DEFER_ROWS, 0);  // wave-uniform, 10 rows of a segment
                        if (defer_row) {  // the row's bits are decided again later: keep its responses (bf16, truncated)
                            uint2 pk;
                            pk.x = __builtin_amdgcn_perm(__float_as_uint(dv[1]), __float_as_uint(dv[0]), 0x07060302u);
                            pk.y = __builtin_amdgcn_perm(__float_as_uint(dv[3]), __float_as_uint(dv[2]), 0x07060302u);
                            s_def[wv][y - ys][lane_id_here()] = pk;
                        }
                        if (cj_any != 0ull && !defer_row) {  // wave-uniform; most rows have no candidate
                            float sel[4];  // the candidate's response, or the current maximum itself
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(sel[j]) : "v"(cmax), "v"(dv[j]), "s"(cj[j]));
                            cmax = fmaxf(fmaxf(cmax, sel[0]), sel[1]);
                            cmax = fmaxf(fmaxf(cmax, sel[2]), sel[3]);
                        }
                    }
                    y_pushed = y;
                }
                if (__builtin_expect((y & 31) == 31 || y == ye - 1, 0)) {  // word row complete (segments are 32-row aligned)
                    const int ln = lane_id_here();
                    const int c0h = xs - 4 + 4 * ln;  // == c0, derived here (see lane_id_here)
                    auto store_block = [&](int yb, const uint32_t (&w)[4], float m) {
                        if (c0h >= xs && c0h < xe) {  // == lane_valid
                            a.cand_max[((size_t)frame * a.mask_yb + yb) * (a.mask_wpr >> 2) + ((MASK_PAD_X + c0h) >> 2)] = m;
                            uint32_t *dst = mask_f + (size_t)yb * a.mask_wpr + MASK_PAD_X + c0h;
                            if (A4) {
                                *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (c0h + j < W) dst[j] = w[j];
                            }
                        }
                    };
                    // the deferred rows of the segment's first word row, decided against the running threshold as
                    // it stands now (row ys + q is bit q of the word: segments start on a word row)
                    auto refilter = [&](uint32_t (&w)[4], float &m) {
                        const float thr_now = __builtin_bit_cast(float, thr_run_bits);
                        // rows with a response: 1 <= ys + q <= min(ye, H - 1) - 1 (wave-uniform bounds; a rolled loop:
                        // the body sits in each of the seven unrolled row bodies)
                        const int q_lo = ys < 1 ? 1 - ys : 0;
                        const int q_hi = min(K1_DEFER_ROWS, min(ye, H - 1) - ys);
#pragma unroll 1
                        for (int q = q_lo; q < q_hi; ++q) {
                            const uint2 pk = s_def[wv][q][ln];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t half = j < 2 ? pk.x : pk.y;
                                const uint32_t raw = (j & 1) ? (half & 0xffff0000u) : (half << 16);
                                const float hi = __uint_as_float(raw), lo = __uint_as_float(raw | 0xffffu);  // hi >= response >= lo (negative)
                                const bool bit = ((w[j] >> q) & 1u) != 0u;
                                const bool keep = bit && lo < thr_now;
                                if (bit && !keep) w[j] &= ~(1u << q);
                                if (keep) m = fmaxf(m, hi);
                            }
                        }
                    };
                    const int blk = (y >> 5) - (ys >> 5);  // word row within the segment (wave-uniform)
                    if (DEFER && blk == 1) {  // the segment's first word row is decided now and goes out first
                        uint32_t o0[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) o0[j] = s_blk0[wv][j][ln];
                        float cm0 = __uint_as_float(s_blk0[wv][4][ln]);
                        refilter(o0, cm0);
                        store_block((y >> 5) - 1, o0, cm0);
                    }
                    // the last row pushed sits at bit 0: reverse, then move it to bit (row & 31)
                    const int fix = 31 - (y_pushed & 31);
#pragma unroll
                    for (int j = 0; j < 4; ++j) mw[j] = __brev(mw[j]) >> fix;
                    if (DEFER && blk == 0 && y != ye - 1) {
                        // first word row, more rows follow: decided when the second one is complete
#pragma unroll
                        for (int j = 0; j < 4; ++j) s_blk0[wv][j][ln] = mw[j];
                        s_blk0[wv][4][ln] = __float_as_uint(cmax);
                    } else {
                        if (DEFER && blk == 0) refilter(mw, cmax);  // a segment of one word row
                        store_block(y >> 5, mw, cmax);
                    }
                    mw[0] = mw[1] = mw[2] = mw[3] = 0u;
                    cmax = -__builtin_inff();
                }
            }
            up = mid;
            mid = dn;
        }
    }
    // per-frame min: wave reduction, one atomic per wave
    const float run_min = wave_min_f32(lane_min());
    // The wave's exact minimum goes to the frame's word unless the word is known to hold something at
    // least as small already (the last value polled from it; it only ever decreases).  A word never
    // polled non-zero may still be unset: then the wave publishes in any case, so the word ends up valid.
    if (async_poll) polled = __builtin_amdgcn_readfirstlane(polled_v);
    else if (first_sync) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(polled) : : "memory");  // (a wave without a sync point)
    const bool known_smaller = polled != 0u && !(run_min < f32_from_order_key(~polled));
    if (lane == 0 && !known_smaller) atomicMax(&ctr.min_key_inv, ~f32_order_key(run_min));
}

// The Hessian determinant of pixel c[0] of the blur plane -- same expression, same operands as K1.
__device__ __forceinline__ float det_at(const float *c, int W)
{
    const float v11 = c[-W - 1], v12 = c[-W], v13 = c[-W + 1];
    const float v21 = c[-1], v22 = c[0], v23 = c[1];
    const float v31 = c[W - 1], v32 = c[W], v33 = c[W + 1];
    const float t22 = v22 * 2.0f;
    const float lxx = (v21 - t22) + v23;
    const float lyy = (v12 - t22) + v32;
    const float lxy = (((v13 - v11) + v31) - v33) * 0.25f;
    return lxx * lyy - lxy * lxy;
}

// Re-test the set bits of one mask word against the final threshold (exact response from the blur
// plane); returns the bits that stay.  Four bits per round: their 36 loads are in flight together
// (a word's bits are mostly a vertical run, so the windows share lines); a round with fewer bits
// left repeats the last one.
__device__ __forceinline__ uint32_t retest_word(const float *blur, int W, int yb, int x, uint32_t m0, float thr)
{
    uint32_t m = m0, keep = m0;
    const float *col = blur + (size_t)(yb * 32) * W + x;  // interior pixels only (K1 sets no border bits)
    while (m) {
        int b[4];
        b[0] = __ffs(m) - 1;
        m &= m - 1;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            b[q] = m ? __ffs(m) - 1 : b[q - 1];
            m &= m - 1;  // 0 stays 0
        }
        float d[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = det_at(col + (size_t)b[q] * W, W);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (!(d[q] < thr)) keep &= ~(1u << b[q]);
    }
    return keep;
}

// ------------------------------------------------------------------------------------------
// K2: verify + seeds in ONE pass over the mask.  K1 left a superset of the candidates (threshold
// from a running minimum).  Where the weakest candidate K1 admitted in a word's block does not
// already pass the final threshold, the Hessian determinant is recomputed from the blur plane at
// the set bits -- same expression, same operands as K1 -- and a bit stays only if
// resp < 0.05*min_frame (detector.rs:418, :177).  Flood seeds come out of the same registers: a
// seed is a candidate with no candidate to its left and none above, minus those whose run along
// the row reaches (within 7 columns) a pixel with a candidate above it -- such a pixel is
// 4-connected to an earlier one, so it cannot be the first pixel of its component, and dropping it
// here saves a whole flood (2.1 -> 1.1 seeds per cluster).
//
// One wave owns 56 columns x up to 4 word rows (128 image rows, the height of a K1 segment): lane = column, lane 0 and lanes
// 57..63 are halo lanes (the left neighbour and the 7 columns of look-ahead), verified redundantly
// (verification is idempotent, so it does not matter whether the owning wave has already rewritten
// a word).  The kernel is a chain of memory round trips, so each of them is made wide: all of a
// lane's words, then all their block maxima, are fetched together; the bits that need the blur
// plane go into a work list in LDS (wave prefix sum) and are re-tested one bit per lane with all
// loads in flight, whatever lane they came from; failures clear their bit with an LDS atomic.  The
// seeds of the tile get their places by a prefix sum over the lanes and go to the frame's list behind one atomic.
// ------------------------------------------------------------------------------------------
constexpr int VS_OWN = 56;         // owner lanes of a wave (1 left halo + 56 + 7 look-ahead = 64)
constexpr int VS_ROWS = 4;         // word rows per tile
constexpr uint32_t VS_LIST = 512;  // re-test work list entries per pass

// Workgroups of VS_WAVES independent waves (each with its own part of the LDS arrays, no workgroup barrier): the
// launch used to be 41 000 single-wave workgroups for 256 frames and was bound by the rate at which workgroups
// are dispatched (~580 per us sustained, half of the wave slots empty); a wave = a slot of its frame as before.
#ifndef AGX_VS_ATTR
// k_verify_seeds is a chain of memory round trips: as many waves as possible per SIMD.  Left alone the compiler
// takes 106 scalar registers (6 waves per SIMD); with 80 it fits 8: 59 -> 54 us for 256 frames.
#define AGX_VS_ATTR __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8)))
#endif
#ifndef AGX_VS_WAVES
#define AGX_VS_WAVES 1
#endif
constexpr int VS_WAVES = AGX_VS_WAVES;
#ifndef AGX_VS_RT
#define AGX_VS_RT 2
#endif
constexpr int VS_RT = AGX_VS_RT;  // re-tested bits per lane and round trip of k_verify_seeds
// orders the LDS traffic of ONE wave (its lanes exchange data through LDS; the LDS queue of a wave is in order)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The frame's seed list as the fused sparse kernel (k_sparse_frame) keeps it: the first SEED_LDS_CAP seeds in LDS, the rest
// (noise frames) in the frame's global list behind them; the counters and flags of the frame in LDS.
constexpr uint32_t SEED_LDS_CAP = 4096;
struct FrameLds {
    uint32_t *n_seeds, *flags;  // LDS words
    uint32_t *seeds;            // LDS [SEED_LDS_CAP]
};

// Seeds of one word row (lane = column): kw = the lane's verified word, upw = bit q set where the pixel above (column,
// row q) is a candidate.  A seed is a candidate with no candidate to its left and none above, minus those whose run along
// the row reaches (within 7 columns) a pixel with a candidate above it (see k_verify_seeds).  Valid in lanes 1 .. 56 of
// a 64-lane tile (1 left halo lane, 7 look-ahead lanes).
__device__ __forceinline__ uint32_t seed_bits(uint32_t kw, uint32_t upw)
{
    uint32_t sd = kw & ~from_left_u(kw) & ~upw;
    uint32_t alive = sd, mk = kw, uk = upw;  // rows whose run still continues and has not met a pixel with one above
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        if (!__any(alive != 0u)) break;  // wave-uniform: no run of the wave reaches this far (clusters are a few pixels wide)
        mk = from_right_u(mk);
        uk = from_right_u(uk);
        alive &= mk;
        const uint32_t kill = alive & uk;
        sd &= ~kill;
        alive &= ~kill;
    }
    return sd;
}

// Tiles first_tile, first_tile + tile_stride, ... of `frame` by one wave of k_verify_seeds (the tile's seeds are collected
// by a prefix sum over the lanes and appended to the frame's global list with one atomic per tile).
// DBG: the instantiation that looks at the debug_ablation bits (statistics, phase clocks, ablations); the product's has none of
// their tests in its loops (k_verify_seeds -2 us per batch: the kernel is a chain of short scalar-controlled steps).
constexpr uint32_t K2_DBG_BITS = 32u | 64u | 128u | 256u | 2048u | 8192u;
// GRID3: the launch is (frames, column groups, row chunks) -- one tile per wave, its coordinates are the workgroup's own (no
// integer divisions per wave: three of them were a tenth of the wave's instructions).
template <bool DBG, bool GRID3 = false>
__device__ __forceinline__ void verify_tiles(const ChainArgs &a, int frame, int first_tile, int tile_stride, uint32_t *s_keep,
                                             uint32_t *s_list)
{
    const uint32_t dbg = DBG ? a.dbg : 0u;
    FrameCounters &ctr = a.ctr[frame];
    uint32_t *mask = a.mask + (size_t)frame * (size_t)a.mask_plane;
    const float *blur = a.blur + (size_t)frame * (size_t)a.plane;
    const float *cmax_f = a.cand_max + (size_t)frame * (size_t)a.mask_yb * (size_t)(a.mask_wpr >> 2);
    const int W = a.W, wpr = a.mask_wpr;
    const int lane = threadIdx.x & 63;
    const int n_yb = (a.H + 31) >> 5;
    const int groups = GRID3 ? (int)gridDim.y : (W + VS_OWN - 1) / VS_OWN;
    const int tiles = GRID3 ? first_tile + 1 : ((n_yb + VS_ROWS - 1) / VS_ROWS) * groups;
    // debug_ablation & 8192: where a wave's time goes -- 10 ns ticks per phase summed into the frame's stats[0..5]
    // (first loads, block maxima + threshold, work list + re-tests, seeds, list append, rest), tiles in stats[7]
    const bool phase_on = (dbg & 8192) != 0;
    unsigned long long t_prev = phase_on ? wall_clock64() : 0ull;
    auto phase = [&](int which) {
        if (phase_on) {
            const unsigned long long now = wall_clock64();
            if (lane == 0) atomicAdd(&ctr.stats[which], (uint32_t)(now - t_prev));
            t_prev = now;
        }
    };
    for (int t = first_tile; t < tiles; t += tile_stride) {  // wave-uniform
        const int ch = GRID3 ? (int)blockIdx.z : t / groups, g = GRID3 ? (int)blockIdx.y : t - ch * groups;
        const int yb0 = ch * VS_ROWS;
        const int x = g * VS_OWN - 1 + lane;  // -1 .. W + 62: inside the mask's zero padding
        uint32_t *wp0 = mask + (size_t)yb0 * wpr + MASK_PAD_X + x;
        const size_t cm_col = (size_t)((MASK_PAD_X + x) >> 2);
        // round trip 1: the lane's words, the word above the tile, their blocks' maxima, the frame's minimum -- eleven
        // loads issued back to back and awaited together.  All of them unconditional: word rows past the image exist and are zero (the
        // mask plane has mask_yb = H/32 + 4 of them), the maxima of such rows are never looked at (their words are
        // zero), and the row above the first tile is replaced by the tile's own first row and discarded.  (With a
        // condition per row the compiler used to wait for each load before it issued the next: five round trips,
        // 40 % of the launch's wave time.)
        uint32_t m[VS_ROWS];
#pragma unroll
        for (int r = 0; r < VS_ROWS; ++r) m[r] = wp0[(size_t)r * wpr];
        const int up_row = yb0 > 0 ? -1 : 0;
        uint32_t uword = wp0[(ptrdiff_t)up_row * wpr];
        float cm_pre[VS_ROWS];
#pragma unroll
        for (int r = 0; r < VS_ROWS; ++r) cm_pre[r] = cmax_f[(size_t)(yb0 + r) * (wpr >> 2) + cm_col];
        float cmu_pre = cmax_f[(size_t)(yb0 + up_row) * (wpr >> 2) + cm_col];
        uint32_t min_key_inv = ctr.min_key_inv;  // the frame's minimum (K1 is through): in the same round trip
        asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(uword), "+v"(cm_pre[0]), "+v"(cm_pre[1]), "+v"(cm_pre[2]),
                     "+v"(cm_pre[3]), "+v"(cmu_pre), "+v"(min_key_inv));  // (all consumed here: nothing is left to be loaded behind a branch)
        const float thr = f32_from_order_key(~min_key_inv) * 0.05f;  // detector.rs:418
        static_assert(VS_ROWS == 4, "the asm above names the tile's four rows");
        if (yb0 == 0) uword = 0u;
        uint32_t any = 0u;
#pragma unroll
        for (int r = 0; r < VS_ROWS; ++r) any |= m[r];
        const bool tile_empty = !__any(any != 0u);
        phase(0);
        if (phase_on && lane == 0) atomicAdd(&ctr.stats[tile_empty ? 6 : 7], 1u);
        if (tile_empty) continue;
        // every candidate K1 admitted in a word's 4-column x 32-row block is <= cand_max;
        // if that is below the final threshold they all pass and nothing needs recomputing
        float cm[VS_ROWS];
#pragma unroll
        for (int r = 0; r < VS_ROWS; ++r) cm[r] = m[r] ? cm_pre[r] : -__builtin_inff();
        // the pixel above row 0 of the tile is bit 31 of the word above: its verified state
        const bool up_need = (m[0] & 1u) && (uword >> 31);
        const float cmu = up_need ? cmu_pre : -__builtin_inff();
        uint32_t needmask = 0u;
#pragma unroll
        for (int r = 0; r < VS_ROWS; ++r)
            if (m[r] && !(cm[r] < thr)) needmask |= 1u << r;
        const bool up_retest = up_need && !(cmu < thr);
        if (phase_on) { (void)__any(needmask != 0u || up_retest); phase(1); }
        if (dbg & 128) {  // statistics by word row within the K1 segment (4 word rows of 128 rows)
            const bool own = lane >= 1 && lane <= VS_OWN && x < W;
#pragma unroll
            for (int r = 0; r < VS_ROWS; ++r)
                if (own && m[r]) {
                    const int q = (yb0 + r) & 3;
                    atomicAdd(&ctr.stats[q * 4 + 0], 1u);                          // words with candidates
                    atomicAdd(&ctr.stats[q * 4 + 1], (uint32_t)__popc(m[r]));      // candidate bits (superset)
                    if (needmask & (1u << r)) {
                        atomicAdd(&ctr.stats[q * 4 + 2], 1u);                      // words to re-test
                        atomicAdd(&ctr.stats[q * 4 + 3], (uint32_t)__popc(m[r]));  // bits to re-test
                    }
                }
        }
        if (dbg & 2048) {  // where the re-tested bits sit: by image row inside a segment's first word row, by segment
            const bool own = lane >= 1 && lane <= VS_OWN && x < W;
#pragma unroll
            for (int r = 0; r < VS_ROWS; ++r)
                if (own && (needmask & (1u << r))) {
                    const int q = (yb0 + r) & 3, sg = (yb0 + r) >> 2;
                    atomicAdd(&ctr.stats[16 + (sg < 3 ? sg : 3)], (uint32_t)__popc(m[r]));
                    if (q == 0)
                        for (int h = 0; h < 16; ++h) atomicAdd(&ctr.stats[h], (uint32_t)__popc(m[r] & (3u << (2 * h))));
                }
        }
        // Work list in ROW-MAJOR order (image row, then column): neighbouring lanes then re-test
        // neighbouring columns of one row, so their nine loads each fall into one or two cache lines
        // (a list in lane order -- every lane's bits one after the other -- put 64 different rows
        // into every load and was 5x slower).  Built with one ballot per (word row, bit) that any
        // lane needs; most of the volume is the first rows of a K1 segment, where the running
        // threshold was still 0 and every lane has the same rows.
        uint32_t keep[VS_ROWS];
#pragma unroll
        for (int r = 0; r < VS_ROWS; ++r) keep[r] = m[r];
        uint32_t upbit = up_need ? 1u : 0u;
        const unsigned long long need_any = __ballot(needmask != 0u || up_retest);
        if (need_any && !(dbg & 32)) {  // wave-uniform
            const uint32_t rows_any = wave_or_u32(needmask);  // the word rows that need the blur plane
            uint32_t n_list = 0;  // wave-uniform fill of s_list
            auto run_list = [&]() {
                wave_lds_sync();
                for (uint32_t e0 = 0; e0 < n_list; e0 += 64 * VS_RT) {  // VS_RT bits per lane and round: 9 VS_RT loads in flight
                    float d[VS_RT];
                    uint32_t ent[VS_RT];
                    bool on[VS_RT];
#pragma unroll
                    for (int q = 0; q < VS_RT; ++q) {
                        const uint32_t e = e0 + q * 64 + (uint32_t)lane;
                        on[q] = e < n_list;
                        ent[q] = s_list[on[q] ? e : 0u];  // (word row * 64 + lane) << 5 | bit
                        const int ln = (int)((ent[q] >> 5) & 63u), r = (int)(ent[q] >> 11), b = (int)(ent[q] & 31u);
                        const int row = r == VS_ROWS ? yb0 * 32 - 1 : (yb0 + r) * 32 + b;
                        d[q] = (dbg & 256) ? 0.0f : det_at(blur + (size_t)row * W + (g * VS_OWN - 1 + ln), W);
                    }
#pragma unroll
                    for (int q = 0; q < VS_RT; ++q)
                        if (on[q] && !(d[q] < thr)) atomicAnd(&s_keep[ent[q] >> 5], ~(1u << (ent[q] & 31u)));
                }
                wave_lds_sync();
                n_list = 0;
            };
            auto push = [&](bool mine, uint32_t entry) {  // one ballot: the lanes that have this (row, bit)
                const unsigned long long bal = __ballot(mine);
                const uint32_t k = (uint32_t)__popcll(bal);
                if (n_list + k > VS_LIST) run_list();
                if (mine) s_list[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = entry;
                n_list += k;
            };
            s_keep[VS_ROWS * 64 + lane] = 1u;
            if (__any(up_retest)) push(up_retest, (uint32_t)(VS_ROWS * 64 + lane) << 5);
#pragma unroll
            for (int r = 0; r < VS_ROWS; ++r) {
                if (!(rows_any & (1u << r))) continue;  // wave-uniform
                s_keep[r * 64 + lane] = m[r];
                const uint32_t mr = (needmask & (1u << r)) ? m[r] : 0u;
                uint32_t bits_any = wave_or_u32(mr);
                while (bits_any) {  // scalar loop over the image rows of this word row that any lane needs
                    const int b = __builtin_ctz(bits_any);
                    bits_any &= bits_any - 1;
                    push((mr >> b) & 1u, ((uint32_t)(r * 64 + lane) << 5) | (uint32_t)b);
                }
            }
            if (n_list) run_list();
#pragma unroll
            for (int r = 0; r < VS_ROWS; ++r)
                if (rows_any & (1u << r)) keep[r] = s_keep[r * 64 + lane];
            if (up_retest) upbit = s_keep[VS_ROWS * 64 + lane] & 1u;
        }
        phase(2);
        const bool owner = lane >= 1 && lane <= VS_OWN && x < W;
#pragma unroll
        for (int r = 0; r < VS_ROWS; ++r)
            if (owner && keep[r] != m[r]) wp0[(size_t)r * wpr] = keep[r];
        if (dbg & 128) {
#pragma unroll
            for (int r = 0; r < VS_ROWS; ++r)
                if (owner && m[r]) atomicAdd(&ctr.stats[16 + ((yb0 + r) & 3)], (uint32_t)__popc(keep[r]));  // bits that stay
        }
        // seeds, word row by word row; the bit above a word's row 0 is bit 31 of the word before
        uint32_t carry = upbit;
        if (dbg & 64) continue;
        uint32_t sdr[VS_ROWS], c = 0u;  // the lane's seed bits by word row, and how many they are
#pragma unroll
        for (int r = 0; r < VS_ROWS; ++r) {
            const uint32_t kw = keep[r];  // rows past the tile's end hold no bits
            const uint32_t upw = (kw << 1) | carry;  // bit q: the pixel above (column, row q) is a candidate
            carry = kw >> 31;
            sdr[r] = 0u;
            if (__any(kw != 0u)) {  // wave-uniform: a word row of the tile with a candidate
                const uint32_t sd = seed_bits(kw, upw);
                sdr[r] = owner ? sd : 0u;
            }
            c += (uint32_t)__popc(sdr[r]);
        }
        phase(3);
        // The tile's seeds go straight to the frame's list: a prefix sum over the lanes' counts gives every lane its place behind
        // ONE atomic of the wave (no staging in LDS, no atomic per seed; the list's order is the emission's business: it sorts).
        if (__any(c != 0u)) {  // wave-uniform
            // word row by word row (raster order within the tile, as the flood stage likes its neighbouring lanes): two scans of
            // two 16-bit counts each (a row's count over the wave is at most 56 x 32)
            static_assert(VS_ROWS == 4, "two packed scans");
            const uint32_t c01 = (uint32_t)__popc(sdr[0]) | ((uint32_t)__popc(sdr[1]) << 16), c23 = (uint32_t)__popc(sdr[2]) | ((uint32_t)__popc(sdr[3]) << 16);
            const uint32_t i01 = wave_incl_scan_u32(c01), i23 = wave_incl_scan_u32(c23);
            const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)i01, 63), t23 = (uint32_t)__builtin_amdgcn_readlane((int)i23, 63);
            const uint32_t tot[VS_ROWS] = {t01 & 0xffffu, t01 >> 16, t23 & 0xffffu, t23 >> 16};
            const uint32_t exc[VS_ROWS] = {(i01 - c01) & 0xffffu, (i01 - c01) >> 16, (i23 - c23) & 0xffffu, (i23 - c23) >> 16};
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(&ctr.n_seeds, tot[0] + tot[1] + tot[2] + tot[3]);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (phase_on) phase(4);
#pragma unroll
            for (int r = 0; r < VS_ROWS; ++r) {
                uint32_t sd = sdr[r], o = base + exc[r];
                base += tot[r];
                while (sd) {
                    const int b = __ffs(sd) - 1;
                    sd &= sd - 1;
                    const uint32_t pix = (uint32_t)((yb0 + r) * 32 + b) * (uint32_t)W + (uint32_t)x;
                    if (o < a.cap_roots) a.seeds[(size_t)frame * a.cap_roots + o] = pix;
                    else atomicOr(&ctr.flags, FLAG_CAND_OVERFLOW);
                    ++o;
                }
            }
        }
        phase(5);
    }
}

template <bool DBG, bool GRID3 = false>
__global__ void __launch_bounds__(64 * VS_WAVES) AGX_VS_ATTR k_verify_seeds(ChainArgs a)
{
    static_assert(!GRID3 || VS_WAVES == 1, "one tile per workgroup");
    __shared__ uint32_t s_keep_all[VS_WAVES][(VS_ROWS + 1) * 64];  // row VS_ROWS: bit 0 = the pixel above the tile's first row
    __shared__ uint32_t s_list_all[VS_WAVES][VS_LIST];  // re-test work list
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const WaveTimer wt(a, K_THRESHOLD, DBG);
    if (GRID3) {  // x = frame (fastest: tile t of every frame before tile t + 1 of any, as the slot-major linear grid), y, z = the tile
        verify_tiles<DBG, true>(a, a.n_frames - 1 - (int)blockIdx.x, 0, 1, s_keep_all[0], s_list_all[0]);
        return;
    }
    FrameSlot fs = frame_slot(a.n_frames, true);  // latest-written blur planes first (cache)
    fs.slot = fs.slot * VS_WAVES + (uint32_t)wv;  // this wave's slot of the frame
    fs.n_slots *= VS_WAVES;
    verify_tiles<DBG>(a, fs.frame, (int)fs.slot, (int)fs.n_slots, s_keep_all[wv], s_list_all[wv]);
}

// ------------------------------------------------------------------------------------------
// The verify + seed stage of ONE frame by the sixteen waves of a 1024-thread workgroup (k_sparse_frame) -- the same
// results as k_verify_seeds' tiles, organised by round trips instead of by tiles:
//   1  lane = column, tiles of 64 columns x VS_ROWS word rows, VF_TB tiles of a wave in flight together: the words and
//      their blocks' weakest admitted candidates in one round trip; a word that needs the blur plane gets a slot (word
//      index, pixel index of its bit 0, keep bits) in the wave's LDS and its set bits go into the wave's work list in
//      ROW-MAJOR order (one ballot per (word row, bit) that any lane needs, as in k_verify_seeds);
//   2  the wave's list -- all of its tiles' bits together, four per lane and round, 36 loads in flight --: failures
//      clear their bit in the slot; the slots go back to the mask with plain stores (this CU's L1 stays coherent);
// The seeds then come from frame_seeds (a pass of its own over the verified mask: a seed depends on the verified words of up to
// eight neighbouring columns and the row above, which other waves own).
// A frame's tiles are no longer a chain of two to three dependent round trips each (ten tiles per wave: 39 us per frame
// with sixteen waves per frame) but about eight round trips per wave.
// ------------------------------------------------------------------------------------------
constexpr int VF_TB = 4;               // tiles in flight per wave
constexpr uint32_t VFW_SLOTS = 256;    // re-tested words per wave and pass
constexpr uint32_t VFW_LIST = 512;     // re-tested bits per wave and pass
constexpr uint32_t VFW_WORDS = 3 * VFW_SLOTS + VFW_LIST;  // LDS words per wave
struct VerifyLds {  // this wave's part
    uint32_t *idx, *pix, *keep;  // [VFW_SLOTS]: word index in the frame's mask plane, pixel index of its bit 0, surviving bits
    uint32_t *list;              // [VFW_LIST]: slot << 5 | bit
};

__device__ __forceinline__ void verify_frame(const ChainArgs &a, int frame, const VerifyLds &vl)
{
    const uint32_t t = threadIdx.x;
    const int lane = (int)(t & 63u);
    const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    FrameCounters &ctr = a.ctr[frame];
    uint32_t *mask = a.mask + (size_t)frame * (size_t)a.mask_plane;
    const float *blur = a.blur + (size_t)frame * (size_t)a.plane;
    const float *cmax_f = a.cand_max + (size_t)frame * (size_t)a.mask_yb * (size_t)(a.mask_wpr >> 2);
    const int W = a.W, wpr = a.mask_wpr;
    const int n_yb = (a.H + 31) >> 5;
    const int n_ch = (n_yb + VS_ROWS - 1) / VS_ROWS;
    const float thr = f32_from_order_key(~ctr.min_key_inv) * 0.05f;  // detector.rs:418 (K1 is through)
    auto stamp = [&](int which) {  // debug_ablation & 131072 (see k_sparse_frame)
        if ((a.dbg & 131072) && t == 0) ctr.stats[which] = (uint32_t)wall_clock64();
    };
    // debug_ablation & 131072: where wave 0 spends the stage -- 10 ns ticks summed into the frame's stats[8..11]: waiting for the
    // tiles' words, slots + work list, the list's re-tests, write-back
    const bool wclk = (a.dbg & 131072) && wv == 0;
    unsigned long long t_prev = wclk ? wall_clock64() : 0ull;
    auto wphase = [&](int which) {
        if (wclk) {
            const unsigned long long now = wall_clock64();
            if (lane == 0) ctr.stats[8 + which] += (uint32_t)(now - t_prev);
            t_prev = now;
        }
    };
    // ---- 1 + 2: the words that need the blur plane ----
    uint32_t n_list = 0, n_slots = 0;  // wave-uniform
    auto run_list = [&]() {
        wphase(1);
        wave_lds_sync();
        constexpr int EB = 4;  // bits in flight per lane: 36 loads
        for (uint32_t e0 = 0; e0 < n_list; e0 += 64u * EB) {
            float d[EB];
            uint32_t ent[EB];
            bool on[EB];
#pragma unroll
            for (int k = 0; k < EB; ++k) {
                const uint32_t e = e0 + (uint32_t)k * 64u + (uint32_t)lane;
                on[k] = e < n_list;
                ent[k] = vl.list[on[k] ? e : 0u];
                d[k] = (a.dbg & 256) ? 0.0f : det_at(blur + (size_t)vl.pix[ent[k] >> 5] + (size_t)(ent[k] & 31u) * W, W);
            }
#pragma unroll
            for (int k = 0; k < EB; ++k)
                if (on[k] && !(d[k] < thr)) atomicAnd(&vl.keep[ent[k] >> 5], ~(1u << (ent[k] & 31u)));
        }
        wave_lds_sync();
        n_list = 0;
        wphase(2);
    };
    auto flush_slots = [&]() {
        run_list();
        for (uint32_t sl = (uint32_t)lane; sl < n_slots; sl += 64u) mask[vl.idx[sl]] = vl.keep[sl];
        wave_lds_sync();
        n_slots = 0;
        wphase(3);
    };
    auto push = [&](bool mine, uint32_t entry) {  // one ballot: the lanes that have this (row, bit)
        const unsigned long long bal = __ballot(mine);
        const uint32_t k = (uint32_t)__popcll(bal);
        if (n_list + k > VFW_LIST) run_list();
        if ((a.dbg & 131072) && lane == 0) atomicAdd(&ctr.stats[14], k);  // re-tested bits of the frame
        if (mine) vl.list[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = entry;
        n_list += k;
    };
    {
        const int groups = (W + 63) >> 6;  // tiles of 64 columns (no halo: every word belongs to one lane)
        const int tiles = n_ch * groups;
        for (int tb = wv; tb < tiles && !(a.dbg & 32); tb += 16 * VF_TB) {  // wave-uniform
            uint32_t m[VF_TB][VS_ROWS];
            float cm[VF_TB][VS_ROWS];
#pragma unroll
            for (int k = 0; k < VF_TB; ++k) {
                const int tl = tb + 16 * k;
                const bool tile_on = tl < tiles;  // wave-uniform
                const int ch = tile_on ? tl / groups : 0, g = tile_on ? tl - ch * groups : 0;
                const int x = g * 64 + lane;  // < W + 63: inside the mask's zero padding
                const uint32_t *wp0 = mask + (size_t)(ch * VS_ROWS) * wpr + MASK_PAD_X + x;
                const float *cp0 = cmax_f + (size_t)(ch * VS_ROWS) * (wpr >> 2) + ((MASK_PAD_X + x) >> 2);
#pragma unroll
                for (int r = 0; r < VS_ROWS; ++r) {  // (word rows past the image exist and are zero; their maxima are never looked at)
                    m[k][r] = tile_on ? wp0[(size_t)r * wpr] : 0u;
                    cm[k][r] = cp0[(size_t)r * (wpr >> 2)];
                }
            }
            if (wclk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wphase(0);
            }
#pragma unroll
            for (int k = 0; k < VF_TB; ++k) {
                const int tl = tb + 16 * k;
                if (tl >= tiles) break;  // wave-uniform
                const int ch = tl / groups, g = tl - ch * groups;
                const int yb0 = ch * VS_ROWS;
                const int x = g * 64 + lane;
                unsigned long long bal[VS_ROWS];
                uint32_t nw = 0;
#pragma unroll
                for (int r = 0; r < VS_ROWS; ++r) {
                    bal[r] = __ballot(m[k][r] != 0u && !(cm[k][r] < thr));
                    nw += (uint32_t)__popcll(bal[r]);
                }
                if (nw == 0) continue;  // wave-uniform: nothing to re-test in this tile
                if (n_slots + nw > VFW_SLOTS) flush_slots();
#pragma unroll
                for (int r = 0; r < VS_ROWS; ++r) {
                    if (!bal[r]) continue;  // wave-uniform
                    const bool need = (bal[r] >> lane) & 1ull;
                    const uint32_t slot = n_slots + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[r], 0u));
                    n_slots += (uint32_t)__popcll(bal[r]);
                    const uint32_t mr = need ? m[k][r] : 0u;
                    if (need) {
                        vl.idx[slot] = (uint32_t)((yb0 + r) * wpr + MASK_PAD_X + x);
                        vl.pix[slot] = (uint32_t)((yb0 + r) * 32) * (uint32_t)W + (uint32_t)x;
                        vl.keep[slot] = mr;
                    }
                    uint32_t bits_any = wave_or_u32(mr);
                    while (bits_any) {  // scalar loop over the image rows of this word row that any lane needs
                        const int b = __builtin_ctz(bits_any);
                        bits_any &= bits_any - 1;
                        push((mr >> b) & 1u, (slot << 5) | (uint32_t)b);
                    }
                }
            }
        }
        flush_slots();
    }
    wphase(1);
    __syncthreads();  // (workgroup scope: this CU's stores before this CU's loads)
    stamp(4);
}

// Stage 3 of k_sparse_frame: the flood seeds
// of the frame from its verified mask, tile by tile as in k_verify_seeds (lane = column, 1 + 7 halo lanes, DPP neighbours),
// VF_TB tiles of a wave in flight, into the frame's seed list in LDS.
__device__ __forceinline__ void frame_seeds(const ChainArgs &a, int frame, const FrameLds &fl)
{
    const uint32_t t = threadIdx.x;
    const int lane = (int)(t & 63u);
    const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint32_t *mask = a.mask + (size_t)frame * (size_t)a.mask_plane;
    const int W = a.W, wpr = a.mask_wpr;
    const int n_ch = (((a.H + 31) >> 5) + VS_ROWS - 1) / VS_ROWS;
    if (a.dbg & 64) return;
    const int groups = (W + VS_OWN - 1) / VS_OWN;
    const int tiles = n_ch * groups;
    for (int tb = wv; tb < tiles; tb += 16 * VF_TB) {  // wave-uniform
        uint32_t m[VF_TB][VS_ROWS], uw[VF_TB];
#pragma unroll
        for (int k = 0; k < VF_TB; ++k) {
            const int tl = tb + 16 * k;
            const bool tile_on = tl < tiles;
            const int ch = tile_on ? tl / groups : 0, g = tile_on ? tl - ch * groups : 0;
            const int yb0 = ch * VS_ROWS;
            const int x = g * VS_OWN - 1 + lane;  // -1 .. W + 62: inside the mask's zero padding
            const uint32_t *wp0 = mask + (size_t)yb0 * wpr + MASK_PAD_X + x;
#pragma unroll
            for (int r = 0; r < VS_ROWS; ++r) m[k][r] = tile_on ? wp0[(size_t)r * wpr] : 0u;
            uw[k] = wp0[(ptrdiff_t)(yb0 > 0 ? -1 : 0) * wpr];
        }
#pragma unroll
        for (int k = 0; k < VF_TB; ++k) {
            const int tl = tb + 16 * k;
            if (tl >= tiles) break;  // wave-uniform
            const int ch = tl / groups, g = tl - ch * groups;
            const int yb0 = ch * VS_ROWS;
            const int x = g * VS_OWN - 1 + lane;
            const bool owner = lane >= 1 && lane <= VS_OWN && x < W;
            uint32_t carry = yb0 > 0 ? (uw[k] >> 31) : 0u;  // the pixel above row 0 of the tile: bit 31 of the word above
#pragma unroll
            for (int r = 0; r < VS_ROWS; ++r) {
                const uint32_t kw = m[k][r];
                const uint32_t upw = (kw << 1) | carry;  // bit q: the pixel above (column, row q) is a candidate
                carry = kw >> 31;
                if (!__any(kw != 0u)) continue;  // wave-uniform: most word rows of most tiles hold nothing
                uint32_t sd = seed_bits(kw, upw);
                if (!owner) sd = 0u;
                while (sd) {
                    const int b = __ffs(sd) - 1;
                    sd &= sd - 1;
                    const uint32_t pix = (uint32_t)((yb0 + r) * 32 + b) * (uint32_t)W + (uint32_t)x;
                    const uint32_t i = atomicAdd(fl.n_seeds, 1u);  // the workgroup's list: LDS first, the frame's global list behind it
                    if (i >= a.cap_roots) atomicOr(fl.flags, FLAG_CAND_OVERFLOW);
                    else if (i < SEED_LDS_CAP) fl.seeds[i] = pix;
                    else a.seeds[(size_t)frame * a.cap_roots + i] = pix;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// K3: bit-parallel flood fill, one seed per lane.  Window: 32 columns [sx-16, sx+15] x 32 rows
// [sy-1, sy+30]; comp[c] / cand[c] hold column c of the window as a 32-bit word (bit = row).
// Each sweep ORs the two neighbouring columns into a column and fills whole vertical runs of
// the mask that contain a set bit with a carry-propagation add (fill_runs).  The seed is the
// canonical one iff its component contains no pixel with a smaller raster index: then the lane
// emits the cluster with exact integer sums.  A component that touches the window's left /
// right / bottom edge may continue outside: it goes to the second tier (wave_flood_128x64,
// 128 x 64 window, by the whole wave that found the seed); a component that leaves that window too
// sets FLAG_BIG_CLUSTER and the whole frame is redone by the generic path (k_rare).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fill_runs(uint32_t s, uint32_t m)
{
    // all bits of the runs of m that contain a bit of s (s subset of m)
    const uint32_t upw = (((m + s) ^ m) | s) & m;  // from each run's lowest seed upwards
    const uint32_t mr = __brev(m), sr = __brev(s);
    const uint32_t dnw = __brev((((mr + sr) ^ mr) | sr) & mr);
    return upw | dnw;
}

__device__ __forceinline__ uint32_t bitpos_sum(uint32_t c)
{
    return (uint32_t)__popc(c & 0xAAAAAAAAu) + 2u * (uint32_t)__popc(c & 0xCCCCCCCCu) +
           4u * (uint32_t)__popc(c & 0xF0F0F0F0u) + 8u * (uint32_t)__popc(c & 0xFF00FF00u) +
           16u * (uint32_t)__popc(c & 0xFFFF0000u);
}

#ifndef AGX_FLOOD_COLS
#define AGX_FLOOD_COLS 32
#endif
#ifndef AGX_FLOOD_WPE
#define AGX_FLOOD_WPE 4  // waves per SIMD the fused flood + refine kernel is compiled for
#endif
// First-tier flood window: FLOOD_COLS columns [x0, x0 + FLOOD_COLS), x0 = (sx - FLOOD_SEED_LO) rounded down to 4, x 32 rows [sy - 1, sy + 30].  Both column arrays
// of the window live in registers (32 columns: 4 waves per SIMD; narrower windows were measured -- 24 columns run at 5
// waves per SIMD but send three times as many seeds to the second tier, and the launch got slower).
constexpr int FLOOD_COLS = AGX_FLOOD_COLS;
constexpr int FLOOD_SEED_LO = FLOOD_COLS / 2 - 2;  // the seed sits in window column FLOOD_SEED_LO .. FLOOD_SEED_LO + 3 (aligned window)
static_assert(FLOOD_COLS % 4 == 0 && MASK_PAD_X % 4 == 0 && MASK_PAD_X >= FLOOD_COLS, "the flood window is fetched as aligned quads");

// Second tier, wave-wide: lane t holds columns sx-64+2t and sx-63+2t of a 128-column x 64-row
// window [sy-1, sy+62] as two 64-bit words (bit = row); each round ORs the neighbouring columns
// (own pair + DPP from the adjacent lanes) and fills vertical runs; rounds repeat until no column
// changes.  All 64 lanes of the wave take part.  Returns nothing: emits the cluster, or flags the
// frame for the generic path when the component leaves this window too.
__device__ __forceinline__ unsigned long long brev64(unsigned long long v)
{
    return ((unsigned long long)__brev((uint32_t)v) << 32) | (unsigned long long)__brev((uint32_t)(v >> 32));
}
__device__ __forceinline__ unsigned long long fill_runs64(unsigned long long s, unsigned long long m)
{
    const unsigned long long upw = (((m + s) ^ m) | s) & m;
    const unsigned long long mr = brev64(m), sr = brev64(s);
    const unsigned long long dnw = brev64((((mr + sr) ^ mr) | sr) & mr);
    return upw | dnw;
}
__device__ __forceinline__ uint32_t bitpos_sum64(unsigned long long c)
{
    return (uint32_t)__popcll(c & 0xAAAAAAAAAAAAAAAAull) + 2u * (uint32_t)__popcll(c & 0xCCCCCCCCCCCCCCCCull) +
           4u * (uint32_t)__popcll(c & 0xF0F0F0F0F0F0F0F0ull) + 8u * (uint32_t)__popcll(c & 0xFF00FF00FF00FF00ull) +
           16u * (uint32_t)__popcll(c & 0xFFFF0000FFFF0000ull) + 32u * (uint32_t)__popcll(c & 0xFFFFFFFF00000000ull);
}
__device__ __forceinline__ unsigned long long from_left_u64(unsigned long long v)
{
    return (unsigned long long)from_left_u((uint32_t)v) | ((unsigned long long)from_left_u((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ unsigned long long from_right_u64(unsigned long long v)
{
    return (unsigned long long)from_right_u((uint32_t)v) | ((unsigned long long)from_right_u((uint32_t)(v >> 32)) << 32);
}

enum : int { FLOOD_NONE = 0, FLOOD_CLUSTER = 1, FLOOD_BIG = 2 };

// Second tier: the whole wave floods the 128 x 64 window around seed p.  Returns (in every lane) FLOOD_CLUSTER with
// the component's exact integer sums, FLOOD_NONE if the seed is not the component's first pixel, or FLOOD_BIG
// if the component leaves this window too (the frame then goes to the generic path: FLAG_BIG_CLUSTER is set).
__device__ __forceinline__ int wave_flood_128x64(const ChainArgs &a, uint32_t *flags, const uint32_t *mask, uint32_t p, int lane,
                                                 uint32_t &cnt_out, uint32_t &sumx_out, uint32_t &sumy_out)
{
    const uint32_t W = (uint32_t)a.W;
    const uint32_t sx = p % W, sy = p / W;
    const int sh = (int)((sy - 1u) & 31u);
    const uint32_t *wp = mask + (size_t)((sy - 1u) >> 5) * a.mask_wpr + MASK_PAD_X + ((int)sx - 64) + 2 * lane;
    unsigned long long cand[2], comp[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const unsigned long long w01 = (unsigned long long)wp[c] | ((unsigned long long)wp[a.mask_wpr + c] << 32);
        const unsigned long long w2 = wp[2 * a.mask_wpr + c];
        cand[c] = sh ? ((w01 >> sh) | (w2 << (64 - sh))) : w01;  // bit r = row sy-1+r
        comp[c] = 0ull;
    }
    if (lane == 32) comp[0] = 2ull;  // the seed: column sx (window column 64), row sy
    for (;;) {
        const unsigned long long from_l = from_left_u64(comp[1]);   // column left of this lane's pair
        const unsigned long long from_r = from_right_u64(comp[0]);  // column right of it
        const unsigned long long f0 = fill_runs64((comp[0] | from_l | comp[1]) & cand[0], cand[0]);
        const unsigned long long f1 = fill_runs64((comp[1] | comp[0] | from_r) & cand[1], cand[1]);
        const bool ch = f0 != comp[0] || f1 != comp[1];
        comp[0] = f0;
        comp[1] = f1;
        if (!__any(ch)) break;
    }
    const unsigned long long both = comp[0] | comp[1];
    // not the canonical seed: a pixel of the component precedes it in raster order
    const bool earlier = (both & 1ull) != 0ull || (lane < 32 && (both & 2ull) != 0ull);
    const bool edge = (lane == 0 && comp[0] != 0ull) || (lane == 63 && comp[1] != 0ull) || (both >> 63) != 0ull;
    if (__any(earlier)) return FLOOD_NONE;
    if (__any(edge)) {
        if (lane == 0) atomicOr(flags, FLAG_BIG_CLUSTER);
        return FLOOD_BIG;
    }
    const uint32_t n0 = (uint32_t)__popcll(comp[0]), n1 = (uint32_t)__popcll(comp[1]);
    const uint32_t x0 = sx - 64u + 2u * (uint32_t)lane;
    uint32_t s_n = n0 + n1, s_x = n0 * x0 + n1 * (x0 + 1u), s_y = bitpos_sum64(comp[0]) + bitpos_sum64(comp[1]) + (n0 + n1) * (sy - 1u);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_n += __shfl_xor(s_n, off, 64);
        s_x += __shfl_xor(s_x, off, 64);
        s_y += __shfl_xor(s_y, off, 64);
    }
    cnt_out = s_n;
    sumx_out = s_x;
    sumy_out = s_y;
    return FLOOD_CLUSTER;
}

// debug_ablation & 16384: where the waves of k_flood_refine spend their time -- 10 ns ticks of s_memrealtime per
// phase, summed over the working waves of a frame into the frame's stats[8..] (tools/flood_phases.py).
// (A second instantiation of the kernel: the clock's registers would cost the product kernel scratch.)
struct NoClock {
    static constexpr bool on = false;
    __device__ __forceinline__ void start(uint32_t *) {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void join() {}
};
struct PhaseClock {
    static constexpr bool on = true;
    uint32_t *stats;
    unsigned long long t;
    __device__ __forceinline__ void start(uint32_t *st)
    {
        stats = st;
        t = wall_clock64();
    }
    __device__ __forceinline__ void mark(int which)  // the first active lane records; every active lane moves on
    {
        {
            const unsigned long long now = wall_clock64();
            const unsigned long long act = __ballot(true);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&stats[which], (uint32_t)(now - t));
            t = now;
        }
    }
    __device__ __forceinline__ void join()  // after a divergent stretch: everyone continues from the latest mark
    {
        {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(t >> 32), off, 64) << 32) | __shfl_xor((uint32_t)t, off, 64);
                t = o > t ? o : t;
            }
        }
    }
};

// First flood tier for one seed (one lane): 32 x 32 window.  Returns FLOOD_CLUSTER with the component's exact
// integer sums if the seed is the canonical one and the component stays inside the window, FLOOD_BIG if the
// component may continue outside (second tier), FLOOD_NONE if the seed is not the component's first pixel.
template <typename CLK>
__device__ __forceinline__ int flood_lane(const ChainArgs &a, const uint32_t *mask, uint32_t W, uint32_t p, uint32_t &cnt_out,
                                          uint32_t &sumx_out, uint32_t &sumy_out, CLK &clk)
{
    const uint32_t sx = p % W, sy = p / W;
    const int sh = (int)((sy - 1u) & 31u);
    // The window starts at a multiple of four columns, FLOOD_SEED_LO .. FLOOD_SEED_LO + 3 columns left of the seed: its
    // two word rows are 2 x FLOOD_COLS / 4 aligned 16-byte loads (64 one-word loads of 64 different lines each kept
    // the CU's L1 busy for most of the launch: the tag look-ups of divergent loads, not the memory behind them).
    const int x0 = ((int)sx - FLOOD_SEED_LO) & ~3;  // (two's complement: rounds down below zero too; the mask has MASK_PAD_X zero columns)
    const int sc = (int)sx - x0;                    // the seed's window column, FLOOD_SEED_LO .. FLOOD_SEED_LO + 3
    const uint4 *wq = reinterpret_cast<const uint4 *>(mask + (size_t)((sy - 1u) >> 5) * a.mask_wpr + MASK_PAD_X + x0);
    const uint4 *wq1 = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(wq) + a.mask_wpr);
    uint32_t cand[FLOOD_COLS], comp[FLOOD_COLS];
    // both word rows of the window in flight together (comp[] holds the second one until the two are combined)
#pragma unroll
    for (int q = 0; q < FLOOD_COLS / 4; ++q) {
        const uint4 v = wq[q];
        cand[4 * q] = v.x; cand[4 * q + 1] = v.y; cand[4 * q + 2] = v.z; cand[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < FLOOD_COLS / 4; ++q) {
        const uint4 v = wq1[q];
        comp[4 * q] = v.x; comp[4 * q + 1] = v.y; comp[4 * q + 2] = v.z; comp[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int c = 0; c < FLOOD_COLS; ++c) asm volatile("" : "+v"(cand[c]), "+v"(comp[c]));  // (consumed here, all 2 x FLOOD_COLS words)
    clk.mark(9);  // the window's words have arrived
#pragma unroll
    for (int c = 0; c < FLOOD_COLS; ++c) {
        const unsigned long long two = (unsigned long long)cand[c] | ((unsigned long long)comp[c] << 32);
        cand[c] = (uint32_t)(two >> sh);  // bit r = row sy-1+r of column x0+c
        comp[c] = 0u;
    }
#pragma unroll
    for (int c = FLOOD_SEED_LO; c < FLOOD_SEED_LO + 4; ++c) comp[c] = sc == c ? 2u : 0u;  // the seed: column sx, row sy
    // One left-to-right and one right-to-left sweep, then the fixed-point test.  After a sweep every
    // comp[c] is a union of whole vertical runs of cand[c], so the update rule would change column c
    // exactly if a neighbouring column holds a component pixel next to a candidate of c that is not
    // in comp[c] yet: 4 operations per column instead of a third and fourth sweep (convex blobs are
    // complete after the first pair).
    uint32_t pending;
    do {
#pragma unroll
        for (int c = 0; c < FLOOD_COLS; ++c) {  // left-to-right sweep
            uint32_t s = comp[c];
            if (c > 0) s |= comp[c - 1];
            if (c < FLOOD_COLS - 1) s |= comp[c + 1];
            comp[c] = fill_runs(s & cand[c], cand[c]);
        }
#pragma unroll
        for (int c = FLOOD_COLS - 1; c >= 0; --c) {  // right-to-left sweep
            uint32_t s = comp[c];
            if (c > 0) s |= comp[c - 1];
            if (c < FLOOD_COLS - 1) s |= comp[c + 1];
            comp[c] = fill_runs(s & cand[c], cand[c]);
        }
        pending = 0u;
#pragma unroll
        for (int c = 0; c < FLOOD_COLS; ++c) {
            uint32_t nb = 0u;
            if (c > 0) nb |= comp[c - 1];
            if (c < FLOOD_COLS - 1) nb |= comp[c + 1];
            pending |= nb & cand[c] & ~comp[c];
        }
    } while (pending);
    uint32_t all = 0u, left_of_seed = 0u;
#pragma unroll
    for (int c = 0; c < FLOOD_COLS; ++c) {
        all |= comp[c];
        if (c < FLOOD_SEED_LO) left_of_seed |= comp[c];
        else if (c < FLOOD_SEED_LO + 3) left_of_seed |= c < sc ? comp[c] : 0u;
    }
    // a pixel of the component precedes the seed in raster order -> not the canonical seed
    const bool canonical = !((all & 1u) || (left_of_seed & 2u));
    if (canonical) {
        if ((all >> 31) || comp[0] || comp[FLOOD_COLS - 1]) {
            return FLOOD_BIG;  // may continue outside the window: second tier
        } else {
            uint32_t cnt = 0, sumx = 0, sumy = 0;
#pragma unroll
            for (int c = 0; c < FLOOD_COLS; ++c) {
                const uint32_t w = comp[c];
                const uint32_t nc = (uint32_t)__popc(w);
                cnt += nc;
                sumy += bitpos_sum(w);
                sumx += nc * (uint32_t)c;
                // (column by column: left to itself the scheduler starts all 32 columns' popcounts at once and spills at 128 registers)
                if ((c & 3) == 3) asm volatile("" : "+v"(cnt), "+v"(sumx), "+v"(sumy));
            }
            sumx += cnt * (uint32_t)x0;  // window column 0 is image column x0 (mod 2^32 arithmetic)
            sumy += cnt * (sy - 1u);     // window row 0 is image row sy-1
            cnt_out = cnt;
            sumx_out = sumx;
            sumy_out = sumy;
            return FLOOD_CLUSTER;
        }
    }
    return FLOOD_NONE;
}

// ------------------------------------------------------------------------------------------
// K3g: generic fallback, run only for frames flagged FLAG_BIG_CLUSTER (every workgroup checks the
// flag and leaves at once otherwise).  Works for components of any size and shape.
//   phase 1   mask -> candidate list (pixel | left<<30 | up<<31), slot plane
//   phase 2   lock-free union-find; links go from the larger slot to the smaller; every access
//             to parent[] is an agent-scope atomic
//   phase 3   per-root integer sums and smallest pixel; root list
//   phase 4   roots -> cluster records (replacing the flood path's records of the frame)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool frame_is_generic(const ChainArgs &a, const FrameCounters &ctr)
{
    return a.force_generic || (ctr.flags & FLAG_BIG_CLUSTER);
}

// One 1024-thread workgroup per flagged frame runs the four phases back to back; phases are
// separated by a workgroup barrier plus agent-scope release / acquire fences, because the
// phases communicate through global memory (atomics execute at L2 / memory and do not refresh
// this CU's L1).  Unflagged frames cost one launch of workgroups that return at once.
__device__ __forceinline__ void phase_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ uint32_t uf_find_atomic(uint32_t *parent, uint32_t x)
{
    for (;;) {
        uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        x = p;
    }
}
__device__ __forceinline__ void uf_unite(uint32_t *parent, uint32_t x, uint32_t y)
{
    for (;;) {
        x = uf_find_atomic(parent, x);
        y = uf_find_atomic(parent, y);
        if (x == y) return;
        if (x < y) {
            uint32_t tmp = x; x = y; y = tmp;
        }
        uint32_t old = atomicCAS(&parent[x], x, y);
        if (old == x) return;
        x = old;
    }
}

// The generic clustering of one frame by one workgroup (any size): four phases separated by
// workgroup barriers + agent-scope fences.  Replaces the frame's cluster records.
__device__ __forceinline__ void generic_frame(const ChainArgs &a, int frame)
{
    FrameCounters &ctr = a.ctr[frame];
    const uint32_t T = blockDim.x, t = threadIdx.x;
    const uint32_t *mask = a.mask + (size_t)frame * (size_t)a.mask_plane;
    const int wpr = a.mask_wpr, W = a.W;
    const size_t base = (size_t)frame * a.cap_cand;
    uint32_t *parent = a.parent + base;
    uint32_t *slot_plane = a.slot_plane + (size_t)frame * (size_t)a.plane;

    // phase 1: mask -> candidate list (pixel | left<<30 | up<<31), slot plane
    const long long total = (long long)((a.H + 31) >> 5) * W;
    for (long long i = t; i < total; i += T) {
        const int yb = (int)(i / W), x = (int)(i % W);
        const uint32_t *wp = mask + (size_t)yb * wpr + MASK_PAD_X + x;
        uint32_t m = wp[0];
        if (!m) continue;
        const uint32_t lm = wp[-1];                                       // candidate to the left
        const uint32_t um = (m << 1) | (yb > 0 ? (wp[-wpr] >> 31) : 0u);  // candidate above
        while (m) {
            const int b = __ffs(m) - 1;
            m &= m - 1;
            const uint32_t p = (uint32_t)(yb * 32 + b) * (uint32_t)W + (uint32_t)x;
            const uint32_t slot = atomicAdd(&ctr.n_cand, 1u);
            if (slot < a.cap_cand) {
                const size_t o = base + slot;
                a.cand[o] = p | (((lm >> b) & 1u) << 30) | (((um >> b) & 1u) << 31);
                a.parent[o] = slot;
                a.sumx[o] = 0ull;
                a.sumy[o] = 0ull;
                a.cnt[o] = 0u;
                a.minidx[o] = 0xffffffffu;
                slot_plane[p] = slot;
            } else {
                atomicOr(&ctr.flags, FLAG_CAND_OVERFLOW);
            }
        }
    }
    phase_barrier();
    const uint32_t n = __hip_atomic_load(&ctr.n_cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n > a.cap_cand) {  // overflow: slot_plane is incomplete; the frame is reported, not processed
        if (t == 0) ctr.n_clusters = ctr.n_clusters2 = 0u;
        return;
    }
    // phase 2: lock-free union-find; links go from the larger slot to the smaller
    for (uint32_t s = t; s < n; s += T) {
        const uint32_t e = a.cand[base + s];
        const uint32_t p = e & 0x3fffffffu;
        if (e & 0x40000000u) uf_unite(parent, s, slot_plane[p - 1]);
        if (e & 0x80000000u) uf_unite(parent, s, slot_plane[p - a.W]);
    }
    phase_barrier();
    // phase 3: per-root integer sums and smallest pixel; root list
    for (uint32_t s = t; s < n; s += T) {
        const uint32_t p = a.cand[base + s] & 0x3fffffffu;
        const uint32_t r = uf_find_atomic(parent, s);
        atomicAdd(&a.sumx[base + r], (unsigned long long)(p % (uint32_t)a.W));  // 64 bit: a component of millions of
        atomicAdd(&a.sumy[base + r], (unsigned long long)(p / (uint32_t)a.W));  // pixels exceeds 2^32
        atomicAdd(&a.cnt[base + r], 1u);
        atomicMin(&a.minidx[base + r], p);
        if (r == s) {
            uint32_t i = atomicAdd(&ctr.n_roots, 1u);
            if (i < a.cap_roots) a.roots[(size_t)frame * a.cap_roots + i] = s;
            else atomicOr(&ctr.flags, FLAG_ROOT_OVERFLOW);
        }
    }
    phase_barrier();
    // phase 4: roots -> cluster records (replacing the flood path's records of this frame)
    const uint32_t nr = min(__hip_atomic_load(&ctr.n_roots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.cap_roots);
    for (uint32_t i = t; i < nr; i += T) {
        const uint32_t s = __hip_atomic_load(&a.roots[(size_t)frame * a.cap_roots + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const size_t q = (size_t)frame * a.cap_roots + i;
        a.clu_key[q] = __hip_atomic_load(&a.minidx[base + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.clu_cnt[q] = __hip_atomic_load(&a.cnt[base + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the cluster record keeps 32 bits; a sum beyond that saturates (>= 2^24: K4 raises FLAG_CENTROID_INEXACT)
        const unsigned long long sx64 = __hip_atomic_load(&a.sumx[base + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long sy64 = __hip_atomic_load(&a.sumy[base + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.clu_sx[q] = sx64 > 0xffffffffull ? 0xffffffffu : (uint32_t)sx64;
        a.clu_sy[q] = sy64 > 0xffffffffull ? 0xffffffffu : (uint32_t)sy64;
    }
    if (t == 0) {
        ctr.n_clusters = nr;
        ctr.n_clusters2 = 0u;  // whatever the fast path's second tier appended is void
    }
}


// Debug only (agx_debug_fetch AGX_DBG_RESP): the response plane K2 thresholds, materialised.
__global__ void k_debug_resp(const float *__restrict__ blur, float *__restrict__ resp, int W, int H)
{
    const long long n = (long long)W * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        float d = 0.0f;
        if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
            const float *c = blur + i;
            const float v11 = c[-W - 1], v12 = c[-W], v13 = c[-W + 1];
            const float v21 = c[-1], v22 = c[0], v23 = c[1];
            const float v31 = c[W - 1], v32 = c[W], v33 = c[W + 1];
            const float t22 = v22 * 2.0f;
            const float lxx = (v21 - t22) + v23;
            const float lyy = (v12 - t22) + v32;
            const float lxy = (((v13 - v11) + v31) - v33) * 0.25f;
            d = lxx * lyy - lxy * lxy;
        }
        resp[i] = d;
    }
}

// ------------------------------------------------------------------------------------------
// K4: rochade_refine, detector.rs:265-359, one cluster per lane.
// ------------------------------------------------------------------------------------------
// rochade_refine of cluster s of `frame` (detector.rs:265-359).  Appends a RefinedRec through the
// counters given (the global per-frame counters, or a workgroup's LDS copies).
// Core: the cluster with first pixel `key`, cn pixels and the integer coordinate sums sx, sy; the f32 centroid
// (detector.rs:427) comes back in cx, cy (the caller keeps it in the cluster record for agx_debug_fetch).
// Where a refined record goes.  RecSinkGlobal: the frame's global list through the frame's global counters (k_flood_refine,
// the generic path).  RecSinkLds (k_sparse_frame): the workgroup's counters in LDS, the first TAIL_CAP records in LDS as six
// arrays of TAIL_CAP words, every record also in the frame's global list (agx_debug_fetch, and the large-list emission).
constexpr uint32_t TAIL_CAP = 1024;
struct RecSinkGlobal {
    uint32_t *n_refined, *max_k_bits, *flags;
    __device__ __forceinline__ void put(const ChainArgs &a, int frame, uint32_t o, const uint32_t (&f)[6]) const
    {
        // The record is read by another launch (k_rare) or, on the generic path, by other waves of the workgroup behind an
        // agent-scope fence: agent-scope stores.
        uint32_t *rec = reinterpret_cast<uint32_t *>(a.refined + (size_t)frame * a.cap_roots + o);
#pragma unroll
        for (int q = 0; q < 6; ++q) __hip_atomic_store(rec + q, f[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
struct RecSinkLds {
    uint32_t *n_refined, *max_k_bits, *flags;  // LDS words
    uint32_t *rec;                             // LDS [6][TAIL_CAP]
    __device__ __forceinline__ void put(const ChainArgs &a, int frame, uint32_t o, const uint32_t (&f)[6]) const
    {
        if (o < TAIL_CAP) {
#pragma unroll
            for (int q = 0; q < 6; ++q) rec[q * TAIL_CAP + o] = f[q];
        }
        uint32_t *g = reinterpret_cast<uint32_t *>(a.refined + (size_t)frame * a.cap_roots + o);
#pragma unroll
        for (int q = 0; q < 6; ++q) g[q] = f[q];  // (read by this workgroup behind a barrier, or by the host)
    }
};

template <bool VEC, typename CLK, typename SINK>
__device__ __forceinline__ void refine_values(const ChainArgs &a, const RefineConsts &rc, int frame, const float *img, int W, int H,
                                              uint32_t key, uint32_t cn, uint32_t sx, uint32_t sy, const SINK &sink, float &cx,
                                              float &cy, CLK &clk)
{
    if (sx >= (1u << 24) || sy >= (1u << 24)) atomicOr(sink.flags, FLAG_CENTROID_INEXACT);
    const float fn = (float)cn;
    const float initial_x = (float)sx / fn;  // detector.rs:427
    const float initial_y = (float)sy / fn;
    cx = initial_x;
    cy = initial_y;
    const float rxf = roundf(initial_x), ryf = roundf(initial_y);
    const int round_x = (int)rxf, round_y = (int)ryf;
    if (round_y - 4 < 0 || round_y + 4 >= H || round_x - 4 < 0 || round_x + 4 >= W) return;
    // The 9x9 window is streamed row by row through 25 running sums.  Patch value (r,c)
    // receives its 25 taps in the reference's order (:283-297): window rows r..r+4 arrive in
    // ascending order and the 5 taps of a row are added left to right.  Patch row r is
    // complete after window row r+4 and is then folded into the 6 parameter sums (:321-328,
    // i = r*5+c ascending).
    const int wx0 = round_x - 4;
    const int al = VEC ? (wx0 & 3) : 0;
    const float *win = img + (size_t)(round_y - 4) * W + (wx0 - al);
    float conv[25];
#pragma unroll
    for (int q = 0; q < 25; ++q) conv[q] = 0.0f;
    float prm[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // The window's rows are fetched in two batches (rows 0..4, rows 5..8) with all loads of a batch in flight
    // together: fetched row by row -- three loads, a wait, the row's arithmetic -- a lane went through nine
    // dependent memory round trips, which is what a wave that starts late (or runs while the memory system is not
    // saturated by the others) spends its life on.
    constexpr int NQ = VEC ? 3 : 9;  // loads per row: 3 x 16 bytes, or 9 floats
    typedef float rowload_t __attribute__((ext_vector_type(VEC ? 4 : 1)));
    rowload_t raw[5][NQ];
    auto fetch = [&](int first, int n) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (k < n) {
#pragma unroll
                for (int e = 0; e < NQ; ++e)
                    raw[k][e] = *reinterpret_cast<const rowload_t *>(win + (size_t)(first + k) * W + (VEC ? 4 * e : e));
            }
        asm volatile("" ::: "memory");  // every load of the batch is issued before the first one is waited for
    };
    fetch(0, 5);
    if (CLK::on) {  // (the timed instantiation only: the wait is normally row by row)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        clk.mark(11);  // rows 0..4 of the window have arrived
    }
#pragma unroll
    for (int wr = 0; wr < 9; ++wr) {
        if (wr == 5) {
            clk.mark(12);  // arithmetic of rows 0..4
            fetch(5, 4);
            if (CLK::on) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                clk.mark(13);  // rows 5..8 have arrived
            }
        }
        float v[9];
        const int k = wr < 5 ? wr : wr - 5;
        if (VEC) {
            // three aligned 16-byte loads cover the 9 floats; shift by the misalignment
            const rowload_t q0 = raw[k][0], q1 = raw[k][1 % NQ], q2 = raw[k][2 % NQ];
            const bool s1 = (al & 1) != 0, s2 = (al & 2) != 0;
            // shift left by 1 if s1, then by 2 if s2 (all indices compile-time)
            const float u0 = s1 ? q0[1 % (VEC ? 4 : 1)] : q0[0], u1 = s1 ? q0[2 % (VEC ? 4 : 1)] : q0[1 % (VEC ? 4 : 1)], u2 = s1 ? q0[3 % (VEC ? 4 : 1)] : q0[2 % (VEC ? 4 : 1)], u3 = s1 ? q1[0] : q0[3 % (VEC ? 4 : 1)];
            const float u4 = s1 ? q1[1 % (VEC ? 4 : 1)] : q1[0], u5 = s1 ? q1[2 % (VEC ? 4 : 1)] : q1[1 % (VEC ? 4 : 1)], u6 = s1 ? q1[3 % (VEC ? 4 : 1)] : q1[2 % (VEC ? 4 : 1)], u7 = s1 ? q2[0] : q1[3 % (VEC ? 4 : 1)];
            const float u8 = s1 ? q2[1 % (VEC ? 4 : 1)] : q2[0], u9 = s1 ? q2[2 % (VEC ? 4 : 1)] : q2[1 % (VEC ? 4 : 1)], u10 = s1 ? q2[3 % (VEC ? 4 : 1)] : q2[2 % (VEC ? 4 : 1)];
            v[0] = s2 ? u2 : u0; v[1] = s2 ? u3 : u1; v[2] = s2 ? u4 : u2; v[3] = s2 ? u5 : u3; v[4] = s2 ? u6 : u4;
            v[5] = s2 ? u7 : u5; v[6] = s2 ? u8 : u6; v[7] = s2 ? u9 : u7; v[8] = s2 ? u10 : u8;
        } else {
#pragma unroll
            for (int e = 0; e < 9; ++e) v[e] = raw[k][e % NQ][0];
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int pr = wr - r;
            if (pr < 0 || pr > 4) continue;
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int pc = 0; pc < 5; ++pc) conv[r * 5 + c] = conv[r * 5 + c] + v[c + pc] * rc.cone[pr * 5 + pc];
        }
        if (wr >= 4) {
            const int r = wr - 4;
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int j = 0; j < 6; ++j) prm[j] = prm[j] + rc.pmat[(r * 5 + c) * 6 + j] * conv[r * 5 + c];
        }
    }
    if (CLK::on) {
        asm volatile("" : "+v"(prm[0]), "+v"(prm[1]), "+v"(prm[2]), "+v"(prm[3]), "+v"(prm[4]), "+v"(prm[5]));
        clk.mark(16);  // arithmetic of rows 5..8
    }
    const float a1 = prm[0], a2 = prm[1], a3 = prm[2], a4 = prm[3], a5 = prm[4];
    const float fxx = 2.0f * a1, fyy = 2.0f * a3, fxy = a2;
    const float d = fxx * fyy - fxy * fxy;
    if (!(d < 0.0f)) return;
    // find_xy(2a1, a2, a4, a2, 2a3, a5), math_util.rs:5-12: 2x2 LU with row pivoting
    float x0, y0;
    {
        const float A0 = 2.0f * a1, B0 = a2, R0 = -a4;
        const float A1 = a2, B1 = 2.0f * a3, R1 = -a5;
        float pa, pb, pr_, qa, qb, qr;
        if (fabsf(A1) > fabsf(A0)) {
            pa = A1; pb = B1; pr_ = R1; qa = A0; qb = B0; qr = R0;
        } else {
            pa = A0; pb = B0; pr_ = R0; qa = A1; qb = B1; qr = R1;
        }
        const float l = qa / pa;
        const float u22 = qb - l * pb;
        const float y2 = qr - l * pr_;
        y0 = y2 / u22;
        x0 = (pr_ - pb * y0) / pa;
    }
    if (!(fabsf(x0) <= 1.0f && fabsf(y0) <= 1.0f)) return;
    const float c5 = (a1 + a3) / 2.0f;
    const float c4 = (a1 - a3) / 2.0f;
    const float c3 = a2 / 2.0f;
    const float k = sqrtf(c4 * c4 + c3 * c3);
    if (!(fabsf(c5) < k)) return;
    const float PI_F = 3.14159274101257324219f;
    const float phi = acosf(-c5 / k) / 2.0f / PI_F * 180.0f;
    const float theta = atan2f(c3, c4) / 2.0f / PI_F * 180.0f;
    if (CLK::on) {
        float th = theta, ph = phi;
        asm volatile("" : "+v"(th), "+v"(ph));
        clk.mark(17);  // the fit: divisions, sqrt, acos, atan2 (lanes that pass)
    }
    uint32_t o = atomicAdd(sink.n_refined, 1u);
    if (CLK::on) {
        asm volatile("" : "+v"(o));
        clk.mark(15);  // the record's index (atomic round trip)
    }
    // o < n_clusters <= cap_roots -- unless the frame's cluster list has overflowed (the fused flood + refine kernel
    // refines before it knows its record index): such a frame is void as a whole and nothing is stored for it
    if (o >= a.cap_roots) return;
    const uint32_t f[6] = {key, __float_as_uint(rxf + x0), __float_as_uint(ryf + y0),
                           __float_as_uint(k),  __float_as_uint(theta),     __float_as_uint(phi)};
    sink.put(a, frame, o, f);
    atomicMax(sink.max_k_bits, __float_as_uint(k));
}

// rochade_refine of cluster record s of `frame` as it stands in the cluster table.
template <bool VEC>
__device__ __forceinline__ void refine_cluster(const ChainArgs &a, const RefineConsts &rc, int frame, size_t cbase,
                                               const float *img, int W, int H, uint32_t s, uint32_t *n_refined,
                                               uint32_t *max_k_bits)
{
    float cx, cy;
    NoClock clk;
    const RecSinkGlobal sink{n_refined, max_k_bits, &a.ctr[frame].flags};
    refine_values<VEC>(a, rc, frame, img, W, H, a.clu_key[cbase + s], a.clu_cnt[cbase + s], a.clu_sx[cbase + s], a.clu_sy[cbase + s],
                       sink, cx, cy, clk);
    a.clu_sx[cbase + s] = __float_as_uint(cx);  // kept for agx_debug_fetch
    a.clu_sy[cbase + s] = __float_as_uint(cy);
}

// ------------------------------------------------------------------------------------------
// K3: flood + refine in one launch.  One seed per lane: the bit-parallel flood of its 32 x 32 window, and --
// where the seed turns out to be the first pixel of a component that stays inside the window -- the
// component's cluster record and its rochade_refine right behind it, from the sums still in registers.
// (Two launches until round 3: the flood, bound by its slowest waves with a quarter of the wave slots in use,
// and the refinement, bound by the random sector reads of the 9 x 9 windows; fused, the floods of some waves
// run under the window reads of the others.)  Seeds whose component may leave the window are flooded again by
// the whole wave (128 x 64 window) before the refinement; what leaves that window too sends the frame to k_rare.
// ------------------------------------------------------------------------------------------
template <bool VEC, typename CLK>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AGX_FLOOD_WPE, 8))) k_flood_refine(ChainArgs a, RefineConsts rc)
{
    const WaveTimer wt(a, K_FLOOD_REFINE);
    const FrameSlot fs = frame_slot(a.n_frames, true);  // latest-written blur planes first (cache)
    const int frame = fs.frame;
    FrameCounters &ctr = a.ctr[frame];
    if (a.force_generic) return;  // whole frame: k_rare
    const size_t cbase = (size_t)frame * a.cap_roots;
    const int lane = threadIdx.x;
    // the frame's flags and seed count and this lane's first seed in ONE round trip (the seed is fetched before the
    // count is known: any index below cap_roots is inside the list)
    CLK clk;
    clk.start(ctr.stats);
    uint32_t flags0 = ctr.flags, n_seeds0 = ctr.n_seeds;
    const uint32_t i0 = fs.slot * 64u + (uint32_t)lane;
    uint32_t p0 = i0 < a.cap_roots ? a.seeds[cbase + i0] : 0u;
    asm volatile("" : "+v"(flags0), "+v"(n_seeds0), "+v"(p0));  // (all three consumed here: no load is left behind a branch)
    if (flags0 & FLAG_CAND_OVERFLOW) return;  // reported
    const uint32_t n = min(n_seeds0, a.cap_roots);
    const uint32_t *mask = a.mask + (size_t)frame * (size_t)a.mask_plane;
    const float *img = a.blur + (size_t)frame * (size_t)a.plane;
    const uint32_t W = (uint32_t)a.W;
    for (uint32_t base = fs.slot * 64u; base < n; base += fs.n_slots * 64u) {  // wave-uniform trip count
        const uint32_t i = base + (uint32_t)lane;
        uint32_t p = 0u, cnt = 0, sumx = 0, sumy = 0;
        int what = FLOOD_NONE;
        if (CLK::on && lane == 0) atomicAdd(&ctr.stats[19], 1u);  // chunks
        clk.mark(8);  // counters + first seed (or the loop's turn-around)
        if (i < n) {
            p = base == fs.slot * 64u ? p0 : a.seeds[cbase + i];
            what = flood_lane(a, mask, W, p, cnt, sumx, sumy, clk);
        }
        clk.join();
        clk.mark(10);  // the flood's sweeps
        // Second tier: components that may leave the lane's window (about 1 % of the seeds on real frames) are
        // flooded again by the whole wave in a 128 x 64 window, one after the other; the lane that owns the seed
        // takes the result and refines the cluster together with everyone else below.  (A component that leaves
        // that window too flags the frame: k_rare clusters it again by the generic path.)
        unsigned long long big = __ballot(what == FLOOD_BIG);
        if (big && lane == 0) atomicAdd(&ctr.n_big, (uint32_t)__popcll(big));  // (informational: agx_debug_fetch)
        while (big) {  // wave-uniform
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1ull;
            uint32_t c2 = 0, x2 = 0, y2 = 0;
            const int w2 = wave_flood_128x64(a, &ctr.flags, mask, (uint32_t)__builtin_amdgcn_readlane((int)p, src), lane, c2, x2, y2);
            if (lane == src) {
                what = w2 == FLOOD_CLUSTER ? FLOOD_CLUSTER : FLOOD_NONE;
                cnt = c2;
                sumx = x2;
                sumy = y2;
            }
        }
        clk.mark(14);  // second tier
        if (CLK::on) {
            const unsigned long long act = __ballot(what == FLOOD_CLUSTER);
            if (lane == 0) atomicAdd(&ctr.stats[18], (uint32_t)__popcll(act));  // clusters refined
        }
        if (what == FLOOD_CLUSTER) {
            // the index of the cluster record is only needed when the record is written, behind the refinement: the
            // atomic's round trip runs under the window loads (a frame whose cluster list overflows is void as a whole)
            const uint32_t o = atomicAdd(&ctr.n_clusters, 1u);
            float cx, cy;
            const RecSinkGlobal sink{&ctr.n_refined, &ctr.max_k_bits, &ctr.flags};
            refine_values<VEC>(a, rc, frame, img, a.W, a.H, p, cnt, sumx, sumy, sink, cx, cy, clk);
            if (o < a.cap_roots) {
                a.clu_key[cbase + o] = p;
                a.clu_cnt[cbase + o] = cnt;
                a.clu_sx[cbase + o] = __float_as_uint(cx);  // kept for agx_debug_fetch
                a.clu_sy[cbase + o] = __float_as_uint(cy);
            } else {
                atomicOr(&ctr.flags, FLAG_ROOT_OVERFLOW);
            }
        }
        clk.join();
        clk.mark(7);  // record stores, lanes without a record
    }
}

// ------------------------------------------------------------------------------------------
// K4: filter (detector.rs:436-445) and emission in the reference's order = ascending first
// (smallest) pixel index of the cluster.  One workgroup per frame; bitonic sort in LDS.
// ------------------------------------------------------------------------------------------
// The k / phi filter (detector.rs:436-445) and the ordered emission of one frame by one workgroup:
// n refined records, largest k as raw bits; keys / idxs are LDS arrays of lds_entries words each.
__device__ __forceinline__ void filter_sort_emit(const ChainArgs &a, int frame, uint32_t n, uint32_t max_k_bits,
                                                 uint32_t *keys, uint32_t *idxs, uint32_t lds_entries, uint32_t *s_count,
                                                 uint32_t *s_offset, uint32_t *s_fits)
{
    const uint32_t t = threadIdx.x, T = blockDim.x;
    FrameCounters &ctr = a.ctr[frame];
    // records written by other workgroups: every read bypasses this CU's L1 / this XCD's L2 (sc1)
    uint32_t *rec = reinterpret_cast<uint32_t *>(a.refined + (size_t)frame * a.cap_roots);
    auto rec_u = [&](uint32_t i, int field) { return __hip_atomic_load(rec + (size_t)i * 6 + field, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (t == 0) *s_count = 0;
    __syncthreads();
    uint32_t nf = 0;
    bool ok = true;
    if (n != 0) {  // detector.rs:432-434: nothing refined -> empty result
        const float s_max_k = __uint_as_float(max_k_bits) / 10.0f;
        for (uint32_t i = t; i < n; i += T) {
            const float k = __uint_as_float(rec_u(i, 3)), phi = __uint_as_float(rec_u(i, 5));
            if (k >= s_max_k && phi >= a.min_angle && phi <= a.max_angle) {
                uint32_t o = atomicAdd(s_count, 1u);
                if (o < lds_entries) {
                    keys[o] = rec_u(i, 0);
                    idxs[o] = i;
                }
            }
        }
        __syncthreads();
        nf = *s_count;
        uint32_t need = 1;
        while (need < nf) need <<= 1;  // the bitonic network pads to a power of two
        ok = nf <= a.cap_out && need <= lds_entries;
    }
    if (ok && nf) {
        uint32_t np2 = 1;
        while (np2 < nf) np2 <<= 1;
        for (uint32_t i = nf + t; i < np2; i += T) {
            keys[i] = 0xffffffffu;
            idxs[i] = 0;
        }
        __syncthreads();
        for (uint32_t k2 = 2; k2 <= np2; k2 <<= 1) {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                for (uint32_t i = t; i < np2; i += T) {
                    uint32_t l = i ^ j;
                    if (l > i) {
                        const bool asc = (i & k2) == 0;
                        uint32_t ki = keys[i], kl = keys[l];
                        if ((ki > kl) == asc) {
                            keys[i] = kl; keys[l] = ki;
                            uint32_t ti = idxs[i]; idxs[i] = idxs[l]; idxs[l] = ti;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    if (t == 0) {
        uint32_t off = 0, fits = 1;
        if (ok && nf) {
            off = atomicAdd(a.total_out, nf);
            if (off + nf > a.out_total_cap) fits = 0;  // caller's buffer is full
        }
        if (!ok || !fits) atomicOr(&ctr.flags, FLAG_OUT_OVERFLOW);
        *s_offset = off;
        *s_fits = fits;
        ctr.n_out = nf;
        ctr.out_offset = off;
        if (a.frame_table) {
            uint32_t *row = a.frame_table + (size_t)frame * 4;
            const uint32_t flags = __hip_atomic_load(&ctr.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool bad = (flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) != 0;
            row[0] = bad ? 0u : nf;
            row[1] = off;
            row[2] = flags;
            row[3] = __hip_atomic_load(&ctr.n_clusters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                     __hip_atomic_load(&ctr.n_clusters2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!ok || !nf || !*s_fits) return;
    float *out = a.out + (size_t)*s_offset * 5;
    for (uint32_t i = t; i < nf; i += T) {
        const uint32_t src = idxs[i];
#pragma unroll
        for (int q = 0; q < 5; ++q) out[i * 5 + q] = __uint_as_float(rec_u(src, q + 1));
    }
}

// Filter (detector.rs:436-445) and ordered emission of one frame with up to TAIL_CAP refined records by the
// frame's 1024-thread workgroup of k_rare: thread t holds record t, rejected records get the key 0xffffffff, and a
// surviving record's output position is its rank = the number of smaller keys (keys are distinct: the first pixel
// of distinct clusters) -- no sort passes, one memory round trip for the records and one for the output.
__device__ __forceinline__ void emit_wide(const ChainArgs &a, int frame, uint32_t n, uint32_t max_k_bits, uint32_t *keys,
                                          uint32_t *s_misc)
{
    const uint32_t t = threadIdx.x;  // 1024 threads, n <= TAIL_CAP
    FrameCounters &ctr = a.ctr[frame];
    const uint32_t *rec = reinterpret_cast<const uint32_t *>(a.refined + (size_t)frame * a.cap_roots) + (size_t)t * 6;
    const float s_max_k = __uint_as_float(max_k_bits) / 10.0f;  // detector.rs:436
    uint32_t f[6] = {0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    bool pass = false;
    if (t < n) {
        // (written by k_flood_refine, the previous launch -- or by this workgroup's generic path behind an agent-scope fence)
#pragma unroll
        for (int q = 0; q < 6; ++q) f[q] = __hip_atomic_load(rec + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float k = __uint_as_float(f[3]), phi = __uint_as_float(f[5]);
        pass = k >= s_max_k && phi >= a.min_angle && phi <= a.max_angle;
    }
    keys[t] = pass ? f[0] : 0xffffffffu;
    if (t < 4) keys[TAIL_CAP + t] = 0xffffffffu;  // (padding for the 16-byte reads of the rank loop)
    const uint32_t nf = (uint32_t)__syncthreads_count(pass ? 1 : 0);
    // the frame's place in the compact output: one atomic, issued now; its round trip runs under the rank loop
    const bool ok = nf <= a.cap_out;
    uint32_t off = 0;
    if (t == 0 && ok && nf) off = atomicAdd(a.total_out, nf);
    const uint32_t n4 = (n + 3u) >> 2;
    uint32_t rank = 0;
    if (pass) {
        for (uint32_t j = 0; j < n4; ++j) {  // every lane reads the same 16 bytes: LDS broadcast
            const uint4 q = reinterpret_cast<const uint4 *>(keys)[j];
            rank += (q.x < f[0] ? 1u : 0u) + (q.y < f[0] ? 1u : 0u) + (q.z < f[0] ? 1u : 0u) + (q.w < f[0] ? 1u : 0u);
        }
    }
    if (t == 0) {
        uint32_t fits = 1;
        if (ok && nf && off + nf > a.out_total_cap) fits = 0;  // caller's buffer is full
        if (!ok || !fits) atomicOr(&ctr.flags, FLAG_OUT_OVERFLOW);
        s_misc[0] = off;
        s_misc[1] = (ok && fits) ? 1u : 0u;
        ctr.n_out = nf;
        ctr.out_offset = off;
        if (a.frame_table) {
            uint32_t *row = a.frame_table + (size_t)frame * 4;
            const uint32_t flags = __hip_atomic_load(&ctr.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool bad = (flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) != 0;
            row[0] = bad ? 0u : nf;
            row[1] = off;
            row[2] = flags;
            row[3] = __hip_atomic_load(&ctr.n_clusters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                     __hip_atomic_load(&ctr.n_clusters2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!pass || !s_misc[1]) return;
    float *out = a.out + (size_t)s_misc[0] * 5;
#pragma unroll
    for (int q = 0; q < 5; ++q) out[(size_t)rank * 5 + q] = __uint_as_float(f[q + 1]);
}

// Filter and ordered emission of one frame whose refined records (n <= TAIL_CAP of them) are in LDS as six arrays of
// TAIL_CAP words (k_sparse_frame): as emit_wide, without a memory round trip for the records.  flags_lds: the frame's flag
// word in LDS; n_clusters: for the frame table.
__device__ __forceinline__ void emit_wide_lds(const ChainArgs &a, int frame, uint32_t n, uint32_t max_k_bits, const uint32_t *rec,
                                              uint32_t *keys, uint32_t *s_misc, uint32_t *flags_lds, uint32_t n_clusters)
{
    const uint32_t t = threadIdx.x;  // 1024 threads, n <= TAIL_CAP
    FrameCounters &ctr = a.ctr[frame];
    const float s_max_k = __uint_as_float(max_k_bits) / 10.0f;  // detector.rs:436
    uint32_t f[6] = {0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    bool pass = false;
    if (t < n) {
#pragma unroll
        for (int q = 0; q < 6; ++q) f[q] = rec[q * TAIL_CAP + t];
        const float k = __uint_as_float(f[3]), phi = __uint_as_float(f[5]);
        pass = k >= s_max_k && phi >= a.min_angle && phi <= a.max_angle;
    }
    keys[t] = pass ? f[0] : 0xffffffffu;
    if (t < 4) keys[TAIL_CAP + t] = 0xffffffffu;  // (padding for the 16-byte reads of the rank loop)
    const uint32_t nf = (uint32_t)__syncthreads_count(pass ? 1 : 0);
    // the frame's place in the compact output: one atomic, issued now; its round trip runs under the rank loop
    const bool ok = nf <= a.cap_out;
    uint32_t off = 0;
    if (t == 0 && ok && nf) off = atomicAdd(a.total_out, nf);
    const uint32_t n4 = (n + 3u) >> 2;
    uint32_t rank = 0;
    if (pass) {
        for (uint32_t j = 0; j < n4; ++j) {  // every lane reads the same 16 bytes: LDS broadcast
            const uint4 q = reinterpret_cast<const uint4 *>(keys)[j];
            rank += (q.x < f[0] ? 1u : 0u) + (q.y < f[0] ? 1u : 0u) + (q.z < f[0] ? 1u : 0u) + (q.w < f[0] ? 1u : 0u);
        }
    }
    if (t == 0) {
        uint32_t fits = 1;
        if (ok && nf && off + nf > a.out_total_cap) fits = 0;  // caller's buffer is full
        if (!ok || !fits) *flags_lds |= FLAG_OUT_OVERFLOW;
        s_misc[0] = off;
        s_misc[1] = (ok && fits) ? 1u : 0u;
        ctr.n_out = nf;
        ctr.out_offset = off;
        if (a.frame_table) {
            uint32_t *row = a.frame_table + (size_t)frame * 4;
            const uint32_t flags = *flags_lds;
            const bool bad = (flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) != 0;
            row[0] = bad ? 0u : nf;
            row[1] = off;
            row[2] = flags;
            row[3] = n_clusters;
        }
    }
    __syncthreads();
    if (!pass || !s_misc[1]) return;
    float *out = a.out + (size_t)s_misc[0] * 5;
#pragma unroll
    for (int q = 0; q < 5; ++q) out[(size_t)rank * 5 + q] = __uint_as_float(f[q + 1]);
}

// The rare work and the emission of one frame from the frame's GLOBAL lists and counters, by a 1024-thread workgroup:
//   - for a frame where a component left the flood windows (FLAG_BIG_CLUSTER): clusters it again by the generic
//     union-find path and refines it;
//   - filters (detector.rs:436-445) and emits the frame's saddles in the reference's order (emit_wide; lists of
//     more than TAIL_CAP refined records -- FLAG_LARGE_RESULT -- by the large LDS / global-memory sort).
// lds_u: lds_entries * 2 words of LDS (16-byte aligned), s_misc3: three more.
struct FrameHead {  // the frame's flags, the length of its refined list and the largest k
    uint32_t flags, n_refined, max_k_bits;
};
__device__ __forceinline__ FrameHead load_frame_head(const FrameCounters &ctr)
{
    // ONE round trip (one 128-byte line)
    FrameHead h;
    h.flags = __hip_atomic_load(&ctr.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h.n_refined = __hip_atomic_load(&ctr.n_refined, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h.max_k_bits = __hip_atomic_load(&ctr.max_k_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" : "+v"(h.flags), "+v"(h.n_refined), "+v"(h.max_k_bits));  // (all three consumed here: no load is left behind a branch)
    return h;
}

__device__ __forceinline__ void rare_frame(const ChainArgs &a, const RefineConsts &rc, int frame, uint32_t *lds_u, uint32_t lds_entries,
                                           uint32_t *s_misc3, const FrameHead &head)
{
    uint32_t &s_count = s_misc3[0], &s_offset = s_misc3[1], &s_fits = s_misc3[2];
    FrameCounters &ctr = a.ctr[frame];
    uint32_t flags0 = head.flags, n_ref0 = head.n_refined, maxk0 = head.max_k_bits;
    const bool generic = a.force_generic || (flags0 & FLAG_BIG_CLUSTER);
    const size_t cbase = (size_t)frame * a.cap_roots;
    const float *img = a.blur + (size_t)frame * (size_t)a.plane;
    if (generic) {
        generic_frame(a, frame);
        if (threadIdx.x == 0) {  // part of the frame was refined before a second-tier flood flagged it
            __hip_atomic_store(&ctr.n_refined, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctr.max_k_bits, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        phase_barrier();
        const uint32_t flags = __hip_atomic_load(&ctr.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW))) {
            const uint32_t n = min(__hip_atomic_load(&ctr.n_clusters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a.cap_roots);
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
                refine_cluster<false>(a, rc, frame, cbase, img, a.W, a.H, i, &ctr.n_refined, &ctr.max_k_bits);
        }
        phase_barrier();
    }
    if (generic) {  // the generic path has rewritten all three
        flags0 = __hip_atomic_load(&ctr.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n_ref0 = __hip_atomic_load(&ctr.n_refined, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        maxk0 = __hip_atomic_load(&ctr.max_k_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // a frame whose seed / cluster lists overflowed is reported, not emitted (whatever was refined before the overflow is void)
    const bool void_frame = (flags0 & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW)) != 0;
    const uint32_t n_ref = void_frame ? 0u : n_ref0;
    const uint32_t maxk = maxk0;
    if (n_ref <= TAIL_CAP && lds_entries * 2 >= TAIL_CAP + 8) {  // the usual case (lds_entries >= 1024 always: k5_lds_bytes)
        emit_wide(a, frame, n_ref, maxk, lds_u, &s_count);  // (s_count, s_offset: two adjacent words)
        return;
    }
    if (threadIdx.x == 0) atomicOr(&ctr.flags, FLAG_LARGE_RESULT);
    __syncthreads();
    if (n_ref <= lds_entries) {
        filter_sort_emit(a, frame, n_ref, maxk, lds_u, lds_u + lds_entries, lds_entries, &s_count, &s_offset, &s_fits);
    } else {
        // Lists beyond the LDS sort (pure-noise frames of several megapixels; the reference's Vec has no
        // limit): the same filter + bitonic network on two per-frame arrays in global memory that are
        // free by now (the flood seeds and the generic path's root list, cap_roots words each).  All
        // accesses come from this workgroup, i.e. one CU and its write-through L1, so the barriers
        // between the passes order them.
        filter_sort_emit(a, frame, n_ref, maxk, a.seeds + (size_t)frame * a.cap_roots, a.roots + (size_t)frame * a.cap_roots,
                         a.cap_roots, &s_count, &s_offset, &s_fits);
    }
}

// The last launch of a batch also clears the OTHER counter set for the next batch (the two sets alternate): a memset
// between the batches costs a fill kernel and two ~5 us gaps on the stream.  (One workgroup per frame.)
__device__ __forceinline__ void clear_next_counters(const ChainArgs &a, int frame)
{
    if (a.ctr_next && threadIdx.x < 64) {
        reinterpret_cast<uint32_t *>(&a.ctr_next[frame])[threadIdx.x] = 0u;
        if (frame == 0) reinterpret_cast<uint32_t *>(&a.ctr_next[a.n_frames])[threadIdx.x] = 0u;  // the output cursor's record
    }
}

// Large lists (more than TAIL_CAP refined records: pure-noise frames carry 8 000) by SEVERAL workgroups per frame: the keys --
// the clusters' first pixels -- are order-preserving, so the frame is cut into `n_parts` bands of pixel indices; every part
// reads all of the frame's records, filters them (detector.rs:436-445), counts the survivors of every band and keeps its own
// band's (key, index) pairs in LDS; it sorts those (bitonic, a fraction of the list) and writes them behind the bands below
// it.  The frame's place in the compact output is allocated by the part that gets there first (one atomic); the others wait
// for the offset -- for a workgroup that is already running and waits for nobody.  Returns false (workgroup-uniform, the same
// in every part) if a band does not fit LDS: part 0 then emits the frame alone (filter_sort_emit).
__device__ __forceinline__ bool emit_large_split(const ChainArgs &a, int frame, uint32_t part, uint32_t n_parts, uint32_t n, uint32_t max_k_bits,
                                                 uint32_t *keys, uint32_t *idxs, uint32_t lds_entries, uint32_t *s_cnt /*[n_parts + 2]*/)
{
    const uint32_t t = threadIdx.x, T = blockDim.x;
    FrameCounters &ctr = a.ctr[frame];
    uint32_t *rec = reinterpret_cast<uint32_t *>(a.refined + (size_t)frame * a.cap_roots);
    auto rec_u = [&](uint32_t i, int field) { return __hip_atomic_load(rec + (size_t)i * 6 + field, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (t < n_parts + 2) s_cnt[t] = 0u;
    __syncthreads();
    const float s_max_k = __uint_as_float(max_k_bits) / 10.0f;
    const unsigned long long plane = (unsigned long long)a.plane;
    for (uint32_t i = t; i < n; i += T) {
        const float k = __uint_as_float(rec_u(i, 3)), phi = __uint_as_float(rec_u(i, 5));
        if (k >= s_max_k && phi >= a.min_angle && phi <= a.max_angle) {
            const uint32_t key = rec_u(i, 0);
            const uint32_t band = (uint32_t)(((unsigned long long)key * n_parts) / plane);  // key < plane
            const uint32_t o = atomicAdd(&s_cnt[band], 1u);
            if (band == part && o < lds_entries) {
                keys[o] = key;
                idxs[o] = i;
            }
        }
    }
    __syncthreads();
    uint32_t nf = 0, below = 0, biggest = 0;
    for (uint32_t b = 0; b < n_parts; ++b) {
        const uint32_t c = s_cnt[b];
        if (b < part) below += c;
        nf += c;
        biggest = max(biggest, c);
    }
    uint32_t need = 1;
    while (need < biggest) need <<= 1;  // the bitonic network pads to a power of two
    if (need > lds_entries) return false;  // (every part computes the same counts: all of them return)
    const uint32_t mine = s_cnt[part];
    // the frame's place in the compact output: the part that arrives first allocates it
    if (t == 0) {
        uint32_t off = 0, fits = 1;
        const uint32_t won = atomicCAS(&ctr.refine_done, 0u, 1u);
        if (won == 0u) {
            const bool ok = nf <= a.cap_out;
            if (ok && nf) {
                off = atomicAdd(a.total_out, nf);
                if (off + nf > a.out_total_cap) fits = 0;  // caller's buffer is full
            }
            if (!ok || !fits) atomicOr(&ctr.flags, FLAG_OUT_OVERFLOW);
            ctr.n_out = nf;
            ctr.out_offset = off;
            if (a.frame_table) {
                uint32_t *row = a.frame_table + (size_t)frame * 4;
                // (part 0 sets FLAG_LARGE_RESULT in the frame's record; the part that writes the row may be here before that lands)
                const uint32_t flags = __hip_atomic_load(&ctr.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | FLAG_LARGE_RESULT;
                const bool bad = (flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) != 0;
                row[0] = bad ? 0u : nf;
                row[1] = off;
                row[2] = flags;
                row[3] = __hip_atomic_load(&ctr.n_clusters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(&ctr.out_offset, off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctr.refine_done, (ok && fits) ? 2u : 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_cnt[n_parts] = off;
            s_cnt[n_parts + 1] = (ok && fits) ? 1u : 0u;
        } else {
            uint32_t st;
            while ((st = __hip_atomic_load(&ctr.refine_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < 2u) __builtin_amdgcn_s_sleep(2);
            s_cnt[n_parts] = __hip_atomic_load(&ctr.out_offset, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_cnt[n_parts + 1] = st == 2u ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_cnt[n_parts + 1] || !mine) return true;  // the frame does not fit the output (reported), or nothing in this band
    uint32_t np2 = 1;
    while (np2 < mine) np2 <<= 1;
    for (uint32_t i = mine + t; i < np2; i += T) {
        keys[i] = 0xffffffffu;
        idxs[i] = 0;
    }
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= np2; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < np2; i += T) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool asc = (i & k2) == 0;
                    const uint32_t ki = keys[i], kl = keys[l];
                    if ((ki > kl) == asc) {
                        keys[i] = kl; keys[l] = ki;
                        const uint32_t ti = idxs[i]; idxs[i] = idxs[l]; idxs[l] = ti;
                    }
                }
            }
            __syncthreads();
        }
    }
    float *out = a.out + ((size_t)s_cnt[n_parts] + below) * 5;
    for (uint32_t i = t; i < mine; i += T) {
        const uint32_t src = idxs[i];
#pragma unroll
        for (int q = 0; q < 5; ++q) out[(size_t)i * 5 + q] = __uint_as_float(rec_u(src, q + 1));
    }
    return true;
}

// K4, the last launch of the multi-launch chain: 1024-thread workgroups, `n_parts` of them per frame.  Part 0 does the frame's
// rare work and its emission (rare_frame); the other parts only exist for frames with a large list, which the parts emit
// together (emit_large_split).
constexpr uint32_t RARE_MAX_PARTS = 8;
__global__ void __launch_bounds__(1024) k_rare(ChainArgs a, RefineConsts rc, uint32_t lds_entries, uint32_t n_parts)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];  // (emit_wide reads its keys 16 bytes at a time)
    __shared__ uint32_t s_misc3[3];
    __shared__ uint32_t s_cnt[RARE_MAX_PARTS + 2];
    const WaveTimer wt(a, K_RARE);
    const int frame = (int)(blockIdx.x / n_parts);
    const uint32_t part = blockIdx.x - (uint32_t)frame * n_parts;
    if (part == 0) clear_next_counters(a, frame);
    FrameCounters &ctr = a.ctr[frame];
    const FrameHead head = load_frame_head(ctr);
    if (n_parts > 1) {
        const bool generic = a.force_generic || (head.flags & FLAG_BIG_CLUSTER);
        const bool void_frame = (head.flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW)) != 0;
        const bool split = !generic && !void_frame && head.n_refined > TAIL_CAP;  // (the same decision in every part of the frame)
        if (split) {
            if (part == 0 && threadIdx.x == 0) atomicOr(&ctr.flags, FLAG_LARGE_RESULT);
            if (emit_large_split(a, frame, part, n_parts, head.n_refined, head.max_k_bits, lds_u, lds_u + lds_entries, lds_entries, s_cnt)) return;
            __syncthreads();
        }
        if (part != 0) return;
    }
    rare_frame(a, rc, frame, lds_u, lds_entries, s_misc3, head);
}

// ------------------------------------------------------------------------------------------
// K_SPARSE: the whole sparse phase of ONE frame in ONE 1024-thread workgroup -- what K2, K3 and K4 do in three
// batch-wide launches.  The reference's barrier between the stages is per frame (the minimum of the frame's response,
// detector.rs:414-418; its clusters and their refinement, :420-445), so the frame's sixteen waves go from stage to
// stage behind workgroup barriers, and the frames (one per CU, 256 CUs) drift apart as their work differs: one
// frame's latency chains run while another's flood keeps the vector ALUs busy.  The frame's lists live in LDS:
//   verify  the frame's tiles, wave by wave (verify_tiles); seeds into the LDS list
//   flood   one seed per lane (flood_lane / wave_flood_128x64), the cluster refined by the lane that flooded it;
//           cluster count, refined records (first TAIL_CAP), largest k in LDS
//   emit    filter + rank + store from LDS (emit_wide_lds)
// Frames with a component beyond the flood windows, or with more than TAIL_CAP refined records, continue on the
// global-memory path (rare_frame), as under k_rare.  Used for batches that fill the chip with one workgroup per frame;
// smaller batches keep the three launches (plan: launch_kernel / use_sparse_frame).
// ------------------------------------------------------------------------------------------
constexpr uint32_t SF_MISC = 16;                                   // LDS words: counters (below)
constexpr uint32_t SF_REC = SF_MISC;                               // refined records [6][TAIL_CAP]
constexpr uint32_t SF_R = SF_REC + 6 * TAIL_CAP;                   // one region, used in turn by: verify_frame (VFW_WORDS per wave),
                                                                   // the flood seeds, the emission's keys
constexpr uint32_t SF_R_WORDS = 16 * VFW_WORDS;
constexpr uint32_t SF_WORDS = SF_R + SF_R_WORDS;
static_assert(SF_R_WORDS >= TAIL_CAP + 8 && SF_R_WORDS >= SEED_LDS_CAP, "the emission's keys and the flood path's seeds fit the shared region");
static_assert(SF_REC % 4 == 0 && SF_R % 4 == 0, "16-byte aligned parts");
static_assert(SF_WORDS * 4 <= 160 * 1024 - 1024, "fits a CU's LDS");
enum : int { SFM_SEEDS = 0, SFM_FLAGS = 1, SFM_CLUSTERS = 2, SFM_REFINED = 3, SFM_MAXK = 4, SFM_BIG = 5, SFM_EMIT = 8, SFM_RARE = 12 };

template <bool VEC, typename CLK>
__global__ void __launch_bounds__(1024) k_sparse_frame(ChainArgs a, RefineConsts rc, uint32_t lds_entries)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
    const int frame = a.n_frames - 1 - (int)blockIdx.x;  // latest-written blur planes first (cache)
    const uint32_t t = threadIdx.x;
    const int lane = (int)(t & 63u);
    const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    clear_next_counters(a, frame);
    uint32_t *misc = lds_u;
    if (t < SF_MISC) misc[t] = 0u;
    __syncthreads();
    FrameCounters &ctr = a.ctr[frame];
    // debug_ablation & 131072: when the frame's workgroup passed its stages (10 ns ticks of the constant clock, low 32 bits)
    // into the frame's stats[0..4]: start, verify done, flood + refine done, end (tools/sparse_frame_phases.py)
    auto stamp = [&](int which) {
        if ((a.dbg & 131072) && t == 0) ctr.stats[which] = (uint32_t)wall_clock64();
    };
    stamp(0);
    const FrameLds fl{&misc[SFM_SEEDS], &misc[SFM_FLAGS], lds_u + SF_R};
    const size_t cbase = (size_t)frame * a.cap_roots;
    const float *img = a.blur + (size_t)frame * (size_t)a.plane;
    const RecSinkLds sink{&misc[SFM_REFINED], &misc[SFM_MAXK], &misc[SFM_FLAGS], lds_u + SF_REC};
    // ---- verify ----
    if (a.sparse_after_verify) {  // k_verify_seeds has run as a launch of its own: the frame's seeds are in its global list
        if (t == 0) {
            misc[SFM_SEEDS] = __hip_atomic_load(&ctr.n_seeds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            misc[SFM_FLAGS] = __hip_atomic_load(&ctr.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    } else {
        uint32_t *wl = lds_u + SF_R + (uint32_t)wv * VFW_WORDS;
        verify_frame(a, frame, VerifyLds{wl, wl + VFW_SLOTS, wl + 2 * VFW_SLOTS, wl + 3 * VFW_SLOTS});  // (ends with a barrier: the mask is final)
    }
    stamp(1);
    // ---- flood seeds ----
    if (!a.sparse_after_verify) {
        frame_seeds(a, frame, fl);
        __syncthreads();  // the seed list is complete
    }
    stamp(5);
    // ---- flood + refine: one seed per lane ----
    const uint32_t n_seeds_raw = misc[SFM_SEEDS];
    if (!a.force_generic && n_seeds_raw <= a.cap_roots && !(misc[SFM_FLAGS] & FLAG_CAND_OVERFLOW)) {
        const uint32_t n = n_seeds_raw;
        const uint32_t *mask = a.mask + (size_t)frame * (size_t)a.mask_plane;
        const uint32_t W = (uint32_t)a.W;
        CLK clk;  // (debug_ablation & 262144: the PhaseClock instantiation, tools/flood_phases.py)
        clk.start(ctr.stats);
        for (uint32_t base = (uint32_t)wv * 64u; base < n; base += 1024u) {  // wave-uniform trip count
            const uint32_t i = base + (uint32_t)lane;
            uint32_t p = 0u, cnt = 0, sumx = 0, sumy = 0;
            int what = FLOOD_NONE;
            if (CLK::on && lane == 0) atomicAdd(&ctr.stats[19], 1u);  // chunks
            clk.mark(8);
            if (i < n) {
                p = (i < SEED_LDS_CAP && !a.sparse_after_verify) ? fl.seeds[i] : a.seeds[cbase + i];
                what = flood_lane(a, mask, W, p, cnt, sumx, sumy, clk);
            }
            clk.join();
            clk.mark(10);  // the flood's sweeps
            // second tier: see k_flood_refine
            unsigned long long big = __ballot(what == FLOOD_BIG);
            if (big && lane == 0) atomicAdd(&misc[SFM_BIG], (uint32_t)__popcll(big));
            while (big) {  // wave-uniform
                const int src = __ffsll((long long)big) - 1;
                big &= big - 1ull;
                uint32_t c2 = 0, x2 = 0, y2 = 0;
                const int w2 = wave_flood_128x64(a, &misc[SFM_FLAGS], mask, (uint32_t)__builtin_amdgcn_readlane((int)p, src), lane, c2, x2, y2);
                if (lane == src) {
                    what = w2 == FLOOD_CLUSTER ? FLOOD_CLUSTER : FLOOD_NONE;
                    cnt = c2;
                    sumx = x2;
                    sumy = y2;
                }
            }
            clk.mark(14);  // second tier
            if (CLK::on) {
                const unsigned long long act = __ballot(what == FLOOD_CLUSTER);
                if (lane == 0) atomicAdd(&ctr.stats[18], (uint32_t)__popcll(act));  // clusters refined
            }
            if (what == FLOOD_CLUSTER) {
                const uint32_t o = atomicAdd(&misc[SFM_CLUSTERS], 1u);
                float cx, cy;
                refine_values<VEC>(a, rc, frame, img, a.W, a.H, p, cnt, sumx, sumy, sink, cx, cy, clk);
                if (o < a.cap_roots) {  // the cluster table (agx_debug_fetch)
                    a.clu_key[cbase + o] = p;
                    a.clu_cnt[cbase + o] = cnt;
                    a.clu_sx[cbase + o] = __float_as_uint(cx);
                    a.clu_sy[cbase + o] = __float_as_uint(cy);
                } else {
                    atomicOr(&misc[SFM_FLAGS], FLAG_ROOT_OVERFLOW);
                }
            }
            clk.join();
            clk.mark(7);  // record stores, lanes without a record
        }
    }  // (a seed list beyond cap_roots has set FLAG_CAND_OVERFLOW in verify_tiles: the frame is reported, not processed)
    __syncthreads();
    stamp(2);
    const uint32_t flags = misc[SFM_FLAGS], n_ref = misc[SFM_REFINED], n_clu = min(misc[SFM_CLUSTERS], a.cap_roots);
    const bool void_frame = (flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW)) != 0;
    const bool generic = a.force_generic || (flags & FLAG_BIG_CLUSTER);
    if (t == 0) {  // the frame's counters as the host (and agx_debug_fetch) reads them
        ctr.n_seeds = min(n_seeds_raw, a.cap_roots);
        ctr.n_big = misc[SFM_BIG];
        ctr.n_clusters = n_clu;
        ctr.n_refined = n_ref;
        ctr.max_k_bits = misc[SFM_MAXK];
    }
    if (generic || (!void_frame && n_ref > TAIL_CAP)) {
        // the global-memory path: its counters and lists are in global memory (every record is mirrored there)
        if (t == 0) atomicOr(&ctr.flags, flags);
        phase_barrier();
        rare_frame(a, rc, frame, lds_u + SF_MISC, lds_entries, &misc[SFM_RARE], load_frame_head(ctr));  // (everything behind the counters is free now)
        return;
    }
    emit_wide_lds(a, frame, void_frame ? 0u : n_ref, misc[SFM_MAXK], lds_u + SF_REC, lds_u + SF_R, &misc[SFM_EMIT], &misc[SFM_FLAGS], n_clu);
    if (t == 0) ctr.flags = misc[SFM_FLAGS];  // (behind emit_wide_lds's last barrier: OUT_OVERFLOW included)
    stamp(3);
}

// ------------------------------------------------------------------------------------------
// host side: tiling plan and launches
// ------------------------------------------------------------------------------------------
// Tuning overrides from the environment (measurement only; the defaults are the product).  Each name is read from the
// environment ONCE per process -- not on every launch -- and kept; tuning_env_reload() (option "reload_tuning_env": the
// sweep tools that change os.environ between runs of one process) forgets what was read.
namespace {
std::mutex g_env_mutex;
std::map<std::string, std::pair<bool, int>, std::less<>> g_env_cache;  // name -> (set, value); std::less<>: looked up by const char * without a temporary string
}
int tuning_env(const char *name, int dflt)
{
    std::lock_guard<std::mutex> lk(g_env_mutex);
    auto it = g_env_cache.find(name);
    if (it == g_env_cache.end()) {
        const char *s = getenv(name);
        it = g_env_cache.emplace(name, std::make_pair(s && *s, (s && *s) ? atoi(s) : 0)).first;
    }
    return it->second.first ? it->second.second : dflt;
}
void tuning_env_reload()
{
    std::lock_guard<std::mutex> lk(g_env_mutex);
    g_env_cache.clear();
}
static int env_int(const char *name, int dflt) { return tuning_env(name, dflt); }

constexpr long long K1_ASYNC_POLL_MAX_WAVES = 4608;  // measured crossover: 1280x800 x 16 (2 400 waves) and 1920x1080 x 16 (4 352) gain, 1280x800 x 32 (4 800) loses (profiles/r4_k1_async_poll.txt)
bool plan_k1(ChainArgs &a, int override_rows_per_seg)
{
    const int W = a.W, H = a.H;
    if (W < 2 || H < 2) return false;
    // K1 strips: at most 62 value lanes (248 columns) per wave, balanced over the width
    int n_strips = (W + 247) / 248;
    int strip_cols = (((W + n_strips - 1) / n_strips) + 3) & ~3;
    // strips that start on 128-byte lines of the blur plane (32 columns) where that costs no extra strip: the plane is
    // stored non-temporally, and a line shared by two strips then goes to memory as two partial writes (1280 wide:
    // 224 instead of 216 columns, K1 -3 %; 1920 and 3840 keep 240 -- 224 would need a ninth / two more strips)
    const int aligned_cols = (strip_cols + 31) & ~31;
    if (aligned_cols <= 248 && (W + aligned_cols - 1) / aligned_cols == n_strips) strip_cols = aligned_cols;
    const int forced_cols = env_int("AGX_K1_STRIP_COLS", 0);  // tuning override (4 .. 248, multiple of 4)
    if (forced_cols >= 4 && forced_cols <= 248 && (forced_cols & 3) == 0) strip_cols = forced_cols;
    n_strips = (W + strip_cols - 1) / strip_cols;
    a.n_strips = n_strips;
    a.strip_cols = strip_cols;
    a.threads = 256;
    // rows per segment: enough waves to fill the chip several times over, but segments long
    // enough that the 8 warm-up rows stay a small fraction
    int rps = override_rows_per_seg > 0 ? override_rows_per_seg : env_int("AGX_K1_ROWS", 0);
    if (rps <= 0) {
        const long long target_waves = 10240;
        long long segs = (target_waves + (long long)a.n_frames * n_strips - 1) / ((long long)a.n_frames * n_strips);
        if (segs < 1) segs = 1;
        rps = (int)((H + segs - 1) / segs);
        // (with the plane's stores non-temporal, segments of 96 rows measured 2 % better than 128 at 1280x800 in all
        // three formats and the same at 3840x2160; 64 the same as 96, 32 worse)
        if (rps > 96) rps = 96;
    }
    // segments are aligned to the 32-row words of the transposed mask
    rps = ((rps + 31) / 32) * 32;
    if (rps < 32) rps = 32;
    if (rps > 128) rps = 128;
    a.rows_per_seg = rps;
    a.n_segs = (H + rps - 1) / rps;
    a.publish_factor = a.n_strips * a.n_segs > 128 ? 1.5f : 1.0f;  // see the publish step of K1
    a.k1_group = env_int("AGX_K1_GROUP", 0);
    // threshold polls as asynchronous vector loads while the waves are (nearly) alone on their SIMDs; see K1
    const int ap = env_int("AGX_K1_ASYNC_POLL", -1);  // tuning override: 0 / 1
    a.k1_async_poll = ap >= 0 ? (ap != 0) : ((long long)a.n_strips * a.n_segs * a.n_frames <= K1_ASYNC_POLL_MAX_WAVES);
    return true;
}

// Dynamic LDS of k_sparse_frame: its own lists, or the counters + the large-list sort of rare_frame.
size_t sparse_frame_lds_bytes(const ChainArgs &a)
{
    return std::max<size_t>((size_t)SF_WORDS * 4, (size_t)SF_MISC * 4 + k5_lds_bytes(a));
}

size_t k5_lds_bytes(const ChainArgs &a)
{
    uint32_t e = 1024;  // at least the TAIL_CAP + 8 words of emit_wide
    while (e < a.cap_out && e < 16384u) e <<= 1;  // 16384 entries = 128 KB of the CU's 160 KB; longer lists sort in global memory
    return (size_t)e * 8;
}

template <int FMT>
static hipError_t launch_k1(const ChainArgs &a, hipStream_t st)
{
    const long long units = (long long)a.n_strips * a.n_segs * a.n_frames;
    dim3 grid((unsigned)((units + 3) / 4)), block(256);
    const size_t k1_lds = (size_t)env_int("AGX_K1_LDS_KB", 0) * 1024;  // experiment: dynamic LDS caps the workgroups per CU
    // the aligned form addresses a frame and its blur plane with 32-bit buffer offsets
    const bool small = a.plane * 4 < (1ll << 31) && (long long)a.H * a.row_stride < (1ll << 31);
    const bool a4 = (a.W & 3) == 0 && small && !a.byte_rows;
    // L8 frames that miss the aligned form only by their width or alignment: its loads and tap table on unaligned dwords (UF)
    const bool uf = !a4 && small && a.W >= 4 && env_int("AGX_K1_UNALIGNED_FAST", 1) != 0;
    if (uf) {
        if (a.resp_dbg) hipLaunchKernelGGL((k_blur_hessian<FMT, false, true, true>), grid, block, k1_lds, st, a);
        else hipLaunchKernelGGL((k_blur_hessian<FMT, false, false, true>), grid, block, k1_lds, st, a);
        return hipGetLastError();
    }
    if (a.resp_dbg) {  // parity-test instantiation: also stores the response it evaluates
        if (a4) hipLaunchKernelGGL((k_blur_hessian<FMT, true, true>), grid, block, k1_lds, st, a);
        else hipLaunchKernelGGL((k_blur_hessian<FMT, false, true>), grid, block, k1_lds, st, a);
    } else if (a4) hipLaunchKernelGGL((k_blur_hessian<FMT, true>), grid, block, k1_lds, st, a);
    else hipLaunchKernelGGL((k_blur_hessian<FMT, false>), grid, block, k1_lds, st, a);
    return hipGetLastError();
}

static int sparse_grid_x(const ChainArgs &a, int per_frame_default, const char *env = nullptr)
{
    if (env) {  // tuning override: workgroups per frame
        const int v = env_int(env, 0);
        if (v > 0) return v;
    }
    // few frames -> more workgroups per frame
    long long gx = 4096 / (a.n_frames > 0 ? a.n_frames : 1);
    if (gx < per_frame_default) gx = per_frame_default;
    if (gx > 256) gx = 256;
    return (int)gx;
}

int launch_kernel(int which, const ChainArgs &a, const RefineConsts &rc, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    switch (which) {
    case K_BLUR_HESSIAN:
        if (a.fmt == 0) return launch_k1<0>(a, st);
        if (a.fmt == 1) return launch_k1<1>(a, st);
        if (a.fmt == 3) return launch_k1<3>(a, st);
        return launch_k1<2>(a, st);
    case K_THRESHOLD: {
        // one workgroup (= one wave) per tile of VS_OWN columns x VS_ROWS word rows, at most 512 per frame
        const int n_yb = (a.H + 31) >> 5;
        int tiles = ((n_yb + VS_ROWS - 1) / VS_ROWS) * ((a.W + VS_OWN - 1) / VS_OWN);
        if (tiles > 512) tiles = 512;
        const int per_frame = (env_int("AGX_G_VERIFY", tiles) + VS_WAVES - 1) / VS_WAVES;  // workgroups of VS_WAVES waves
        dim3 grid((unsigned)per_frame * (unsigned)a.n_frames), block(64 * VS_WAVES);  // slot-major, see frame_slot
        if ((a.dbg & K2_DBG_BITS) || a.wave_times) hipLaunchKernelGGL(k_verify_seeds<true>, grid, block, 0, st, a);
        else if (VS_WAVES == 1 && env_int("AGX_G_VERIFY", 0) <= 0 && (n_yb + VS_ROWS - 1) / VS_ROWS <= 65535 && (a.W + VS_OWN - 1) / VS_OWN <= 65535)
            hipLaunchKernelGGL((k_verify_seeds<false, VS_WAVES == 1>), dim3((unsigned)a.n_frames, (unsigned)((a.W + VS_OWN - 1) / VS_OWN), (unsigned)((n_yb + VS_ROWS - 1) / VS_ROWS)),
                               block, 0, st, a);
        else hipLaunchKernelGGL(k_verify_seeds<false>, grid, block, 0, st, a);
        return hipGetLastError();
    }
    case K_FLOOD_REFINE: {
        // 16 workgroups per frame at 256 frames = the chip's 4 096 wave slots at 128 registers: a frame's waves loop over its
        // chunks of 64 seeds (17 on the bench's frames) instead of 48 workgroups per frame of which 31 find nothing to do
        // (72 -> 68 us, tools/env_sweep.py)
        dim3 grid((unsigned)sparse_grid_x(a, 16, "AGX_G_FLOOD") * (unsigned)a.n_frames), block(64);  // slot-major, see frame_slot
        if (a.dbg & 16384) {  // phase timeline (tools/flood_phases.py)
            if ((a.W & 3) == 0) hipLaunchKernelGGL((k_flood_refine<true, PhaseClock>), grid, block, 0, st, a, rc);
            else hipLaunchKernelGGL((k_flood_refine<false, PhaseClock>), grid, block, 0, st, a, rc);
        } else if ((a.W & 3) == 0) {
            hipLaunchKernelGGL((k_flood_refine<true, NoClock>), grid, block, 0, st, a, rc);
        } else {
            hipLaunchKernelGGL((k_flood_refine<false, NoClock>), grid, block, 0, st, a, rc);
        }
        return hipGetLastError();
    }
    case K_RARE: {
        size_t lds = k5_lds_bytes(a);  // (the large-LDS attribute is set per device in init_device_kernels)
        // several workgroups per frame when the frames alone do not fill the chip: a frame with a large list (pure noise) is then
        // emitted by all of them together; frames with ordinary lists cost the extra workgroups one look at a counter
        uint32_t parts = a.n_frames >= 128 ? 1u : (a.n_frames >= 64 ? 4u : RARE_MAX_PARTS);
        const int forced = env_int("AGX_RARE_PARTS", 0);
        if (forced >= 1 && forced <= (int)RARE_MAX_PARTS) parts = (uint32_t)forced;
        dim3 grid((unsigned)a.n_frames * parts), block(1024);
        hipLaunchKernelGGL(k_rare, grid, block, lds, st, a, rc, (uint32_t)(lds / 8), parts);
        return hipGetLastError();
    }
    case K_SPARSE: {
        const size_t lds = sparse_frame_lds_bytes(a);
        dim3 grid(a.n_frames), block(1024);
        const uint32_t le = (uint32_t)(k5_lds_bytes(a) / 8);
        if (a.dbg & 262144) {  // phase timeline of the flood + refine stage (tools/flood_phases.py)
            if ((a.W & 3) == 0) hipLaunchKernelGGL((k_sparse_frame<true, PhaseClock>), grid, block, lds, st, a, rc, le);
            else hipLaunchKernelGGL((k_sparse_frame<false, PhaseClock>), grid, block, lds, st, a, rc, le);
        } else if ((a.W & 3) == 0) hipLaunchKernelGGL((k_sparse_frame<true, NoClock>), grid, block, lds, st, a, rc, le);
        else hipLaunchKernelGGL((k_sparse_frame<false, NoClock>), grid, block, lds, st, a, rc, le);
        return hipGetLastError();
    }
    default:
        return hipErrorInvalidValue;
    }
}

// Function attributes belong to the function object of the CURRENT device: called by
// agx_detector_create after hipSetDevice, once per handle (any number of devices per process).
int init_device_kernels()
{
    hipError_t e = hipFuncSetAttribute((const void *)k_rare, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_sparse_frame<true, NoClock>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_sparse_frame<false, NoClock>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_sparse_frame<true, PhaseClock>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_sparse_frame<false, PhaseClock>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    return e;
}

// to_luma8 of one staged frame (detector.rs:507; image 0.25.9: Luma16 -> (v + 128) / 257, Rgb8 ->
// (2126 r + 7152 g + 722 b) / 10000 in integers -- the expressions of the host's luma8()), for
// agx_detect on L16 / RGB8 images: the frame is in device memory anyway, and the host spends 1-2 ms on a
// 1920x1080 RGB frame where this kernel and the copy back take 0.05 ms.
template <int FMT>
__global__ void __launch_bounds__(256) k_luma8(const uint8_t *src, size_t pitch, size_t frame_stride, uint8_t *dst, int W, int H,
                                               int n_frames)
{
    const size_t plane = (size_t)W * (size_t)H, n = plane * (size_t)n_frames;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / plane, r = i - f * plane;
        const size_t y = r / (size_t)W, x = r - y * (size_t)W;
        const uint8_t *row = src + f * frame_stride + y * pitch;
        uint32_t v;
        if (FMT == 1) {
            v = ((uint32_t)reinterpret_cast<const uint16_t *>(row)[x] + 128u) / 257u;
        } else {
            v = (2126u * row[3 * x] + 7152u * row[3 * x + 1] + 722u * row[3 * x + 2]) / 10000u;
        }
        dst[i] = (uint8_t)v;
    }
}

int launch_luma8(const void *src, size_t pitch, size_t frame_stride, int n_frames, int format, uint8_t *dst, int W, int H, void *stream)
{
    const size_t n = (size_t)W * (size_t)H * (size_t)n_frames;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 16384);
    if (format == 1)
        hipLaunchKernelGGL((k_luma8<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)src, pitch, frame_stride, dst, W, H, n_frames);
    else if (format == 2)
        hipLaunchKernelGGL((k_luma8<2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)src, pitch, frame_stride, dst, W, H, n_frames);
    else return (int)hipErrorInvalidValue;
    return hipGetLastError();
}

// agx_saddles_batch_fetch: the batch's results from HBM into the detector's mapped pinned host memory by a KERNEL -- one row per
// frame (count, offset, flags, clusters; row n_frames: the total) and the compact saddle array, 16 bytes per store --
// instead of two device-to-host copies with a wait each: a copy shares the DMA queue with whatever else the process is
// copying (the next batch's 262 MB upload: the results then arrive 2 ms late), a kernel does not.
__global__ void __launch_bounds__(256) k_publish(const FrameCounters *ctr, int n_frames, const uint32_t *total_out, const float *out,
                                                 uint32_t h_out_records, uint32_t *h_table, float *h_out)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    const uint32_t total = *total_out;
    for (size_t f = tid; f <= (size_t)n_frames; f += nthr) {
        uint4 row;
        if (f < (size_t)n_frames) {
            const FrameCounters &c = ctr[f];
            row = make_uint4(c.n_out, c.out_offset, c.flags, c.n_clusters + c.n_clusters2);
        } else {
            row = make_uint4(total, 0u, 0u, 0u);
        }
        reinterpret_cast<uint4 *>(h_table)[f] = row;
    }
    const size_t n4 = ((size_t)(total < h_out_records ? total : h_out_records) * 5 + 3) / 4;  // (both arrays are padded to 16 bytes)
    for (size_t i = tid; i < n4; i += nthr) reinterpret_cast<float4 *>(h_out)[i] = reinterpret_cast<const float4 *>(out)[i];
}

int launch_publish(const ChainArgs &a, uint32_t h_out_records, uint32_t *h_table_dev, float *h_out_dev, void *stream)
{
    hipLaunchKernelGGL(k_publish, dim3(256), dim3(256), 0, (hipStream_t)stream, a.ctr, a.n_frames, a.total_out, a.out, h_out_records,
                       h_table_dev, h_out_dev);
    return hipGetLastError();
}

// Zero n counter records (the library's own kernel: a memset node inside a captured HIP graph faulted on the
// graph's second replay on ROCm 7.2, a kernel node does not).
__global__ void __launch_bounds__(256) k_clear_counters(uint32_t *p, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

int launch_clear_counters(FrameCounters *ctr, size_t n_records, void *stream)
{
    const size_t n_words = n_records * (sizeof(FrameCounters) / 4);
    const unsigned grid = (unsigned)std::min<size_t>((n_words + 255) / 256, 1024);
    hipLaunchKernelGGL(k_clear_counters, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint32_t *>(ctr), n_words);
    return hipGetLastError();
}

int launch_debug_resp(const ChainArgs &a, int frame, float *dst, void *stream)
{
    hipLaunchKernelGGL(k_debug_resp, dim3(1024), dim3(256), 0, (hipStream_t)stream,
                       a.blur + (size_t)frame * (size_t)a.plane, dst, a.W, a.H);
    return hipGetLastError();
}

}  // namespace agx
