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
//                       per-frame min (detector.rs:414-417).  Row-marching workgroups: a
//                       workgroup owns a column strip and walks down a segment of rows,
//                       keeping the 7-row vertical window and the 3-row Hessian window in
//                       registers; LDS carries only the +-4 column neighbour exchange.
//   K2 k_threshold      resp < 0.05*min (detector.rs:418,177): candidate compaction
//   K3a k_union         4-connected components (image_util.rs:208-236) as lock-free union-find
//   K3b k_centroid      centroid sums (detector.rs:421-429), cluster list
//   K4 k_refine         rochade_refine (detector.rs:194-361), one cluster per lane
//   K5 k_filter_sort    k/phi filter (detector.rs:436-445), emission in reference order
#include <hip/hip_runtime.h>

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

// Raw pixel words of 4 consecutive pixels: L8 1 dword, L16 2 dwords, RGB8 3 dwords.
template <int FMT>
struct RawPx {
    static constexpr int BPP = FMT == 0 ? 1 : (FMT == 1 ? 2 : 3);
    uint32_t d[BPP];
};

template <int FMT>
__device__ __forceinline__ RawPx<FMT> load_raw(const uint8_t *__restrict__ rowp, int c0, int W)
{
    constexpr int BPP = RawPx<FMT>::BPP;
    RawPx<FMT> r;
    if (c0 + 3 < W) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(rowp + (size_t)c0 * BPP);
#pragma unroll
        for (int i = 0; i < BPP; ++i) r.d[i] = p[i];
    } else {
        // last partial lane / lanes right of the image: clamp-to-edge, byte gather
#pragma unroll
        for (int i = 0; i < BPP; ++i) r.d[i] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = c0 + j;
            c = c < W - 1 ? c : W - 1;
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

// ------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------
// LDS: two ping-pong row buffers for the converted input row and two for the freshly blurred
// row; slot t+1 belongs to thread t, slots 0 and T+1 are the clamp pads.
template <int FMT, bool STORE_RESP>
__global__ void k_blur_hessian(ChainArgs a)
{
    extern __shared__ float4 lds4[];
    const int T = blockDim.x;
    const int t = threadIdx.x;
    float4 *sIn = lds4;                 // [2][T+2]
    float4 *sBl = lds4 + 2 * (T + 2);   // [2][T+2]

    const int strip = blockIdx.x % a.n_strips;
    const int seg = blockIdx.x / a.n_strips;
    const int frame = blockIdx.y;
    const int W = a.W, H = a.H;
    const int xs = strip * a.strip_cols;
    const int xe = min(W, xs + a.strip_cols);
    const int hl = xs > 0 ? 1 : 0;
    const int c0 = xs + 4 * (t - hl);
    const bool lane_valid = (c0 >= xs) && (c0 < xe);
    const int ys = seg * a.rows_per_seg;
    const int ye = min(H, ys + a.rows_per_seg);

    const uint8_t *fbase = a.frames + (size_t)frame * (size_t)a.frame_stride;
    float *blur_f = a.blur + (size_t)frame * (size_t)a.plane;
    float *resp_f = a.resp + (size_t)frame * (size_t)a.plane;
    const bool vec_ok = ((W & 3) == 0) && (c0 + 3 < W);

    const float w0 = a.w[0], w1 = a.w[1], w2 = a.w[2], w3 = a.w[3], w4 = a.w[4], w5 = a.w[5],
                w6 = a.w[6];

    // vertical window of horizontally blurred rows (oldest first)
    float h0[4], h1[4], h2[4], h3[4], h4[4], h5[4], h6[4];
    // Hessian window: rows b-3 (up), b-2 (mid) complete with edge columns; bprev = row b-1
    float up[6], mid[6], bprev[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h0[j] = h1[j] = h2[j] = h3[j] = h4[j] = h5[j] = h6[j] = 0.0f;
        bprev[j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) up[j] = mid[j] = 0.0f;

    float run_min = 0.0f;  // the frame always contains its zero border ring

    const int r0 = ys - 4, r1 = ye + 4;
    auto rowptr = [&](int r) {
        int rr = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
        return fbase + (size_t)rr * (size_t)a.row_stride;
    };
    RawPx<FMT> raw_a = load_raw<FMT>(rowptr(r0), c0, W);
    RawPx<FMT> raw_b = load_raw<FMT>(rowptr(r0 + 1), c0, W);
    RawPx<FMT> raw_c = load_raw<FMT>(rowptr(r0 + 2), c0, W);

#pragma unroll 1
    for (int r = r0; r <= r1; ++r) {
        const int buf = (r - r0) & 1;
        float4 *in_row = sIn + buf * (T + 2);
        float4 *bl_row = sBl + buf * (T + 2);

        float m[4];
        convert_px<FMT>(raw_a, m);
        raw_a = raw_b;
        raw_b = raw_c;
        if (r + 3 <= r1) raw_c = load_raw<FMT>(rowptr(r + 3), c0, W);

        in_row[t + 1] = make_float4(m[0], m[1], m[2], m[3]);
        if (t == 0) in_row[0] = make_float4(m[0], m[0], m[0], m[0]);
        if (t == T - 1) in_row[T + 1] = make_float4(m[3], m[3], m[3], m[3]);
        bl_row[t + 1] = make_float4(bprev[0], bprev[1], bprev[2], bprev[3]);
        __syncthreads();

        const float4 L = in_row[t];
        const float4 R = in_row[t + 2];
        const float bl_edge = bl_row[t].w;      // blur(row b-1, column c0-1)
        const float br_edge = bl_row[t + 2].x;  // blur(row b-1, column c0+4)

        // horizontal pass, image_util.rs:137-185: taps in index order, mul then add
        const float x[10] = {L.y, L.z, L.w, m[0], m[1], m[2], m[3], R.x, R.y, R.z};
        float hn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = x[j] * w0;
            v = v + x[j + 1] * w1;
            v = v + x[j + 2] * w2;
            v = v + x[j + 3] * w3;
            v = v + x[j + 4] * w4;
            v = v + x[j + 5] * w5;
            v = v + x[j + 6] * w6;
            hn[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h0[j] = h1[j]; h1[j] = h2[j]; h2[j] = h3[j]; h3[j] = h4[j];
            h4[j] = h5[j]; h5[j] = h6[j]; h6[j] = hn[j];
        }
        // vertical pass, image_util.rs:187-203: blur row b = r-3
        const int b = r - 3;
        float bcur[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = h0[j] * w0;
            v = v + h1[j] * w1;
            v = v + h2[j] * w2;
            v = v + h3[j] * w3;
            v = v + h4[j] * w4;
            v = v + h5[j] * w5;
            v = v + h6[j] * w6;
            bcur[j] = v;
        }
        if (lane_valid && b >= ys && b < ye) {
            float *dst = blur_f + (size_t)b * W + c0;
            if (vec_ok) {
                *reinterpret_cast<float4 *>(dst) = make_float4(bcur[0], bcur[1], bcur[2], bcur[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + j < W) dst[j] = bcur[j];
            }
        }

        // Hessian determinant of row y = b-2 (rows b-3, b-2, b-1), image_util.rs:88-106
        const int y = b - 2;
        const float dn[6] = {bl_edge, bprev[0], bprev[1], bprev[2], bprev[3], br_edge};
        if (lane_valid && y >= ys && y < ye) {
            float o[4];
            const bool row_border = (y == 0) || (y == H - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v11 = up[j], v12 = up[j + 1], v13 = up[j + 2];
                const float v21 = mid[j], v22 = mid[j + 1], v23 = mid[j + 2];
                const float v31 = dn[j], v32 = dn[j + 1], v33 = dn[j + 2];
                const float t22 = v22 * 2.0f;
                const float lxx = (v21 - t22) + v23;
                const float lyy = (v12 - t22) + v32;
                const float lxy = (((v13 - v11) + v31) - v33) * 0.25f;
                float d = lxx * lyy - lxy * lxy;
                const int c = c0 + j;
                if (row_border || c == 0 || c >= W - 1) d = 0.0f;
                o[j] = d;
                if (c < W) run_min = fminf(run_min, d);
            }
            if (STORE_RESP) {
                float *dst = resp_f + (size_t)y * W + c0;
                if (vec_ok) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < W) dst[j] = o[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            up[j] = mid[j];
            mid[j] = dn[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bprev[j] = bcur[j];
    }

    // per-frame min: wave shuffle reduction -> LDS -> one atomic per workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) run_min = fminf(run_min, __shfl_xor(run_min, off, 64));
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds4);
    if ((t & 63) == 0) red[t >> 6] = run_min;
    __syncthreads();
    if (t == 0) {
        float mn = red[0];
        for (int i = 1; i < (T >> 6); ++i) mn = fminf(mn, red[i]);
        atomicMax(&a.ctr[frame].min_key_inv, ~f32_order_key(mn));
    }
}

// ------------------------------------------------------------------------------------------
// K2: threshold + candidate compaction.  A candidate gets a slot s in the frame's compact
// arrays: cand[s] = pixel | left<<30 | up<<31 (is the left / upper 4-neighbour a candidate
// too), parent[s] = s, zeroed sums; slot_plane[pixel] = s for the neighbour lookups of K3a.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void emit_candidate(const ChainArgs &a, int frame, uint32_t p, bool left,
                                               bool upn)
{
    uint32_t slot = atomicAdd(&a.ctr[frame].n_cand, 1u);
    if (slot < a.cap_cand) {
        size_t o = (size_t)frame * a.cap_cand + slot;
        a.cand[o] = p | (left ? 0x40000000u : 0u) | (upn ? 0x80000000u : 0u);
        a.parent[o] = slot;
        a.sumx[o] = 0u;
        a.sumy[o] = 0u;
        a.cnt[o] = 0u;
        a.minidx[o] = 0xffffffffu;
        a.slot_plane[(size_t)frame * (size_t)a.plane + p] = slot;
    } else {
        atomicOr(&a.ctr[frame].flags, FLAG_CAND_OVERFLOW);
    }
}

template <bool VEC>
__global__ void k_threshold(ChainArgs a)
{
    const int frame = blockIdx.y;
    const float *resp = a.resp + (size_t)frame * (size_t)a.plane;
    const float mn = f32_from_order_key(~a.ctr[frame].min_key_inv);
    const float thr = mn * 0.05f;  // detector.rs:418
    const int W = a.W;
    if (VEC) {
        const long long n4 = a.plane >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
             i += (long long)gridDim.x * blockDim.x) {
            const float4 v = reinterpret_cast<const float4 *>(resp)[i];
            const bool c0 = v.x < thr, c1 = v.y < thr, c2 = v.z < thr, c3 = v.w < thr;
            if (c0 | c1 | c2 | c3) {
                const uint32_t p0 = (uint32_t)(i << 2);
                const uint32_t x0 = p0 % (uint32_t)W;
                // candidates are interior pixels (the border ring is exactly 0 >= thr), so
                // p-1 and p-W exist
                if (c0) emit_candidate(a, frame, p0, (x0 > 0) && (resp[p0 - 1] < thr), resp[p0 - W] < thr);
                if (c1) emit_candidate(a, frame, p0 + 1, c0, resp[p0 + 1 - W] < thr);
                if (c2) emit_candidate(a, frame, p0 + 2, c1, resp[p0 + 2 - W] < thr);
                if (c3) emit_candidate(a, frame, p0 + 3, c2, resp[p0 + 3 - W] < thr);
            }
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.plane;
             i += (long long)gridDim.x * blockDim.x) {
            const float v = resp[i];
            if (v < thr) {
                const uint32_t p = (uint32_t)i;
                emit_candidate(a, frame, p, resp[p - 1] < thr, resp[p - W] < thr);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K3a: union-find.  Links always point from the larger slot to the smaller one; all accesses
// to parent[] are agent-scope atomics (workgroups of one frame run on any XCD).
// ------------------------------------------------------------------------------------------
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

__global__ void k_union(ChainArgs a)
{
    const int frame = blockIdx.y;
    if (a.ctr[frame].n_cand > a.cap_cand) return;  // overflow: slot_plane is incomplete
    const uint32_t n = a.ctr[frame].n_cand;
    const size_t base = (size_t)frame * a.cap_cand;
    uint32_t *parent = a.parent + base;
    const uint32_t *slot_plane = a.slot_plane + (size_t)frame * (size_t)a.plane;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const uint32_t e = a.cand[base + s];
        const uint32_t p = e & 0x3fffffffu;
        if (e & 0x40000000u) uf_unite(parent, s, slot_plane[p - 1]);
        if (e & 0x80000000u) uf_unite(parent, s, slot_plane[p - a.W]);
    }
}

// K3b: every candidate adds its coordinates to its root's sums (integer: exact and order
// independent; the reference's f32 running sums, detector.rs:424-427, are exact too while
// they stay below 2^24 -- FLAG_CENTROID_INEXACT marks the frames where they would not be);
// roots append themselves to the cluster list.
__global__ void k_centroid(ChainArgs a)
{
    const int frame = blockIdx.y;
    if (a.ctr[frame].n_cand > a.cap_cand) return;  // overflow: frame is reported, not processed
    const uint32_t n = a.ctr[frame].n_cand;
    const size_t base = (size_t)frame * a.cap_cand;
    const uint32_t *parent = a.parent + base;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const uint32_t p = a.cand[base + s] & 0x3fffffffu;
        uint32_t r = s;
        for (;;) {
            uint32_t q = parent[r];
            if (q == r) break;
            r = q;
        }
        const uint32_t x = p % (uint32_t)a.W, y = p / (uint32_t)a.W;
        atomicAdd(&a.sumx[base + r], x);
        atomicAdd(&a.sumy[base + r], y);
        atomicAdd(&a.cnt[base + r], 1u);
        atomicMin(&a.minidx[base + r], p);
        if (r == s) {
            uint32_t i = atomicAdd(&a.ctr[frame].n_roots, 1u);
            if (i < a.cap_roots) a.roots[(size_t)frame * a.cap_roots + i] = s;
            else atomicOr(&a.ctr[frame].flags, FLAG_ROOT_OVERFLOW);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K4: rochade_refine, detector.rs:265-359, one cluster per lane.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_refine(ChainArgs a, RefineConsts rc)
{
    const int frame = blockIdx.y;
    const uint32_t n = min(a.ctr[frame].n_roots, a.cap_roots);
    const size_t cbase = (size_t)frame * a.cap_cand;
    const float *img = a.blur + (size_t)frame * (size_t)a.plane;
    const int W = a.W, H = a.H;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t s = a.roots[(size_t)frame * a.cap_roots + i];
        const uint32_t sx = a.sumx[cbase + s], sy = a.sumy[cbase + s], cn = a.cnt[cbase + s];
        if (sx >= (1u << 24) || sy >= (1u << 24)) atomicOr(&a.ctr[frame].flags, FLAG_CENTROID_INEXACT);
        const float fn = (float)cn;
        const float initial_x = (float)sx / fn;  // detector.rs:427
        const float initial_y = (float)sy / fn;
        a.sumx[cbase + s] = __float_as_uint(initial_x);  // kept for agx_debug_fetch
        a.sumy[cbase + s] = __float_as_uint(initial_y);
        const float rxf = roundf(initial_x), ryf = roundf(initial_y);
        const int round_x = (int)rxf, round_y = (int)ryf;
        if (round_y - 4 < 0 || round_y + 4 >= H || round_x - 4 < 0 || round_x + 4 >= W) continue;
        const float *win = img + (size_t)(round_y - 4) * W + (round_x - 4);
        float v[81];
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = 0; c < 9; ++c) v[r * 9 + c] = win[(size_t)r * W + c];
        // cone-filtered 5x5 patch (:283-297) folded straight into the 6 parameter sums
        // (:321-328): params[j] = sum_i pmat[i][j]*patch[i], i ascending
        float prm[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                float conv = 0.0f;
#pragma unroll
                for (int pr = 0; pr < 5; ++pr)
#pragma unroll
                    for (int pc = 0; pc < 5; ++pc)
                        conv = conv + v[(r + pr) * 9 + (c + pc)] * rc.cone[pr * 5 + pc];
#pragma unroll
                for (int j = 0; j < 6; ++j) prm[j] = prm[j] + rc.pmat[(r * 5 + c) * 6 + j] * conv;
            }
        const float a1 = prm[0], a2 = prm[1], a3 = prm[2], a4 = prm[3], a5 = prm[4];
        const float fxx = 2.0f * a1, fyy = 2.0f * a3, fxy = a2;
        const float d = fxx * fyy - fxy * fxy;
        if (!(d < 0.0f)) continue;
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
        if (!(fabsf(x0) <= 1.0f && fabsf(y0) <= 1.0f)) continue;
        const float c5 = (a1 + a3) / 2.0f;
        const float c4 = (a1 - a3) / 2.0f;
        const float c3 = a2 / 2.0f;
        const float k = sqrtf(c4 * c4 + c3 * c3);
        if (!(fabsf(c5) < k)) continue;
        const float PI_F = 3.14159274101257324219f;
        const float phi = acosf(-c5 / k) / 2.0f / PI_F * 180.0f;
        const float theta = atan2f(c3, c4) / 2.0f / PI_F * 180.0f;
        uint32_t o = atomicAdd(&a.ctr[frame].n_refined, 1u);  // o < n_roots <= cap_roots
        RefinedRec rec;
        rec.key = a.minidx[cbase + s];
        rec.x = rxf + x0;
        rec.y = ryf + y0;
        rec.k = k;
        rec.theta = theta;
        rec.phi = phi;
        a.refined[(size_t)frame * a.cap_roots + o] = rec;
        atomicMax(&a.ctr[frame].max_k_bits, __float_as_uint(k));
    }
}

// ------------------------------------------------------------------------------------------
// K5: filter (detector.rs:436-445) and emission in the reference's order = ascending first
// (smallest) pixel index of the cluster.  One workgroup per frame; bitonic sort in LDS.
// ------------------------------------------------------------------------------------------
__global__ void k_filter_sort(ChainArgs a, uint32_t lds_entries)
{
    extern __shared__ uint32_t lds_u[];
    uint32_t *keys = lds_u;
    uint32_t *idxs = lds_u + lds_entries;
    __shared__ uint32_t s_count, s_offset, s_fits;
    const int frame = blockIdx.x;
    const uint32_t t = threadIdx.x, T = blockDim.x;
    FrameCounters &ctr = a.ctr[frame];
    const uint32_t n = ctr.n_refined;
    const RefinedRec *rec = a.refined + (size_t)frame * a.cap_roots;
    if (t == 0) s_count = 0;
    __syncthreads();
    uint32_t nf = 0;
    bool ok = true;
    if (n != 0) {  // detector.rs:432-434: nothing refined -> empty result
        const float s_max_k = __uint_as_float(ctr.max_k_bits) / 10.0f;
        for (uint32_t i = t; i < n; i += T) {
            const float k = rec[i].k, phi = rec[i].phi;
            if (k >= s_max_k && phi >= a.min_angle && phi <= a.max_angle) {
                uint32_t o = atomicAdd(&s_count, 1u);
                if (o < lds_entries) {
                    keys[o] = rec[i].key;
                    idxs[o] = i;
                }
            }
        }
        __syncthreads();
        nf = s_count;
        ok = nf <= a.cap_out && nf <= lds_entries;
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
        s_offset = off;
        s_fits = fits;
        ctr.n_out = nf;
        ctr.out_offset = off;
        if (a.frame_table) {
            uint32_t *row = a.frame_table + (size_t)frame * 4;
            const uint32_t flags = ctr.flags;
            const bool bad = (flags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) != 0;
            row[0] = bad ? 0u : nf;
            row[1] = off;
            row[2] = flags;
            row[3] = ctr.n_roots;
        }
    }
    __syncthreads();
    if (!ok || !nf || !s_fits) return;
    float *out = a.out + (size_t)s_offset * 5;
    for (uint32_t i = t; i < nf; i += T) {
        const RefinedRec r = rec[idxs[i]];
        out[i * 5 + 0] = r.x;
        out[i * 5 + 1] = r.y;
        out[i * 5 + 2] = r.k;
        out[i * 5 + 3] = r.theta;
        out[i * 5 + 4] = r.phi;
    }
}

// ------------------------------------------------------------------------------------------
// host side: tiling plan and launches
// ------------------------------------------------------------------------------------------
static int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

bool plan_k1(ChainArgs &a, int override_rows_per_seg)
{
    const int W = a.W, H = a.H;
    if (W < 2 || H < 2) return false;
    const int max_threads = 512;              // <= 8 waves per workgroup
    const int max_cols = 4 * (max_threads - 2);
    int n_strips = (W + max_cols - 1) / max_cols;
    int strip_cols = (((W + n_strips - 1) / n_strips) + 3) & ~3;
    n_strips = (W + strip_cols - 1) / strip_cols;
    int lanes = strip_cols / 4 + (n_strips > 1 ? 2 : 0);
    int threads = ((lanes + 63) / 64) * 64;
    a.n_strips = n_strips;
    a.strip_cols = strip_cols;
    a.threads = threads;
    // rows per segment: enough workgroups to fill 256 CUs a few times over, but segments
    // long enough that the 9 warm-up rows stay a small fraction
    int rps = override_rows_per_seg > 0 ? override_rows_per_seg : env_int("AGX_K1_ROWS", 0);
    if (rps <= 0) {
        const long long target_wgs = 2048;
        long long segs = (target_wgs + (long long)a.n_frames * n_strips - 1) /
                         ((long long)a.n_frames * n_strips);
        if (segs < 1) segs = 1;
        rps = (int)((H + segs - 1) / segs);
        if (rps < 16) rps = 16;
        if (rps > 256) rps = 256;
    }
    if (rps > H) rps = H;
    a.rows_per_seg = rps;
    a.n_segs = (H + rps - 1) / rps;
    return true;
}

size_t k5_lds_bytes(const ChainArgs &a)
{
    uint32_t e = 1;
    while (e < a.cap_out) e <<= 1;
    return (size_t)e * 8;
}

template <int FMT>
static hipError_t launch_k1(const ChainArgs &a, hipStream_t st)
{
    dim3 grid(a.n_strips * a.n_segs, a.n_frames), block(a.threads);
    size_t lds = (size_t)4 * (a.threads + 2) * sizeof(float4);
    hipLaunchKernelGGL((k_blur_hessian<FMT, true>), grid, block, lds, st, a);
    return hipGetLastError();
}

static int sparse_grid_x(const ChainArgs &a, int per_frame_default)
{
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
        return launch_k1<2>(a, st);
    case K_THRESHOLD: {
        const bool vec = (a.W & 3) == 0;
        long long items = vec ? (a.plane >> 2) : a.plane;
        long long gx = (items + 255) / 256;
        long long cap = 8192 / (a.n_frames > 0 ? a.n_frames : 1);
        if (cap < 16) cap = 16;
        if (gx > cap) gx = cap;
        dim3 grid((unsigned)gx, a.n_frames), block(256);
        if (vec) hipLaunchKernelGGL(k_threshold<true>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(k_threshold<false>, grid, block, 0, st, a);
        return hipGetLastError();
    }
    case K_UNION: {
        dim3 grid(sparse_grid_x(a, 8), a.n_frames), block(256);
        hipLaunchKernelGGL(k_union, grid, block, 0, st, a);
        return hipGetLastError();
    }
    case K_CENTROID: {
        dim3 grid(sparse_grid_x(a, 8), a.n_frames), block(256);
        hipLaunchKernelGGL(k_centroid, grid, block, 0, st, a);
        return hipGetLastError();
    }
    case K_REFINE: {
        dim3 grid(sparse_grid_x(a, 16), a.n_frames), block(64);
        hipLaunchKernelGGL(k_refine, grid, block, 0, st, a, rc);
        return hipGetLastError();
    }
    case K_FILTER_SORT: {
        size_t lds = k5_lds_bytes(a);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void *)k_filter_sort,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        dim3 grid(a.n_frames), block(256);
        hipLaunchKernelGGL(k_filter_sort, grid, block, lds, st, a, (uint32_t)(lds / 8));
        return hipGetLastError();
    }
    default:
        return hipErrorInvalidValue;
    }
}

}  // namespace agx
