// tail_kernels.hip -- TagDetector::detect's board search and tag decode on the device (gfx950).
//
// What the reference does after refined_saddle_points (src/detector.rs:510-539): up to max_num_of_boards rounds of
// try_find_best_board (:588-639; init_quads :543-586, board::Board src/board.rs, is_valid_quad src/saddle.rs:17-67) and
// try_decode_quad (:448-476) over a few hundred saddles per frame.  The chain leaves those saddles in device memory; this
// kernel runs the same search there, one wave per frame, so that a batch's tags -- a few KB -- are all that crosses PCIe and
// no host thread spends a millisecond per frame on it (host_tail.cpp is the same algorithm on the host and stays the
// reference-exact arbiter, below).
//
// Exactness.  Everything the search decides on is binary32 arithmetic in the reference's operand order (this file is
// compiled with -ffp-contract=off like the chain), integer work, or one of three libm calls: atan2f (angle_degree,
// src/math_util.rs:31-33) and cosf / sinf (src/saddle.rs:28-29).  atan2f is evaluated here by the routine glibc itself
// uses, operation for operation (libm_f32.h; the detector checks at run time that the host's atan2f is that routine and
// refuses the device tail otherwise), so the angle comparisons are the reference's own expressions.  cosf / sinf are not
// reproducible that way (a table-driven binary64 routine built with and without FMA): the one test that uses them --
// "filter white block", 60 <= |angle| <= 120 -- is decided from a binary64 evaluation when the angle is farther from both
// thresholds than a 1-ulp change of cosf / sinf and the reference's own roundings can move it (kBandAbs); a closer one that
// matters (every other test of the quad passes) raises TAIL_UNCERTAIN for the frame, which then takes the host tail (libm
// itself).  So do frames beyond the fixed list sizes (TAIL_CAPACITY).  A frame the kernel reports TAIL_OK for has the host
// tail's tags, bit for bit (tests/test_gpu_device_tail.py).
//
// Mapping.  One 64-lane workgroup (one wave) per frame; the frame's saddles, a uniform-grid index for the 3-NN queries
// of find_closest_potential_saddle_idxs (src/board.rs:177-233), the candidate quads of the current seed and the boards
// under construction live in LDS (~70 KB: two frames per CU).  The phases of a seed -- 50-NN (bitonic sort of the distance
// keys), the same / different orientation lists, the candidate quads (lanes over the (d0, d1) combinations, written in the
// reference's order by ballot + prefix count) -- are wave-parallel; the boards of a seed's candidate quads are built SIXTEEN
// AT A TIME, one lane each (board.rs's recursion as an explicit stack in the lane's LDS slot), which is where the time goes.
#include <hip/hip_runtime.h>

#include "libm_f32.h"
#include "tail_kernels.h"

namespace agx {
namespace {

constexpr int TN = TAIL_MAX_SADDLES;
constexpr int TGC = 1024;    // cells of the k-NN grid
constexpr int TCAND = 1024;  // candidate quads of one seed
constexpr int TB = 16;       // boards built side by side
constexpr int BCELLS = 128;  // cells (found or not) of one board
constexpr int BGR = 12, BGN = 2 * BGR + 1;  // board cells live within +-BGR of the seed's cell
constexpr int TTAGS = 128;   // distinct tag ids of one frame

// a board's slot in LDS (bytes)
constexpr int SL_QUAD = 0;       // u16[BCELLS][4]
constexpr int SL_XY = 1024;      // i8[BCELLS][2]
constexpr int SL_FOUND = 1280;   // u8[BCELLS]
constexpr int SL_GRID = 1408;    // u8[BGN * BGN] cell coordinates -> cell number (0xff none)
constexpr int SL_ACTIVE = 2048;  // u32[TN / 32]: board.rs active_idxs
constexpr int SL_STACK = 2112;   // u8[BCELLS][2]: cell, next direction
constexpr int SL_TMP = 2368;     // u16[12]: the candidate lists of try_expand_one
constexpr int SL_BYTES = 2396;   // 599 dwords (odd: the slots of neighbouring lanes start in different banks)
static_assert(BGN * BGN <= SL_ACTIVE - SL_GRID, "board grid");
static_assert(TN / 8 <= SL_STACK - SL_ACTIVE, "active mask");

// the wave's LDS (bytes)
constexpr int OFF_SX = 0, OFF_SY = OFF_SX + TN * 4, OFF_ST = OFF_SY + TN * 4;
constexpr int OFF_GX = OFF_ST + TN * 4, OFF_GY = OFF_GX + TN * 4, OFF_GI = OFF_GY + TN * 4;
constexpr int OFF_GSTART = OFF_GI + TN * 2;                // u16[TGC + 1]
constexpr int OFF_SEEDS = OFF_GSTART + (TGC + 4) * 2;      // u16[TN]
constexpr int OFF_CAND = OFF_SEEDS + TN * 2;               // u64[TCAND]; also: u32[2 * TGC] while the grid is built, u64[TN] distance keys,
                                                           // decode results
constexpr int OFF_PAIRS = OFF_CAND + TCAND * 8;            // u16[1176 + pad]
constexpr int OFF_SMALL = OFF_PAIRS + 1184 * 2;            // u16[4][64]: same, diff, s1 that pass part 1, spare
constexpr int OFF_QUADS = OFF_SMALL + 512;                 // u64[BCELLS]
constexpr int OFF_TAGIDS = OFF_QUADS + BCELLS * 8;         // u32[TTAGS]
constexpr int OFF_USED = OFF_TAGIDS + TTAGS * 4;           // u32[TN / 32]
constexpr int OFF_HIST = OFF_USED + TN / 8;                // u32[364]
constexpr int OFF_BOARDS = OFF_HIST + 364 * 4;             // (TB + 1) slots: the last one keeps the best board's cells
constexpr int LDS_BYTES = OFF_BOARDS + (TB + 1) * SL_BYTES;
static_assert(OFF_CAND % 8 == 0 && OFF_QUADS % 8 == 0 && OFF_BOARDS % 4 == 0, "alignment");

constexpr float kPiF = 3.14159274101257324219f;
// The white-block angle: cosf / sinf within 1 ulp move the direction by < 1.2e-7 rad (7e-6 degrees), the reference's six
// binary32 roundings of the two atan2f operands by < 1.1e-5, atan2f itself (<= 1 ulp at <= 2.1 rad) by 1.4e-5, the
// conversion to degrees (two roundings, a binary32 pi that is divided by here as well) by 1.6e-5: < 5e-5 degrees.  Twice that.
constexpr double kBandAbs = 1e-4;
constexpr double kDegD = 180.0 / (double)kPiF;
typedef unsigned long long u64;

struct Ctx {
    const float *sx, *sy, *st;
    const float *gx, *gy;
    const uint16_t *gi, *gstart;
    float ox, oy, inv_cell;
    int nx, ny, n;
};

__device__ __forceinline__ float theta_dist(float t0, float t1)  // math_util.rs:15-23
{
    float d = t0 - t1 + 90.0f;
    if (d < 0.0f) d += 180.0f;
    else if (d > 180.0f) d -= 180.0f;
    return d > 90.0f ? d - 90.0f : 90.0f - d;
}
__device__ __forceinline__ float round_half_away(float x)  // f32::round
{
    if (!(fabsf(x) < 8388608.0f)) return x;
    float t = (float)(int32_t)x;
    const float d = x - t;
    if (d >= 0.5f) t += 1.0f;
    else if (d <= -0.5f) t -= 1.0f;
    return t;
}
__device__ __forceinline__ uint32_t f32_as_u32(float v)  // Rust `as u32`
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}
__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }

__device__ __forceinline__ int cell_x(const Ctx &c, float x)
{
    const int i = (int)floorf((x - c.ox) * c.inv_cell);
    return i < 0 ? 0 : (i >= c.nx ? c.nx - 1 : i);
}
__device__ __forceinline__ int cell_y(const Ctx &c, float y)
{
    const int i = (int)floorf((y - c.oy) * c.inv_cell);
    return i < 0 ? 0 : (i >= c.ny ? c.ny - 1 : i);
}

__device__ __forceinline__ void top3_insert(u64 x, u64 &k0, u64 &k1, u64 &k2)
{
    u64 lo = k0 < x ? k0 : x;
    x = k0 < x ? x : k0;
    k0 = lo;
    lo = k1 < x ? k1 : x;
    x = k1 < x ? x : k1;
    k1 = lo;
    k2 = k2 < x ? k2 : x;
}
__device__ __forceinline__ u64 dist_key(float qx, float qy, float px, float py, uint32_t idx)
{
    const float dx = qx - px, dy = qy - py;
    const float d2 = dx * dx + dy * dy;  // kdtree's squared_euclidean, folded from 0.0 (0.0 + dx * dx is dx * dx: never -0)
    return (u64)__float_as_uint(d2) << 32 | idx;
}

// The three nearest saddles of (qx, qy) among those within r2, ascending (distance, index) -- which is what
// tree.nearest(.., 3, ..) filtered by `dist_sq <= radius_sq` leaves (board.rs:193-212): a saddle within the radius that
// is among the three nearest of all is among the three nearest of those within the radius, and the other way round.
// Only the grid cells the radius reaches are looked at.
__device__ __forceinline__ void nn3_within(const Ctx &c, float qx, float qy, float r2, u64 &k0, u64 &k1, u64 &k2)
{
    k0 = k1 = k2 = ~0ull;
    const float r = sqrtf(r2) * 1.0001f + 1e-3f;
    if (!(r < 3e38f)) {  // (not on image coordinates) everything
        for (int t = 0; t < c.n; ++t) top3_insert(dist_key(qx, qy, c.gx[t], c.gy[t], c.gi[t]), k0, k1, k2);
        return;
    }
    const int xa = cell_x(c, qx - r), xb = cell_x(c, qx + r), ya = cell_y(c, qy - r), yb = cell_y(c, qy + r);
    for (int y = ya; y <= yb; ++y) {
        const int t0 = c.gstart[y * c.nx + xa], t1 = c.gstart[y * c.nx + xb + 1];
        for (int t = t0; t < t1; ++t) top3_insert(dist_key(qx, qy, c.gx[t], c.gy[t], c.gi[t]), k0, k1, k2);
    }
}

__device__ __forceinline__ float angle_degree(float v0x, float v0y, float v1x, float v1y)  // math_util.rs:31-33
{
    return fdlibm_atan2f(v1y * v0x - v1x * v0y, v0x * v1x + v0y * v1y) * 180.0f / kPiF;
}

// saddle.rs:26-38 "filter white block" for (s0, s1): 1 passes, 0 fails, 2 too close to a threshold to say here
__device__ __forceinline__ int white_block(float s0_theta, float v02x, float v02y)
{
    const float th = s0_theta / 180.0f * kPiF;
    double sd, cd;
    sincos((double)th, &sd, &cd);
    const double y = sd * (double)v02x - cd * (double)v02y, x = (double)v02x * cd + (double)v02y * sd;
    // s1 == s0 (try_expand_one pairs the same saddle with itself when the candidate lists overlap): both atan2f operands are
    // zeros whatever cosf / sinf return, the angle is 0 or 180
    if (v02x == 0.0f && v02y == 0.0f) return 0;
    const double m = fabs(y) + fabs(x);
    if (!(m > 0.0) || !(m < 1e300)) return 2;
    const double a = fabs(atan2(y, x)) * kDegD;
    if (a < 60.0 - kBandAbs || a > 120.0 + kBandAbs) return 0;
    return (a > 60.0 + kBandAbs && a < 120.0 - kBandAbs) ? 1 : 2;
}

// is_valid_quad (saddle.rs:17-67) without the white-block test: the tests are free of side effects, the cheap ones first
__device__ bool quad_rest(const Ctx &c, int i0, int i1, int i2, int i3)
{
    if (theta_dist(c.st[i1], c.st[i3]) > 5.0f) return false;  // :18-21
    const float s0x = c.sx[i0], s0y = c.sy[i0], d0x = c.sx[i1], d0y = c.sy[i1];
    const float s1x = c.sx[i2], s1y = c.sy[i2], d1x = c.sx[i3], d1y = c.sy[i3];
    const float v01x = d0x - s0x, v01y = d0y - s0y;
    const float v03x = d1x - s0x, v03y = d1y - s0y;
    const float v02x = s1x - s0x, v02y = s1y - s0y;
    if (cross2(v01x, v01y, v02x, v02y) * cross2(v02x, v02y, v03x, v03y) < 0.0f) return false;  // :44-46
    const float v12x = s1x - d0x, v12y = s1y - d0y;
    const float v23x = d1x - s1x, v23y = d1y - s1y;
    if (cross2(v01x, v01y, v12x, v12y) * cross2(v12x, v12y, v23x, v23y) < 0.0f) return false;  // :51-53
    if (dot2(v01x, v01y, v02x, v02y) < 0.0f || dot2(v03x, v03y, v02x, v02y) < 0.0f) return false;  // :62-64
    const float v30x = s0x - d1x, v30y = s0y - d1y;
    const float a0 = angle_degree(v01x, v01y, v12x, v12y), a2 = angle_degree(v23x, v23y, v30x, v30y);  // :55-61
    if (fabsf(a0 - a2) > 10.0f) return false;
    const float a1 = angle_degree(v12x, v12y, v23x, v23y), a3 = angle_degree(v30x, v30y, v01x, v01y);
    if (fabsf(a1 - a3) > 10.0f) return false;
    return true;
}
// is_valid_quad: 1 valid, 0 not, 2 everything but the white-block test passes and that one is undecided here
__device__ int valid_quad(const Ctx &c, int i0, int i1, int i2, int i3)
{
    if (!quad_rest(c, i0, i1, i2, i3)) return 0;
    return white_block(c.st[i0], c.sx[i2] - c.sx[i0], c.sy[i2] - c.sy[i0]);
}

// ---- a board in a lane's LDS slot --------------------------------------------------------------------------------

__device__ __forceinline__ u64 slot_quad(const uint8_t *slot, int cell)
{
    const uint32_t *p = reinterpret_cast<const uint32_t *>(slot + SL_QUAD) + 2 * cell;
    return (u64)p[1] << 32 | p[0];
}
__device__ __forceinline__ void slot_set_quad(uint8_t *slot, int cell, u64 q)
{
    uint32_t *p = reinterpret_cast<uint32_t *>(slot + SL_QUAD) + 2 * cell;
    p[0] = (uint32_t)q;
    p[1] = (uint32_t)(q >> 32);
}
__device__ __forceinline__ bool slot_active(const uint8_t *slot, int i)
{
    return (reinterpret_cast<const uint32_t *>(slot + SL_ACTIVE)[i >> 5] >> (i & 31)) & 1u;
}
__device__ __forceinline__ void slot_use(uint8_t *slot, int i)
{
    reinterpret_cast<uint32_t *>(slot + SL_ACTIVE)[i >> 5] &= ~(1u << (i & 31));
}
__device__ __forceinline__ int q_at(u64 q, int j) { return (int)((q >> (16 * j)) & 0xffffull); }
__device__ __forceinline__ u64 q_make(int a, int b, int c, int d) { return (u64)a | (u64)b << 16 | (u64)c << 32 | (u64)d << 48; }

// find_closest_potential_saddle_idxs (board.rs:177-233) for the ordered pair (i0, i1): candidates next to i0 into
// o[0..3), next to i1 into o[3..6) (LDS), counts returned
__device__ __forceinline__ void closest_pair(const Ctx &c, const uint8_t *slot, int i0, int i1, uint16_t *o, int &n0, int &n1)
{
    const float s0x = c.sx[i0], s0y = c.sy[i0], s1x = c.sx[i1], s1y = c.sy[i1];
    const float ratio0 = 1.0f + 0.3f;
    const float ex = s0x - s1x, ey = s0y - s1y;
    const float radius_sq = 0.5f * (ex * ex + ey * ey);
    const float v10x = s1x - s0x, v10y = s1y - s0y;
    n0 = n1 = 0;
    u64 k0, k1, k2;
    nn3_within(c, s0x + v10x * ratio0, s0y + v10y * ratio0, radius_sq, k0, k1, k2);
    {
        const float t0 = c.st[i0];
        const u64 ks[3] = {k0, k1, k2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = (int)(uint32_t)ks[i];
            if (ks[i] != ~0ull && __uint_as_float((uint32_t)(ks[i] >> 32)) <= radius_sq && slot_active(slot, idx) &&
                theta_dist(t0, c.st[idx]) < 5.0f)
                o[n0++] = (uint16_t)idx;
        }
    }
    if (!n0) return;  // (try_expand_one's loops are empty whatever the other list holds)
    nn3_within(c, s1x + v10x * ratio0, s1y + v10y * ratio0, radius_sq, k0, k1, k2);
    {
        const float t1 = c.st[i1];
        const u64 ks[3] = {k0, k1, k2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = (int)(uint32_t)ks[i];
            if (ks[i] != ~0ull && __uint_as_float((uint32_t)(ks[i] >> 32)) <= radius_sq && slot_active(slot, idx) &&
                theta_dist(t1, c.st[idx]) < 5.0f)
                o[3 + n1++] = (uint16_t)idx;
        }
    }
}

// try_expand_one (board.rs:153-176); qs = the quad rotated as try_expand passes it
__device__ bool expand_one(const Ctx &c, uint8_t *slot, u64 qs, u64 &out, uint32_t &status)
{
    uint16_t *tmp = reinterpret_cast<uint16_t *>(slot + SL_TMP);
    int n0, n1, n2, n3;
    closest_pair(c, slot, q_at(qs, 0), q_at(qs, 1), tmp, n0, n1);
    if (n0 == 0 || n1 == 0) return false;
    closest_pair(c, slot, q_at(qs, 3), q_at(qs, 2), tmp + 6, n3, n2);  // (s3's candidates at tmp[6..9), s2's at tmp[9..12))
    if (n3 == 0 || n2 == 0) return false;
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1)
            for (int i2 = 0; i2 < n2; ++i2)
                for (int i3 = 0; i3 < n3; ++i3) {
                    const int a = tmp[i0], b = tmp[3 + i1], cc = tmp[9 + i2], d = tmp[6 + i3];
                    const int v = valid_quad(c, a, b, cc, d);
                    if (v == 1) {
                        out = q_make(a, b, cc, d);
                        return true;
                    }
                    if (v == 2) status |= TAIL_UNCERTAIN | (1u << 8);
                }
    return false;
}

// Board::new (board.rs:26-48): the board grown from a seed quad; returns its score, the cells stay in the slot
__device__ int build_board(const Ctx &c, uint8_t *slot, u64 seed, int &n_cells_out, uint32_t &status)
{
    {
        uint32_t *act = reinterpret_cast<uint32_t *>(slot + SL_ACTIVE);
        for (int i = 0; i < TN / 32; ++i) act[i] = 0xffffffffu;
        uint32_t *g = reinterpret_cast<uint32_t *>(slot + SL_GRID);
        for (int i = 0; i < (SL_ACTIVE - SL_GRID) / 4; ++i) g[i] = 0xffffffffu;
    }
    uint8_t *grid = slot + SL_GRID, *found = slot + SL_FOUND, *stack = slot + SL_STACK;
    int8_t *xy = reinterpret_cast<int8_t *>(slot + SL_XY);
    for (int j = 1; j < 4; ++j) slot_use(slot, q_at(seed, j));  // :35-37
    int n_cells = 1, score = 1;
    slot_set_quad(slot, 0, seed);
    xy[0] = 0;
    xy[1] = 0;
    found[0] = 1;
    grid[BGR * BGN + BGR] = 0;
    stack[0] = 0;
    stack[1] = 0;
    int sp = 1;
    while (sp > 0) {  // try_expand (:114-152), its recursion as a stack of (cell, next direction)
        const int cell = stack[2 * (sp - 1)], i = stack[2 * (sp - 1) + 1];
        if (i == 4) {
            --sp;
            continue;
        }
        stack[2 * (sp - 1) + 1] = (uint8_t)(i + 1);
        const int nx = xy[2 * cell] + (i == 0 ? 1 : (i == 2 ? -1 : 0)), ny = xy[2 * cell + 1] + (i == 1 ? -1 : (i == 3 ? 1 : 0));
        if (nx < -BGR || nx > BGR || ny < -BGR || ny > BGR) {
            status |= TAIL_CAPACITY | (1u << 11);
            break;
        }
        const int gpos = (ny + BGR) * BGN + (nx + BGR);
        const int e = grid[gpos];
        if (e != 0xff && found[e]) continue;
        const u64 quad = slot_quad(slot, cell);
        const u64 qs = i ? (quad >> (16 * i) | quad << (64 - 16 * i)) : quad;  // qs[j] = quad[(j + i) & 3]
        u64 nq;
        const bool ok = expand_one(c, slot, qs, nq, status);
        int at = e;
        if (at == 0xff) {
            if (n_cells == BCELLS) {
                status |= TAIL_CAPACITY | (1u << 12);
                break;
            }
            at = n_cells++;
            grid[gpos] = (uint8_t)at;
            xy[2 * at] = (int8_t)nx;
            xy[2 * at + 1] = (int8_t)ny;
        }
        if (ok) {
            const u64 v = i ? (nq << (16 * i) | nq >> (64 - 16 * i)) : nq;  // v[(j + i) & 3] = nq[j]
            for (int j = 0; j < 4; ++j) slot_use(slot, q_at(v, j));
            ++score;
            slot_set_quad(slot, at, v);
            found[at] = 1;
            stack[2 * sp] = (uint8_t)at;
            stack[2 * sp + 1] = 0;
            ++sp;  // (depth <= found cells <= BCELLS)
        } else {
            slot_set_quad(slot, at, 0ull);
            found[at] = 0;
        }
    }
    n_cells_out = n_cells;
    return score;
}

// ---- decode (detector.rs:42-169, 448-476; image_util.rs:39-70) -----------------------------------------------------

__device__ u64 rotate_bits(u64 bits, int edge_bits)
{
    u64 out = 0;
    int count = 0;
    for (int r = edge_bits - 1; r >= 0; --r)
        for (int cc = 0; cc < edge_bits; ++cc, ++count) out |= ((bits >> (r + cc * edge_bits)) & 1ull) << count;
    return out;
}

__device__ bool decode_quad(const TailArgs &a, const uint8_t *luma, const float q[8], int &tag_id, int &rot_out)
{
    const uint32_t w = (uint32_t)a.W, h = (uint32_t)a.H;
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = f32_as_u32(round_half_away(q[2 * i])), y = f32_as_u32(round_half_away(q[2 * i + 1]));
        if (x >= w || y >= h) return false;
    }
    // tag_affine: least squares over the corners of an axis-aligned square, in binary64, rounded once (host_tail.cpp)
    float aff[6];
    {
        const int side_bits = a.border * 2 + a.edge;
        const double S = (double)((float)side_bits - 1.0f + 0.5f), m = 0.5;
        const double cc = 0.5 * (S - m), aa = 0.5 * (S + m);
        const double su[4] = {-aa, -aa, aa, aa}, sv[4] = {-aa, aa, aa, -aa};
#pragma unroll
        for (int axis = 0; axis < 2; ++axis) {
            double gu = 0, gv = 0, mean = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const double t = q[2 * p + axis];
                gu += su[p] * t;
                gv += sv[p] * t;
                mean += t;
            }
            const double hu = gu / (4.0 * aa * aa), hv = gv / (4.0 * aa * aa);
            aff[3 * axis + 0] = (float)hu;
            aff[3 * axis + 1] = (float)hv;
            aff[3 * axis + 2] = (float)(mean / 4.0 - hu * cc - hv * cc);
        }
    }
    const int nb = a.edge * a.edge;
    u64 samples_lo = 0;  // sample values are compared twice: keep them (8 bits each would need 36 bytes) -- second pass re-reads
    (void)samples_lo;
    int lo = 255, hi = 0;
    for (int gx = a.border; gx < a.border + a.edge; ++gx)
        for (int gy = a.border; gy < a.border + a.edge; ++gy) {
            const float fx = (float)gx, fy = (float)gy;
            const float px = aff[0] * fx + aff[1] * fy + aff[2] * 1.0f;
            const float py = aff[3] * fx + aff[4] * fy + aff[5] * 1.0f;
            const uint32_t ix = f32_as_u32(round_half_away(px)), iy = f32_as_u32(round_half_away(py));
            if (ix >= w || iy >= h) return false;
            const int b = luma[(size_t)iy * (size_t)a.luma_row_stride + ix];
            lo = b < lo ? b : lo;
            hi = b > hi ? b : hi;
        }
    if (hi - lo < 50) return false;
    const int mid = (int)(uint8_t)f32_as_u32(round_half_away(((float)lo + (float)hi) / 2.0f));
    u64 bits = 0;
    uint32_t invalid = 0;
    {
        int n = 0;  // sample number in the reference's order; the first sample is the most significant bit
        for (int gx = a.border; gx < a.border + a.edge; ++gx)
            for (int gy = a.border; gy < a.border + a.edge; ++gy, ++n) {
                const float fx = (float)gx, fy = (float)gy;
                const float px = aff[0] * fx + aff[1] * fy + aff[2] * 1.0f;
                const float py = aff[3] * fx + aff[4] * fy + aff[5] * 1.0f;
                const uint32_t ix = f32_as_u32(round_half_away(px)), iy = f32_as_u32(round_half_away(py));
                const int b = luma[(size_t)iy * (size_t)a.luma_row_stride + ix];
                const int d = mid - b;
                if ((d < 0 ? -d : d) < 10) ++invalid;
                if (b > mid) bits |= 1ull << (nb - 1 - n);
            }
    }
    if (invalid > 3) return false;
    for (int rotated = 0; rotated < 4; ++rotated) {  // best_tag :142-169
        int best = 0;
        unsigned best_score = (unsigned)__popcll(a.codes[0] ^ bits);
        for (int i = 1; i < a.n_codes; ++i) {
            const unsigned s = (unsigned)__popcll(a.codes[i] ^ bits);
            if (s < best_score) {
                best_score = s;
                best = i;
            }
        }
        if (best_score < (unsigned)a.hamming) {
            tag_id = best;
            rot_out = rotated;
            return true;
        }
        if (rotated == 3) break;
        bits = rotate_bits(bits, a.edge);
    }
    return false;
}

// ---- wave helpers ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ float wave_min_f(float v)
{
    for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v)
{
    for (int o = 32; o; o >>= 1) {
        const uint32_t w = (uint32_t)__shfl_xor((int)v, o);
        v = w > v ? w : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_or_u(uint32_t v)
{
    for (int o = 32; o; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o);
    return v;
}

__global__ void __launch_bounds__(64) k_board_tail(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int f = blockIdx.x, lane = threadIdx.x;
    const u64 below = (1ull << lane) - 1ull;
    float *sx = reinterpret_cast<float *>(lds + OFF_SX), *sy = reinterpret_cast<float *>(lds + OFF_SY), *st = reinterpret_cast<float *>(lds + OFF_ST);
    float *gx = reinterpret_cast<float *>(lds + OFF_GX), *gy = reinterpret_cast<float *>(lds + OFF_GY);
    uint16_t *gi = reinterpret_cast<uint16_t *>(lds + OFF_GI), *gstart = reinterpret_cast<uint16_t *>(lds + OFF_GSTART);
    uint16_t *seeds = reinterpret_cast<uint16_t *>(lds + OFF_SEEDS);
    u64 *cand = reinterpret_cast<u64 *>(lds + OFF_CAND);
    uint32_t *tmp32 = reinterpret_cast<uint32_t *>(lds + OFF_CAND);
    uint16_t *pairs = reinterpret_cast<uint16_t *>(lds + OFF_PAIRS);
    uint16_t *same = reinterpret_cast<uint16_t *>(lds + OFF_SMALL), *diff = same + 64, *s1ok = same + 128;
    u64 *quads = reinterpret_cast<u64 *>(lds + OFF_QUADS);
    uint32_t *tagids = reinterpret_cast<uint32_t *>(lds + OFF_TAGIDS);
    uint32_t *used = reinterpret_cast<uint32_t *>(lds + OFF_USED);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + OFF_HIST);
    uint8_t *boards = lds + OFF_BOARDS;
    uint8_t *best_slot = boards + TB * SL_BYTES;

    uint32_t status = 0;  // per lane; merged at the end
    int n_tags = 0;
    const FrameCounters &fc = a.ctr[f];
    int n = (int)fc.n_out;
    const uint32_t cflags = fc.flags;
    if (cflags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) {
        if (lane == 0) {
            a.table[2 * f] = 0;
            a.table[2 * f + 1] = TAIL_CHAIN;
        }
        return;
    }
    if (n > TN) {
        if (lane == 0) {
            a.table[2 * f] = 0;
            a.table[2 * f + 1] = TAIL_CAPACITY;
        }
        return;
    }
    {
        const float *src = a.saddles + (size_t)fc.out_offset * 5;
        for (int i = lane; i < n; i += 64) {
            const float x = src[5 * i], y = src[5 * i + 1], t = src[5 * i + 3];
            sx[i] = x;
            sy[i] = y;
            st[i] = t;
            // coordinates of an image and half an atan2 in degrees; anything else (NaN included) is not this kernel's business
            if (!(fabsf(x) < 1e6f && fabsf(y) < 1e6f && t >= -180.0f && t <= 180.0f)) status |= TAIL_CAPACITY | (1u << 13);
        }
    }
    const uint8_t *luma = a.luma + (size_t)f * (size_t)a.luma_frame_stride;
    const uint32_t tag_cap = a.tag_cap < (uint32_t)TTAGS ? a.tag_cap : (uint32_t)TTAGS;
    __syncthreads();
    if (wave_or_u(status)) n = 0;  // (uniform) nothing is searched; the status goes out below

    for (int round = 0; round < a.max_boards && n > 0; ++round) {
        // ---- the k-NN grid over this round's saddles -----------------------------------------------------------------
        Ctx c;
        c.sx = sx; c.sy = sy; c.st = st; c.gx = gx; c.gy = gy; c.gi = gi; c.gstart = gstart; c.n = n;
        {
            float x0 = 3e38f, x1 = -3e38f, y0 = 3e38f, y1 = -3e38f;
            for (int i = lane; i < n; i += 64) {
                x0 = fminf(x0, sx[i]); x1 = fmaxf(x1, sx[i]);
                y0 = fminf(y0, sy[i]); y1 = fmaxf(y1, sy[i]);
            }
            x0 = wave_min_f(x0); x1 = wave_max_f(x1); y0 = wave_min_f(y0); y1 = wave_max_f(y1);
            const float w = fmaxf(1e-3f, x1 - x0), h = fmaxf(1e-3f, y1 - y0);
            float cell = fmaxf(1.0f, sqrtf(w * h / (float)n));
            int nx = (int)(w / cell) + 1, ny = (int)(h / cell) + 1;
            while (nx * ny > TGC) {
                cell *= 1.5f;
                nx = (int)(w / cell) + 1;
                ny = (int)(h / cell) + 1;
            }
            c.ox = x0; c.oy = y0; c.inv_cell = 1.0f / cell; c.nx = nx; c.ny = ny;
            const int ncell = nx * ny;
            uint32_t *cnt = tmp32, *fill = tmp32 + TGC;
            for (int i = lane; i < ncell; i += 64) cnt[i] = 0;
            __syncthreads();
            for (int i = lane; i < n; i += 64) atomicAdd(&cnt[cell_y(c, sy[i]) * nx + cell_x(c, sx[i])], 1u);
            __syncthreads();
            // exclusive scan: 16 consecutive cells per lane
            uint32_t local = 0;
            for (int k = 0; k < TGC / 64; ++k) {
                const int ci = lane * (TGC / 64) + k;
                if (ci < ncell) local += cnt[ci];
            }
            uint32_t incl = local;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
                if (lane >= o) incl += v;
            }
            uint32_t run = incl - local;
            for (int k = 0; k < TGC / 64; ++k) {
                const int ci = lane * (TGC / 64) + k;
                if (ci < ncell) {
                    gstart[ci] = (uint16_t)run;
                    fill[ci] = run;
                    run += cnt[ci];
                }
            }
            if (lane == 0) gstart[ncell] = (uint16_t)n;
            __syncthreads();
            for (int i = lane; i < n; i += 64) {
                const uint32_t pos = atomicAdd(&fill[cell_y(c, sy[i]) * nx + cell_x(c, sx[i])], 1u);
                gi[pos] = (uint16_t)i;
                gx[pos] = sx[i];
                gy[pos] = sy[i];
            }
            __syncthreads();
        }

        // ---- seeds: the most populated round(theta) bin (ties: the smallest angle), in index order -------------------
        int n_seeds = 0;
        {
            for (int i = lane; i < 364; i += 64) hist[i] = 0;
            __syncthreads();
            for (int i = lane; i < n; i += 64) atomicAdd(&hist[(int)round_half_away(st[i]) + 180], 1u);
            __syncthreads();
            uint32_t bestk = 0;
            for (int i = lane; i < 361; i += 64) {
                const uint32_t k = hist[i] << 16 | (uint32_t)(0xffff - i);
                bestk = k > bestk ? k : bestk;
            }
            bestk = wave_max_u(bestk);
            const int best_angle = (int)(0xffff - (bestk & 0xffffu)) - 180;
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                const bool is = i < n && (int)round_half_away(st[i]) == best_angle;
                const u64 m = __ballot(is);
                if (is) seeds[n_seeds + __popcll(m & below)] = (uint16_t)i;
                n_seeds += __popcll(m);
            }
            __syncthreads();
        }

        // ---- try_find_best_board's loop over the seeds (detector.rs:613-629) -------------------------------------------
        int best_score = 0, best_cells = 0;
        int count = 0;
        while (n_seeds > 0 && count < 30) {
            const int s0 = seeds[--n_seeds];
            const float s0x = sx[s0], s0y = sy[s0], s0t = st[s0];
            int nc = 0;
            // init_quads (:543-586).  50 nearest: every distance key, sorted
            int P = 64;
            while (P < n) P <<= 1;
            u64 *keys = cand;
            for (int i = lane; i < P; i += 64) keys[i] = i < n ? dist_key(s0x, s0y, sx[i], sy[i], (uint32_t)i) : ~0ull;
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = lane; i < P; i += 64) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const u64 ka = keys[i], kb = keys[ixj];
                            if ((ka > kb) == ((i & k) == 0)) {
                                keys[i] = kb;
                                keys[ixj] = ka;
                            }
                        }
                    }
                    __syncthreads();
                }
            const int m = n < 50 ? n : 50;
            int ns = 0, nd = 0;
            {
                const int idx = (lane >= 1 && lane < m) ? (int)(uint32_t)keys[lane] : 0;
                const float td = theta_dist(s0t, st[idx]);
                const bool is_s = lane >= 1 && lane < m && td < 5.0f;
                const bool is_d = lane >= 1 && lane < m && !is_s && td > 80.0f;
                const u64 ms = __ballot(is_s), md = __ballot(is_d);
                if (is_s) same[__popcll(ms & below)] = (uint16_t)idx;
                if (is_d) diff[__popcll(md & below)] = (uint16_t)idx;
                ns = __popcll(ms);
                nd = __popcll(md);
            }
            __syncthreads();  // (keys are dead from here: cand is written below)
            int n_pairs = 0, n_s1 = 0;
            if (ns > 0 && nd >= 2) {
                // the (d0, d1) combinations in the reference's order (itertools combinations(2)) that pass :18-21
                for (int p = 0; p < nd; ++p) {
                    const bool ok = lane > p && lane < nd && !(theta_dist(st[diff[p]], st[diff[lane < nd ? lane : 0]]) > 5.0f);
                    const u64 mk = __ballot(ok);
                    if (ok) pairs[n_pairs + __popcll(mk & below)] = (uint16_t)(p | lane << 8);
                    n_pairs += __popcll(mk);
                }
                // the white-block test depends on (s0, s1) only: once per s1 (s1ok: 0 fails, 1 passes, 2 undecided here)
                if (lane < ns) s1ok[lane] = (uint16_t)white_block(s0t, sx[same[lane]] - s0x, sy[same[lane]] - s0y);
                n_s1 = ns;
            }
            __syncthreads();
            if (n_pairs > 0) {
                for (int si = 0; si < n_s1; ++si) {
                    const int wb = s1ok[si];
                    if (wb == 0) continue;
                    const int s1 = same[si];
                    const float v02x = sx[s1] - s0x, v02y = sy[s1] - s0y;
                    for (int base = 0; base < n_pairs; base += 64) {
                        const int p = base + lane;
                        bool ok = false;
                        u64 q = 0;
                        if (p < n_pairs) {
                            const int pa = pairs[p] & 0xff, pb = pairs[p] >> 8;
                            const int d0 = diff[pa], d1 = diff[pb];
                            ok = quad_rest(c, s0, d0, s1, d1);
                            if (ok && wb == 2) {  // a quad hangs on the undecided test
                                status |= TAIL_UNCERTAIN | (1u << 9);
                                ok = false;
                            }
                            const float c0 = cross2(sx[d0] - s0x, sy[d0] - s0y, v02x, v02y);
                            q = c0 > 0.0f ? q_make(s0, d0, s1, d1) : q_make(s0, d1, s1, d0);
                        }
                        const u64 mk = __ballot(ok);
                        const int at = nc + __popcll(mk & below);
                        if (ok && at < TCAND) cand[at] = q;
                        nc += __popcll(mk);
                    }
                }
                if (nc > TCAND) {
                    status |= TAIL_CAPACITY | (1u << 14);
                    nc = TCAND;
                }
            }
            __syncthreads();
            // a board per candidate quad, in order; the first one that beats the best so far is kept (:616-622)
            for (int base = 0; base < nc; base += TB) {
                int score = 0, cells = 0;
                if (lane < TB && base + lane < nc) score = build_board(c, boards + lane * SL_BYTES, cand[base + lane], cells, status);
                const uint32_t key = wave_max_u((uint32_t)score << 8 | (uint32_t)(63 - lane));
                const int top = (int)(key >> 8), who = 63 - (int)(key & 0xffu);
                if (top > best_score) {  // (uniform)
                    best_score = top;
                    best_cells = __shfl(cells, who);
                    __syncthreads();
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(boards + who * SL_BYTES);
                    uint32_t *dst = reinterpret_cast<uint32_t *>(best_slot);
                    for (int i = lane; i < SL_GRID / 4; i += 64) dst[i] = src[i];  // quads, coordinates, found flags
                }
                __syncthreads();
            }
            if (best_score >= 36) break;
            ++count;
            if (wave_or_u(status)) break;  // the frame goes to the host anyway
        }
        if (wave_or_u(status)) break;
        if (best_score == 0) break;  // None: the remaining rounds would find nothing either

        // ---- try_fix_missing + all_tag_indexes (board.rs:49-112), cells in insertion order ----------------------------
        int n_quads = 0;
        if (lane == 0) {
            const int8_t *xy = reinterpret_cast<const int8_t *>(best_slot + SL_XY);
            uint8_t *found = best_slot + SL_FOUND;
            auto find = [&](int x, int y) -> int {
                for (int i = 0; i < best_cells; ++i)
                    if (xy[2 * i] == x && xy[2 * i + 1] == y) return i;
                return -1;
            };
            uint8_t *fa = best_slot + SL_STACK, *fb = fa + BCELLS;  // the fix list: cell numbers of the two found neighbours, and of the cell
            uint8_t *fc_ = best_slot + SL_GRID;
            int n_fix = 0;
            for (int i = 0; i < best_cells; ++i) {
                if (found[i]) continue;
                const int x = xy[2 * i], y = xy[2 * i + 1];
                const int c0 = find(x + 1, y), c1 = find(x - 1, y);
                if (c0 >= 0 && c1 >= 0) {
                    if (found[c0] && found[c1]) { fa[n_fix] = (uint8_t)c0; fb[n_fix] = (uint8_t)c1; fc_[n_fix] = (uint8_t)i; ++n_fix; }
                } else {
                    const int c2 = find(x, y + 1), c3 = find(x, y - 1);
                    if (c2 >= 0 && c3 >= 0 && found[c2] && found[c3]) { fa[n_fix] = (uint8_t)c2; fb[n_fix] = (uint8_t)c3; fc_[n_fix] = (uint8_t)i; ++n_fix; }
                }
            }
            for (int k = 0; k < n_fix; ++k) {
                const u64 q0 = slot_quad(best_slot, fa[k]), q1 = slot_quad(best_slot, fb[k]);
                int mid[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int i0 = q_at(q0, i), i1 = q_at(q1, i);
                    const float x = (sx[i0] + sx[i1]) / 2.0f, y = (sy[i0] + sy[i1]) / 2.0f;
                    u64 bestk = ~0ull;
                    for (int t = 0; t < n; ++t) {
                        const u64 kk = dist_key(x, y, sx[t], sy[t], (uint32_t)t);
                        bestk = kk < bestk ? kk : bestk;
                    }
                    mid[i] = (int)(uint32_t)bestk;
                }
                const int v = valid_quad(c, mid[0], mid[1], mid[2], mid[3]);
                if (v == 2) status |= TAIL_UNCERTAIN | (1u << 10);
                if (v == 1) {  // the cell between the two is the missing one ((b0 + b1) / 2, :100): it exists, not found
                    slot_set_quad(best_slot, fc_[k], q_make(mid[0], mid[1], mid[2], mid[3]));
                    found[fc_[k]] = 1;
                }
            }
            for (int i = 0; i < best_cells; ++i)
                if (found[i]) quads[n_quads++] = slot_quad(best_slot, i);
        }
        n_quads = __shfl(n_quads, 0);
        __syncthreads();
        if (wave_or_u(status)) break;

        // ---- decode the board's quads (detector.rs:514-527); results in the candidates' space -------------------------
        float *dec_xy = reinterpret_cast<float *>(lds + OFF_CAND);          // [BCELLS][8]
        int *dec_id = reinterpret_cast<int *>(lds + OFF_CAND + BCELLS * 32);  // [BCELLS]: tag id or -1
        for (int base = 0; base < n_quads; base += 64) {
            const int qi = base + lane;
            if (qi < n_quads) {
                const u64 q = quads[qi];
                float qxy[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qxy[2 * i] = sx[q_at(q, i)];
                    qxy[2 * i + 1] = sy[q_at(q, i)];
                }
                int id = -1, rot = 0;
                if (!decode_quad(a, luma, qxy, id, rot)) id = -1;
                dec_id[qi] = id;
                if (id >= 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {  // rotate_left(rot) then reverse, :468-469
                        const int src = ((3 - i) + rot) & 3;
                        float vx = qxy[0], vy = qxy[1];
                        if (src == 1) { vx = qxy[2]; vy = qxy[3]; }
                        if (src == 2) { vx = qxy[4]; vy = qxy[5]; }
                        if (src == 3) { vx = qxy[6]; vy = qxy[7]; }
                        dec_xy[8 * qi + 2 * i] = vx;
                        dec_xy[8 * qi + 2 * i + 1] = vy;
                    }
                }
            }
        }
        for (int i = lane; i < TN / 32; i += 64) used[i] = 0;
        __syncthreads();
        int n_used = 0;
        if (lane == 0) {
            for (int qi = 0; qi < n_quads; ++qi) {
                const int id = dec_id[qi];
                if (id < 0) continue;
                int at = -1;
                for (int t = 0; t < n_tags; ++t)
                    if (tagids[t] == (uint32_t)id) at = t;
                if (at < 0) {
                    if ((uint32_t)n_tags >= tag_cap) {
                        status |= TAIL_CAPACITY | (1u << 15);
                        break;
                    }
                    at = n_tags++;
                    tagids[at] = (uint32_t)id;
                }
                agx_tag *o = a.tags + (size_t)f * a.tag_cap + at;
                o->id = (uint32_t)id;
                for (int i = 0; i < 8; ++i) o->xy[i] = dec_xy[8 * qi + i];
                const u64 q = quads[qi];
                for (int i = 0; i < 4; ++i) {
                    const int s = q_at(q, i);
                    if (!((used[s >> 5] >> (s & 31)) & 1u)) ++n_used;
                    used[s >> 5] |= 1u << (s & 31);
                }
            }
        }
        n_tags = __shfl(n_tags, 0);
        n_used = __shfl(n_used, 0);
        __syncthreads();
        if (wave_or_u(status)) break;
        if (n_used == 0) break;  // nothing removed: the next round would repeat this one
        // the saddles of decoded quads leave the list (:528-538), order kept
        int kept = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const bool keep = i < n && !((used[i >> 5] >> (i & 31)) & 1u);
            const float x = i < n ? sx[i] : 0.0f, y = i < n ? sy[i] : 0.0f, t = i < n ? st[i] : 0.0f;
            const u64 mk = __ballot(keep);
            __syncthreads();
            if (keep) {
                const int at = kept + __popcll(mk & below);
                sx[at] = x;
                sy[at] = y;
                st[at] = t;
            }
            kept += __popcll(mk);
            __syncthreads();
        }
        n = kept;
    }
    status = wave_or_u(status);
    if (lane == 0) {
        a.table[2 * f] = status ? 0u : (uint32_t)n_tags;
        a.table[2 * f + 1] = status;
    }
}

}  // namespace

int init_tail_kernels()
{
    return (int)hipFuncSetAttribute((const void *)k_board_tail, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

int launch_board_tail(const TailArgs &t, void *stream)
{
    if (t.n_frames <= 0) return (int)hipSuccess;
    hipLaunchKernelGGL(k_board_tail, dim3((unsigned)t.n_frames), dim3(64), LDS_BYTES, (hipStream_t)stream, t);
    return (int)hipGetLastError();
}

}  // namespace agx
