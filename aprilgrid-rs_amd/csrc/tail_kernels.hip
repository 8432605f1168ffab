// tail_kernels.hip -- TagDetector::detect's board search and tag decode on the device (gfx950).
//
// What the reference does after refined_saddle_points (src/detector.rs:510-539): up to max_num_of_boards rounds of
// try_find_best_board (:588-639; init_quads :543-586, board::Board src/board.rs, is_valid_quad src/saddle.rs:17-67) and
// try_decode_quad (:448-476) over a few hundred saddles per frame.  The chain leaves those saddles in device memory; this
// kernel runs the same search there, one workgroup per frame, so that a batch's tags -- a few KB -- are all that crosses
// PCIe and no host thread spends a millisecond per frame on it (host_tail.cpp is the same algorithm on the host and stays
// the reference-exact arbiter, below).
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
// Mapping.  One workgroup of eight waves per frame (a frame = a CU); the frame's saddles, a uniform-grid index for the 3-NN
// queries of find_closest_potential_saddle_idxs (src/board.rs:177-233), two memo tables and the family's codes are shared in
// LDS, every wave has its own candidate list, angle tables, the board it is growing and the best board it has grown (155 KB).
//   seeds     the first seed of try_find_best_board alone (it usually ends the loop), then eight at a time, one wave each; what
//             a seed contributes to the sequential loop (:613-629) is its best score and the first quad that reaches it, so the
//             merge walks the group's results in the reference's order with the reference's rules (strict improvement, stop at
//             >= 36, 30 seeds) -- as host_tail.cpp's find_best_board_parallel does on threads;
//   init_quads  50-NN by a bitonic sort of the distance keys; the same / different orientation lists and the (d0, d1)
//             combinations by ballot + prefix count, in the reference's order; a lane per combination, the angle terms that
//             depend on fewer than four saddles from tables;
//   boards    the group's candidate quads are handed out to the waves from a common counter; a board is grown by the whole wave:
//             board.rs's recursion is a stack walked in lock step (its top in registers), the four 3-NN queries of a
//             try_expand_one run on 16 lanes each over the grid cells their radius reaches (memoised per pair), its up to 81
//             candidate quadruples are tested one per lane (is_valid_quad memoised) and the first valid one in the reference's
//             loop order is taken; the wave that grew the chosen board keeps its cells;
//   the rest  (try_fix_missing, decode -- best_tag by rows of 16 lanes --, the tag map, removing the used saddles) on the first wave.
#include <hip/hip_runtime.h>

#include "libm_f32.h"
#include "tail_kernels.h"

namespace agx {
namespace {

constexpr int TN = TAIL_MAX_SADDLES;
constexpr int TGC = 1024;    // cells of the k-NN grid
constexpr int TCAND = 256;   // candidate quads of one seed
constexpr int TA3 = 512;     // (d0, d1) pairs of a seed whose angle a3 is kept in a table
constexpr int TMEMO = 2048;  // entries of each of the two memo tables (direct mapped)
constexpr int TW = 8;        // waves per frame
constexpr int BCELLS = 128;  // cells (found or not) of one board
constexpr int BGR = 12, BGN = 2 * BGR + 1;  // board cells live within +-BGR of the seed's cell
constexpr int TTAGS = 128;   // distinct tag ids of one frame

// a board's slot in LDS (bytes)
constexpr int SL_QUAD = 0;       // u16[BCELLS][4]
constexpr int SL_XY = 1024;      // i8[BCELLS][2]
constexpr int SL_FOUND = 1280;   // u8[BCELLS]
constexpr int SL_GRID = 1408;    // u8[BGN * BGN] cell coordinates -> cell number (0xff none)
constexpr int SL_ACTIVE = 2048;  // u32[TN / 32]: board.rs active_idxs
constexpr int SL_STACK = SL_ACTIVE + TN / 8;  // u8[BCELLS][2]: cell, next direction
constexpr int SL_BYTES = SL_STACK + 2 * BCELLS;
static_assert(BGN * BGN <= SL_ACTIVE - SL_GRID, "board grid");
static_assert(TN / 8 <= SL_STACK - SL_ACTIVE && SL_BYTES % 8 == 0 && TN <= 1024, "active mask; an index is 10 bits in the memo keys");

// a wave's own LDS (bytes)
constexpr int WV_CAND = 0;                      // u64[TCAND]; with what lies behind it (up to the kept board) also u64[TN] distance keys; with the table, on wave 0, the decode's corners
constexpr int WV_A3 = WV_CAND + TCAND * 8;      // f32[TA3]: angle(v30, v01) of the seed's (d0, d1) pairs
constexpr int WV_PAIRS = WV_A3 + TA3 * 4;       // u16[1176 + pad]; wave 0: the decode's bits and ids
constexpr int WV_SMALL = WV_PAIRS + 1184 * 2;   // u16[3][64]: same, diff, the white-block test per s1; f32[2][64]: a0, a2 of the current s1
constexpr int WV_SLOT = WV_SMALL + 1024;        // the board under construction
constexpr int WV_KEEP = WV_SLOT + SL_BYTES;     // cells (quads, coordinates, found flags) of the best board this wave has grown in the round
constexpr int WV_BYTES = WV_KEEP + SL_GRID;
static_assert(WV_BYTES % 8 == 0 && WV_KEEP - WV_CAND >= TN * 8, "alignment / the keys' space (the candidate list, the tables behind it and the board: all dead while a seed's distances are sorted)");
// the frame's LDS (bytes)
constexpr int OFF_SX = 0, OFF_SY = OFF_SX + TN * 4, OFF_ST = OFF_SY + TN * 4;
constexpr int OFF_GX = OFF_ST + TN * 4, OFF_GY = OFF_GX + TN * 4, OFF_GI = OFF_GY + TN * 4;
constexpr int OFF_GSTART = OFF_GI + TN * 2;                // u16[TGC + 1]
constexpr int OFF_SEEDS = OFF_GSTART + (TGC + 4) * 2;      // u16[TN]
constexpr int OFF_QUADS = OFF_SEEDS + TN * 2;              // u64[BCELLS]
constexpr int OFF_TAGIDS = OFF_QUADS + BCELLS * 8;         // u32[TTAGS]
constexpr int OFF_USED = OFF_TAGIDS + TTAGS * 4;           // u32[TN / 32]
constexpr int OFF_HIST = OFF_USED + TN / 8;                // u32[364]
constexpr int OFF_SHARED = OFF_HIST + 364 * 4;             // u32[8 + 6 * TW]: what the waves tell each other (the per-group words twice: groups alternate)
// Boards grown from different seed quads ask the same questions again: what find_closest_potential_saddle_idxs finds for an
// ordered pair of saddles before the board's own "still unused" test (a function of the pair), and is_valid_quad of four
// saddles (a function of the four).  Both are kept per round in direct-mapped tables shared by the frame's waves -- an
// entry is one aligned 64-bit word carrying its whole key, read and written atomically; a collision just overwrites.
constexpr int OFF_MEMO_P = OFF_SHARED + (8 + 6 * TW) * 4;  // u64[TMEMO]: key (i0, i1, side) -> up to three candidates
constexpr int OFF_MEMO_Q = OFF_MEMO_P + TMEMO * 8;         // u64[TMEMO]: key (four indices) -> is_valid_quad's 0 / 1 / 2
constexpr int OFF_CODES = OFF_MEMO_Q + TMEMO * 8;          // u64[TCODES]: the family's code list (best_tag reads all of it per quad and rotation)
constexpr int TCODES = 640;
constexpr int OFF_WAVES = OFF_CODES + TCODES * 8;
constexpr int LDS_BYTES = OFF_WAVES + TW * WV_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS of a CU");
static_assert(2 * TGC * 4 <= TMEMO * 8, "the grid is built in the first memo table's space");
static_assert(OFF_QUADS % 8 == 0 && OFF_SHARED % 8 == 0 && OFF_MEMO_P % 8 == 0 && OFF_WAVES % 8 == 0, "alignment");

constexpr float kPiF = 3.14159274101257324219f;
// The white-block angle: cosf / sinf within 1 ulp move the direction by < 1.2e-7 rad (7e-6 degrees), the reference's six
// binary32 roundings of the two atan2f operands by < 1.1e-5, atan2f itself (<= 1 ulp at <= 2.1 rad) by 1.4e-5, the
// conversion to degrees (two roundings, a binary32 pi that is divided by here as well) by 1.6e-5: < 5e-5 degrees.  Twice that.
constexpr double kBandAbs = 1e-4;
constexpr double kDegD = 180.0 / (double)kPiF;
typedef unsigned long long u64;

// Phase timers of the kernel (100 MHz wall clock, frame 0's first wave prints them with AGX_TAIL_DEBUG=2): compiled in only
// with -DAGX_TAIL_TIMERS (make EXTRA=-DAGX_TAIL_TIMERS): a clock read costs about what a dozen instructions cost, and
// try_expand_one would read it four times
#ifdef AGX_TAIL_TIMERS
#define AGX_TT(...) __VA_ARGS__
#else
#define AGX_TT(...)
#endif


struct Ctx {
    const float *sx, *sy, *st;
    const float *gx, *gy;
    const uint16_t *gi, *gstart;
    float ox, oy, inv_cell;
    int nx, ny, n;
    u64 *memo_p, *memo_q;
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

__device__ __forceinline__ float angle_degree(float v0x, float v0y, float v1x, float v1y)  // math_util.rs:31-33
{
    return fdlibm_atan2f(v1y * v0x - v1x * v0y, v0x * v1x + v0y * v1y) * 180.0f / kPiF;
}

// angle_degree within 0.04 degrees: the host tail's first level (host_tail.cpp, LazyAngle::set -- the same binary32 operations, so the
// bound its tests establish holds here: tests/test_abi_cpu.py), NaN where only the exact expression will do (zero / non-finite
// operands, the sign-of-zero cases).  The comparisons below decide from it when they are farther than kCoarseBand per angle from
// their threshold -- nearly always -- and evaluate angle_degree itself otherwise: the decisions are the exact expression's.
constexpr float kCoarseBand = 0.1f;
__device__ __forceinline__ float angle_coarse(float v0x, float v0y, float v1x, float v1y)
{
    const float yf = v1y * v0x - v1x * v0y, xf = v0x * v1x + v0y * v1y;
    const float ya = fabsf(yf), xa = fabsf(xf);
    const float mx = xa > ya ? xa : ya, mn = xa > ya ? ya : xa;
    if (!(mx > 0.0f && mx < 3.0e38f && yf != 0.0f)) return __builtin_nanf("");
    const float z = mn / mx, z2 = z * z;
    float a = z * (0.9953585f + z2 * (-0.2886936f + z2 * 0.07934251f));
    if (ya > xa) a = 1.5707964f - a;
    if (xf < 0.0f) a = 3.1415927f - a;
    if (yf < 0.0f) a = -a;
    return a * 57.29578f;
}
// fabsf(p - q) > 10 (saddle.rs:59-61) from the coarse values: 1 yes, 0 no, -1 the exact angles must say (also for a NaN marker)
__device__ __forceinline__ int differ10_coarse(float pc, float qc)
{
    const float d = fabsf(pc - qc);
    return d > 10.0f + 2.0f * kCoarseBand ? 1 : (d < 10.0f - 2.0f * kCoarseBand ? 0 : -1);
}

// saddle.rs:26-38 "filter white block" for (s0, s1): 1 passes, 0 fails, 2 too close to a threshold to say here.
// Nearly every angle is degrees away from 60 and 120: those are decided from the reference's expression evaluated with
// this device's sincosf (<= 4 ulp, the OpenCL bound: the angle within 2e-4 degrees of the reference's -- the argument of
// kBandAbs with a 4-ulp direction on one side); only within 1e-3 degrees of a threshold is the binary64 evaluation made.
__device__ __forceinline__ int white_block(float s0_theta, float v02x, float v02y)
{
    // s1 == s0 (try_expand_one pairs the same saddle with itself when the candidate lists overlap): both atan2f operands are
    // zeros whatever cosf / sinf return, the angle is 0 or 180
    if (v02x == 0.0f && v02y == 0.0f) return 0;
    const float th = s0_theta / 180.0f * kPiF;
    {
        float sf, cf;
        sincosf(th, &sf, &cf);
        {  // degrees away from both thresholds: from the coarse angle (this device's sincosf adds 2e-5 degrees to its 0.04)
            const float ac = fabsf(angle_coarse(v02x, v02y, cf, sf));
            {  // (tests only, TailArgs::debug_band: a wide band so that a known share of frames takes the hand-back path; the
               // value lies in the frame's shared words -- sh[4] -- so that no register carries it through the search)
                extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
                const float debug_band = *reinterpret_cast<const float *>(lds + OFF_SHARED + 4 * 4);
                if (debug_band > 0.0f && (fabsf(ac - 60.0f) <= debug_band || fabsf(ac - 120.0f) <= debug_band)) return 2;
            }
            if (ac < 60.0f - kCoarseBand || ac > 120.0f + kCoarseBand) return 0;
            if (ac > 60.0f + kCoarseBand && ac < 120.0f - kCoarseBand) return 1;
        }
        const float yf = sf * v02x - cf * v02y, xf = v02x * cf + v02y * sf;
        const float af = fabsf(fdlibm_atan2f(yf, xf) * 180.0f / kPiF);
        if (af < 60.0f - 1e-3f || af > 120.0f + 1e-3f) return 0;
        if (af > 60.0f + 1e-3f && af < 120.0f - 1e-3f) return 1;
    }
    double sd, cd;
    sincos((double)th, &sd, &cd);
    const double y = sd * (double)v02x - cd * (double)v02y, x = (double)v02x * cd + (double)v02y * sd;
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
    {  // :55-61, a0 against a2
        const int r = differ10_coarse(angle_coarse(v01x, v01y, v12x, v12y), angle_coarse(v23x, v23y, v30x, v30y));
        if (r > 0 || (r < 0 && fabsf(angle_degree(v01x, v01y, v12x, v12y) - angle_degree(v23x, v23y, v30x, v30y)) > 10.0f)) return false;
    }
    {  // a1 against a3
        const int r = differ10_coarse(angle_coarse(v12x, v12y, v23x, v23y), angle_coarse(v30x, v30y, v01x, v01y));
        if (r > 0 || (r < 0 && fabsf(angle_degree(v12x, v12y, v23x, v23y) - angle_degree(v30x, v30y, v01x, v01y)) > 10.0f)) return false;
    }
    return true;
}
// is_valid_quad: 1 valid, 0 not, 2 everything but the white-block test passes and that one is undecided here
__device__ int valid_quad(const Ctx &c, int i0, int i1, int i2, int i3)
{
    if (!quad_rest(c, i0, i1, i2, i3)) return 0;
    return white_block(c.st[i0], c.sx[i2] - c.sx[i0], c.sy[i2] - c.sy[i0]);
}


// ---- wave helpers ---------------------------------------------------------------------------------------------------

// LDS written by some lanes of this wave, read by others: order the accesses (the waves of a frame run apart between the
// workgroup barriers, so this is not __syncthreads)
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
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
// minimum over the lane's row of 16, in every lane of the row (DPP: no LDS crossbar)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t row_min_u(uint32_t v)
{
    uint32_t t = dpp_u<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    v = t < v ? t : v;
    t = dpp_u<0x4E>(v);           // quad_perm [2, 3, 0, 1]
    v = t < v ? t : v;
    t = dpp_u<0x141>(v);          // row_half_mirror
    v = t < v ? t : v;
    t = dpp_u<0x140>(v);          // row_mirror
    return t < v ? t : v;
}
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask);
    return (u64)hi << 32 | lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src)
{
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src);
    return (u64)hi << 32 | lo;
}

// ---- a board, grown by the whole wave (its slot in LDS) -----------------------------------------------------------

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
__device__ __forceinline__ int q_at(u64 q, int j) { return (int)((q >> (16 * j)) & 0xffffull); }
__device__ __forceinline__ u64 q_make(int a, int b, int c, int d) { return (u64)a | (u64)b << 16 | (u64)c << 32 | (u64)d << 48; }

// try_expand_one (board.rs:153-176) for the quad qs (rotated as try_expand passes it), by the wave.  The four queries of
// find_closest_potential_saddle_idxs (:177-233) -- next to s0 and s1 along s0 -> s1, next to s3 and s2 along s3 -> s2 -- run on
// 16 lanes each: the three nearest saddles among those within the radius (which is what tree.nearest(.., 3, ..) filtered
// by `dist_sq <= radius_sq` leaves: a saddle within the radius that is among the three nearest of all is among the three
// nearest of those within the radius, and the other way round), found in the grid cells the radius reaches.  Then the
// reference's four nested loops as one combination per lane; the first valid one in loop order is the result.
__device__ bool expand_one_w(const Ctx &c, const uint8_t *slot, u64 qs, u64 &out, int lane, uint32_t &status, unsigned long long *ek)
{
    AGX_TT(unsigned long long e_last = wall_clock64();)
#define EK(i) AGX_TT(do { const unsigned long long t_now = wall_clock64(); ek[i] += t_now - e_last; e_last = t_now; } while (0))
    const int g = lane >> 4, l = lane & 15;
    const int ia = g < 2 ? q_at(qs, 0) : q_at(qs, 3), ib = g < 2 ? q_at(qs, 1) : q_at(qs, 2);  // the pair (first, second)
    const int anchor = (g & 1) ? ib : ia;                                                      // whose neighbour is looked for
    // what the query finds before the board's own test (radius and orientation, :207-216), from the memo or by the search:
    // raw = idx0 | idx1 << 10 | idx2 << 20 | count << 30 (an index has 10 bits: TN = 1024)
    const uint32_t pkey = 0x80000000u | (uint32_t)ia << 11 | (uint32_t)ib << 1 | (uint32_t)(g & 1);
    u64 *pslot = c.memo_p + ((pkey * 2654435761u) >> 21);
    const u64 pe = __hip_atomic_load(pslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t raw = (uint32_t)pe;
    AGX_TT(ek[7] += (unsigned long long)__popcll(__ballot((uint32_t)(pe >> 32) != pkey)) / 16;)  // (queries of this call that miss the memo)
    if ((uint32_t)(pe >> 32) != pkey) {  // (the 16 lanes of a query alike)
        const float ax = c.sx[ia], ay = c.sy[ia], bx = c.sx[ib], by = c.sy[ib];
        const float ratio0 = 1.0f + 0.3f;
        const float ex = ax - bx, ey = ay - by;
        const float radius_sq = 0.5f * (ex * ex + ey * ey);
        const float v10x = bx - ax, v10y = by - ay;
        const float qx = c.sx[anchor] + v10x * ratio0, qy = c.sy[anchor] + v10y * ratio0;
        u64 k0 = ~0ull, k1 = ~0ull, k2 = ~0ull;
        const float r = __builtin_amdgcn_sqrtf(radius_sq) * 1.0001f + 1e-3f;  // (only bounds the cells looked at: the hardware's square root, 1 ulp, and a margin)
        if (!(r < 3e38f)) {  // (not on image coordinates) everything
            for (int t = l; t < c.n; t += 16) top3_insert(dist_key(qx, qy, c.gx[t], c.gy[t], c.gi[t]), k0, k1, k2);
        } else {
            // the cell rows the radius reaches, four at a time: four lanes per row, each takes every fourth saddle of the row's
            // run (a run holds a handful) -- the rows' bounds in one LDS round trip, their saddles in one or two more
            const int xa = cell_x(c, qx - r), xb = cell_x(c, qx + r), ya = cell_y(c, qy - r), yb = cell_y(c, qy + r);
            for (int y0 = ya; y0 <= yb; y0 += 4) {
                const int y = y0 + (l >> 2);
                if (y <= yb) {
                    const int t0 = c.gstart[y * c.nx + xa], t1 = c.gstart[y * c.nx + xb + 1];
                    for (int t = t0 + (l & 3); t < t1; t += 4) top3_insert(dist_key(qx, qy, c.gx[t], c.gy[t], c.gi[t]), k0, k1, k2);
                }
            }
        }
        // the three smallest keys of the 16 lanes ((distance, index): the distance first, then the index among the nearest) ...
        u64 top0, top1, top2;
        uint32_t hi0 = (uint32_t)(k0 >> 32), mh = row_min_u(hi0);
        uint32_t ml = row_min_u(hi0 == mh ? (uint32_t)k0 : 0xffffffffu);
        top0 = (u64)mh << 32 | ml;
        if (k0 == top0 && top0 != ~0ull) { k0 = k1; k1 = k2; k2 = ~0ull; }
        hi0 = (uint32_t)(k0 >> 32); mh = row_min_u(hi0);
        ml = row_min_u(hi0 == mh ? (uint32_t)k0 : 0xffffffffu);
        top1 = (u64)mh << 32 | ml;
        if (k0 == top1 && top1 != ~0ull) { k0 = k1; k1 = k2; k2 = ~0ull; }
        hi0 = (uint32_t)(k0 >> 32); mh = row_min_u(hi0);
        ml = row_min_u(hi0 == mh ? (uint32_t)k0 : 0xffffffffu);
        top2 = (u64)mh << 32 | ml;
        // ... within the radius and of the anchor's orientation: lanes 0 .. 2 of the query take one each
        const u64 mine = l == 0 ? top0 : (l == 1 ? top1 : top2);
        const bool have = l < 3 && mine != ~0ull;
        const int idx = have ? (int)(uint32_t)mine : 0;
        const bool ok = have && __uint_as_float((uint32_t)(mine >> 32)) <= radius_sq && theta_dist(c.st[anchor], c.st[idx]) < 5.0f;
        const uint32_t m3 = (uint32_t)(__ballot(ok) >> (16 * g)) & 7u;  // this query's three
        uint32_t rcnt = 0;
        raw = 0;
        if (m3 & 1u) { raw |= ((uint32_t)top0 & 0x3ffu) << (10 * rcnt); ++rcnt; }
        if (m3 & 2u) { raw |= ((uint32_t)top1 & 0x3ffu) << (10 * rcnt); ++rcnt; }
        if (m3 & 4u) { raw |= ((uint32_t)top2 & 0x3ffu) << (10 * rcnt); ++rcnt; }
        raw |= rcnt << 30;
        if (l == 0) __hip_atomic_store(pslot, (u64)pkey << 32 | raw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    EK(0);
    if (__ballot((raw >> 30) == 0)) {  // a query that finds nothing leaves an empty list, and try_expand_one's loops are empty with it
        AGX_TT(ek[5] += 1;)
        return false;
    }
    // The four lists to every lane (they sit in lanes 0, 16, 32, 48), then the reference's four nested loops (:160-174) as one
    // combination per lane, numbered in loop order over the lists BEFORE the board's own test (:207 active_idxs): a combination
    // counts if its four saddles are still unused by this board -- dropping the used ones from the lists first, as the
    // reference does, leaves the same combinations in the same order.  The used-bits and the memo are read side by side.
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)raw, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)raw, 16);
    const uint32_t r3 = (uint32_t)__builtin_amdgcn_readlane((int)raw, 32), r2 = (uint32_t)__builtin_amdgcn_readlane((int)raw, 48);
    const int n0 = (int)(r0 >> 30), n1 = (int)(r1 >> 30), n2 = (int)(r2 >> 30), n3 = (int)(r3 >> 30);
    EK(2);
    AGX_TT(ek[5] += 1; ek[6] += 1;)
    for (int pass = 0; pass < 2; ++pass) {
        const int L = lane + 64 * pass;  // ((i0 * 3 + i1) * 3 + i2) * 3 + i3
        const int j0 = L / 27, j1 = (L / 9) % 3, j2 = (L / 3) % 3, j3 = L % 3;
        const bool in = L < 81 && j0 < n0 && j1 < n1 && j2 < n2 && j3 < n3;
        const int a = (int)((r0 >> (10 * (j0 < 3 ? j0 : 0))) & 0x3ffu), b = (int)((r1 >> (10 * j1)) & 0x3ffu);
        const int cc = (int)((r2 >> (10 * j2)) & 0x3ffu), d = (int)((r3 >> (10 * j3)) & 0x3ffu);
        int v = 0;
        if (in) {
            const u64 qkey = 1ull << 63 | (u64)a | (u64)b << 10 | (u64)cc << 20 | (u64)d << 30;
            u64 *qslot = c.memo_q + (((uint32_t)qkey * 2654435761u ^ (uint32_t)(qkey >> 20) * 40503u) >> 21);
            const u64 qe = __hip_atomic_load(qslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool unused = slot_active(slot, a) && slot_active(slot, b) && slot_active(slot, cc) && slot_active(slot, d);
            if (unused) {  // is_valid_quad of these four: from the memo, or evaluated and kept
                if ((qe & ~(3ull << 40)) == qkey) {
                    v = (int)((qe >> 40) & 3ull);
                } else {
                    v = valid_quad(c, a, b, cc, d);
                    __hip_atomic_store(qslot, qkey | (u64)v << 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        const u64 mv = __ballot(v == 1), mu = __ballot(v == 2);
        const int first = mv ? __ffsll((long long)mv) - 1 : 64;
        if (mu & (first == 64 ? ~0ull : ((1ull << first) - 1ull))) status |= TAIL_UNCERTAIN;  // an undecided one before it
        if (mv) {
            out = shfl_u64(q_make(a, b, cc, d), first);
            EK(3);
            return true;
        }
        if (n0 < 3) break;  // (combinations 64 .. 80 have i0 = 2)
    }
    EK(3);
    return false;
}

// minimum over the lane's quad of four, in every lane of the four
__device__ __forceinline__ uint32_t quad_min_u(uint32_t v)
{
    uint32_t t = dpp_u<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    v = t < v ? t : v;
    t = dpp_u<0x4E>(v);           // quad_perm [2, 3, 0, 1]
    return t < v ? t : v;
}

// The SIXTEEN queries of a cell's four directions at once, four lanes each (a lane per cell row): which directions cannot be
// expanded whatever the board has used -- one of the direction's four lists is empty before the board's own test, and
// try_expand_one's loops are empty with it (:160-174).  Most directions of most cells are such (a board grown from a quad that
// is no tag's: all four): try_expand then need not ask.  Asked of a board's first cell only -- most boards end there; for the
// cells found later it costs more than it saves (measured).  The lists go to the memo: a direction that can be expanded
// finds them there.  -> bit d: direction d cannot be expanded
__device__ uint32_t probe4_w(const Ctx &c, u64 quad, int lane)
{
    const int q16 = lane >> 2, sub = lane & 3, dir = q16 >> 2, g = q16 & 3;
    const u64 qs = dir ? (quad >> (16 * dir) | quad << (64 - 16 * dir)) : quad;  // qs[j] = quad[(j + dir) & 3]
    const int ia = g < 2 ? q_at(qs, 0) : q_at(qs, 3), ib = g < 2 ? q_at(qs, 1) : q_at(qs, 2);
    const int anchor = (g & 1) ? ib : ia;
    const uint32_t pkey = 0x80000000u | (uint32_t)ia << 11 | (uint32_t)ib << 1 | (uint32_t)(g & 1);
    u64 *pslot = c.memo_p + ((pkey * 2654435761u) >> 21);
    const u64 pe = __hip_atomic_load(pslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t raw = (uint32_t)pe;
    if ((uint32_t)(pe >> 32) != pkey) {  // (the four lanes of a query alike)
        const float ax = c.sx[ia], ay = c.sy[ia], bx = c.sx[ib], by = c.sy[ib];
        const float ratio0 = 1.0f + 0.3f;
        const float ex = ax - bx, ey = ay - by;
        const float radius_sq = 0.5f * (ex * ex + ey * ey);
        const float v10x = bx - ax, v10y = by - ay;
        const float qx = c.sx[anchor] + v10x * ratio0, qy = c.sy[anchor] + v10y * ratio0;
        u64 k0 = ~0ull, k1 = ~0ull, k2 = ~0ull;
        const float r = __builtin_amdgcn_sqrtf(radius_sq) * 1.0001f + 1e-3f;
        if (!(r < 3e38f)) {
            for (int t = sub; t < c.n; t += 4) top3_insert(dist_key(qx, qy, c.gx[t], c.gy[t], c.gi[t]), k0, k1, k2);
        } else {
            const int xa = cell_x(c, qx - r), xb = cell_x(c, qx + r), ya = cell_y(c, qy - r), yb = cell_y(c, qy + r);
            for (int y = ya + sub; y <= yb; y += 4) {
                const int t0 = c.gstart[y * c.nx + xa], t1 = c.gstart[y * c.nx + xb + 1];
                for (int t = t0; t < t1; ++t) top3_insert(dist_key(qx, qy, c.gx[t], c.gy[t], c.gi[t]), k0, k1, k2);
            }
        }
        u64 top0, top1, top2;
        uint32_t hi0 = (uint32_t)(k0 >> 32), mh = quad_min_u(hi0);
        uint32_t ml = quad_min_u(hi0 == mh ? (uint32_t)k0 : 0xffffffffu);
        top0 = (u64)mh << 32 | ml;
        if (k0 == top0 && top0 != ~0ull) { k0 = k1; k1 = k2; k2 = ~0ull; }
        hi0 = (uint32_t)(k0 >> 32); mh = quad_min_u(hi0);
        ml = quad_min_u(hi0 == mh ? (uint32_t)k0 : 0xffffffffu);
        top1 = (u64)mh << 32 | ml;
        if (k0 == top1 && top1 != ~0ull) { k0 = k1; k1 = k2; k2 = ~0ull; }
        hi0 = (uint32_t)(k0 >> 32); mh = quad_min_u(hi0);
        ml = quad_min_u(hi0 == mh ? (uint32_t)k0 : 0xffffffffu);
        top2 = (u64)mh << 32 | ml;
        const u64 mine = sub == 0 ? top0 : (sub == 1 ? top1 : top2);
        const bool have = sub < 3 && mine != ~0ull;
        const int idx = have ? (int)(uint32_t)mine : 0;
        const bool ok = have && __uint_as_float((uint32_t)(mine >> 32)) <= radius_sq && theta_dist(c.st[anchor], c.st[idx]) < 5.0f;
        const uint32_t m3 = (uint32_t)(__ballot(ok) >> (4 * q16)) & 7u;  // this query's three
        uint32_t rcnt = 0;
        raw = 0;
        if (m3 & 1u) { raw |= ((uint32_t)top0 & 0x3ffu) << (10 * rcnt); ++rcnt; }
        if (m3 & 2u) { raw |= ((uint32_t)top1 & 0x3ffu) << (10 * rcnt); ++rcnt; }
        if (m3 & 4u) { raw |= ((uint32_t)top2 & 0x3ffu) << (10 * rcnt); ++rcnt; }
        raw |= rcnt << 30;
        if (sub == 0) __hip_atomic_store(pslot, (u64)pkey << 32 | raw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const u64 empty = __ballot((raw >> 30) == 0);
    return ((empty & 0xffffull) ? 1u : 0u) | ((empty >> 16 & 0xffffull) ? 2u : 0u) | ((empty >> 32 & 0xffffull) ? 4u : 0u) | ((empty >> 48) ? 8u : 0u);
}

// Board::new (board.rs:26-48): the board grown from a seed quad; returns its score, the cells stay in the slot.
// try_expand's recursion (:114-152) is a stack of (cell, next direction) walked by all lanes alike: the top of the stack lives
// in registers, lane 0 writes the slot, one wave-level ordering point per step.  A grid byte = the cell's number, bit 7 = found.
__device__ int build_board_w(const Ctx &c, uint8_t *slot, u64 seed, int lane, int &n_cells_out, uint32_t &status, unsigned long long *ek)
{
    {
        uint32_t *g = reinterpret_cast<uint32_t *>(slot + SL_GRID);
        for (int i = lane; i < (SL_ACTIVE - SL_GRID) / 4; i += 64) g[i] = 0xffffffffu;
        if (lane < TN / 32) {  // active_idxs: everything but the seed quad's saddles 1 .. 3 (:35-37) -- a word per lane, no read-modify-write
            uint32_t word = 0xffffffffu;
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const int idx = q_at(seed, j);
                if ((idx >> 5) == lane) word &= ~(1u << (idx & 31));
            }
            reinterpret_cast<uint32_t *>(slot + SL_ACTIVE)[lane] = word;
        }
    }
    wsync();
    uint8_t *grid = slot + SL_GRID, *found = slot + SL_FOUND, *stack = slot + SL_STACK;
    int8_t *xy = reinterpret_cast<int8_t *>(slot + SL_XY);
    if (lane == 0) {
        slot_set_quad(slot, 0, seed);
        xy[0] = 0;
        xy[1] = 0;
        found[0] = 1;
        grid[BGR * BGN + BGR] = 0x80;
    }
    wsync();
    int n_cells = 1, score = 1, sp = 1;
    int cur = 0, cur_i = 0, cur_x = 0, cur_y = 0;  // the top of the stack: cell, next direction, the cell's coordinates and quad
    u64 cur_quad = seed;
    uint32_t cur_dead = probe4_w(c, seed, lane);  // its directions that cannot be expanded (probe4_w)
    for (;;) {
        if (cur_i == 4) {  // this cell is done: back to the one it was reached from
            if (--sp == 0) break;
            cur = stack[2 * (sp - 1)];
            cur_i = stack[2 * (sp - 1) + 1] & 15;
            cur_dead = stack[2 * (sp - 1) + 1] >> 4;
            cur_x = xy[2 * cur];
            cur_y = xy[2 * cur + 1];
            cur_quad = slot_quad(slot, cur);
            continue;
        }
        const int i = cur_i++;
        const int nx = cur_x + (i == 0 ? 1 : (i == 2 ? -1 : 0)), ny = cur_y + (i == 1 ? -1 : (i == 3 ? 1 : 0));
        if (nx < -BGR || nx > BGR || ny < -BGR || ny > BGR) {
            status |= TAIL_CAPACITY;
            break;
        }
        const int gpos = (ny + BGR) * BGN + (nx + BGR);
        const int e = grid[gpos];
        if (e != 0xff && (e & 0x80)) continue;  // :132-136 already found
        const u64 qs = i ? (cur_quad >> (16 * i) | cur_quad << (64 - 16 * i)) : cur_quad;  // qs[j] = quad[(j + i) & 3]
        u64 nq = 0;
        const bool ok = ((cur_dead >> i) & 1u) ? false : expand_one_w(c, slot, qs, nq, lane, status, ek);
        int at = e & 0x7f;
        if (e == 0xff) {
            if (n_cells == BCELLS) {
                status |= TAIL_CAPACITY;
                break;
            }
            at = n_cells++;
        }
        const u64 v = i ? (nq << (16 * i) | nq >> (64 - 16 * i)) : nq;  // v[(j + i) & 3] = nq[j]
        if (lane == 0) {
            grid[gpos] = (uint8_t)(at | (ok ? 0x80 : 0));
            if (e == 0xff) {
                xy[2 * at] = (int8_t)nx;
                xy[2 * at + 1] = (int8_t)ny;
            }
            found[at] = ok ? 1 : 0;
            slot_set_quad(slot, at, ok ? v : 0ull);
            if (ok) {
                stack[2 * (sp - 1)] = (uint8_t)cur;  // where to come back to (depth <= found cells <= BCELLS)
                stack[2 * (sp - 1) + 1] = (uint8_t)(cur_i | cur_dead << 4);
            }
        }
        if (ok && lane < 4) {  // the new cell's four saddles are used (:140-142): a lane each, an atomic each (two may share a word)
            const int idx = q_at(v, lane);
            atomicAnd(reinterpret_cast<uint32_t *>(slot + SL_ACTIVE) + (idx >> 5), ~(1u << (idx & 31)));
        }
        if (ok) {  // try_expand(&new_board_idx), :146
            ++score;
            ++sp;
            cur = at;
            cur_i = 0;
            cur_x = nx;
            cur_y = ny;
            cur_quad = v;
            cur_dead = 0;  // (asked only of the seed's cell: a found cell's neighbours are mostly found or expandable, measured)
        }
        wsync();
    }
    n_cells_out = n_cells;
    return score;
}

// ---- decode (detector.rs:42-169, 448-476; image_util.rs:39-70) -----------------------------------------------------

// decode_positions + bit_code (detector.rs:42-122) of one quad: false = None
// reductions over the lane's EIGHT (half a row of 16), in every lane of the eight
template <typename F>
__device__ __forceinline__ uint32_t half_reduce_u(uint32_t v, F f)
{
    v = f(v, dpp_u<0xB1>(v));   // quad_perm [1, 0, 3, 2]
    v = f(v, dpp_u<0x4E>(v));   // quad_perm [2, 3, 0, 1]
    return f(v, dpp_u<0x141>(v));  // row_half_mirror
}
// decode_positions + bit_code (detector.rs:42-122) of one quad by EIGHT lanes (sub = 0 .. 7 takes every eighth sample of the up to 40:
// the samples' loads side by side instead of one after the other): true + the bits in every lane of the eight; false = None
__device__ bool quad_bits8(const TailArgs &a, const uint8_t *luma, const float q[8], int sub, u64 &bits_out)
{
    const uint32_t w = (uint32_t)a.W, h = (uint32_t)a.H;
    bool outside = false;
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = f32_as_u32(round_half_away(q[2 * i])), y = f32_as_u32(round_half_away(q[2 * i + 1]));
        outside = outside || x >= w || y >= h;
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
    // the samples in the reference's order (x outer, y inner): number sub, sub + 8, ...  (a sample outside the image ends the
    // reference's loop with None: here it is remembered and the coordinates are clamped)
    const int nb = a.edge * a.edge, edge = a.edge, border = a.border, pitch = a.luma_row_stride;
    int lo = 255, hi = 0;
    int vals[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int n = sub + 8 * k;
        const bool valid = n < nb;
        const int nn = valid ? n : 0;
        const float fx = (float)(border + nn / edge), fy = (float)(border + nn % edge);
        const float px = aff[0] * fx + aff[1] * fy + aff[2] * 1.0f;
        const float py = aff[3] * fx + aff[4] * fy + aff[5] * 1.0f;
        uint32_t ix = f32_as_u32(round_half_away(px)), iy = f32_as_u32(round_half_away(py));
        outside = outside || (valid && (ix >= w || iy >= h));
        ix = ix < w ? ix : w - 1;
        iy = iy < h ? iy : h - 1;
        const int b = luma[(size_t)iy * (size_t)pitch + ix];
        vals[k] = valid ? b : -1;
        lo = valid && b < lo ? b : lo;
        hi = valid && b > hi ? b : hi;
    }
    lo = (int)half_reduce_u((uint32_t)lo, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
    hi = (int)half_reduce_u((uint32_t)hi, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    outside = half_reduce_u(outside ? 1u : 0u, [](uint32_t x, uint32_t y) { return x | y; }) != 0;
    if (outside || hi - lo < 50) return false;  // (the eight alike)
    const int mid = (int)(uint8_t)f32_as_u32(round_half_away(((float)lo + (float)hi) / 2.0f));
    u64 bits = 0;
    uint32_t invalid = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (vals[k] >= 0) {  // the first sample is the most significant bit
            const int b = vals[k], d = mid - b;
            if ((d < 0 ? -d : d) < 10) ++invalid;
            if (b > mid) bits |= 1ull << (nb - 1 - (sub + 8 * k));
        }
    invalid = half_reduce_u(invalid, [](uint32_t x, uint32_t y) { return x + y; });
    if (invalid > 3) return false;
    bits_out = (u64)half_reduce_u((uint32_t)(bits >> 32), [](uint32_t x, uint32_t y) { return x | y; }) << 32 |
               half_reduce_u((uint32_t)bits, [](uint32_t x, uint32_t y) { return x | y; });
    return true;
}

// OR over the lane's row of 16, in every lane of the row
__device__ __forceinline__ uint32_t row_or_u(uint32_t v)
{
    v |= dpp_u<0xB1>(v);
    v |= dpp_u<0x4E>(v);
    v |= dpp_u<0x141>(v);
    return v | dpp_u<0x140>(v);
}
// rotate_bits (detector.rs:124-140) by a row of 16 lanes: output bit `count` is input bit r + c * edge with
// count = (edge - 1 - r) * edge + c; every lane sets the output bits l, l + 16, l + 32
__device__ __forceinline__ u64 rotate_bits_row(u64 bits, int edge, int l)
{
    u64 out = 0;
    for (int count = l; count < edge * edge; count += 16) {
        const int r = edge - 1 - count / edge, cc = count % edge;
        out |= ((bits >> (r + cc * edge)) & 1ull) << count;
    }
    return (u64)row_or_u((uint32_t)(out >> 32)) << 32 | row_or_u((uint32_t)out);
}

// best_tag (detector.rs:142-169) by a row of 16 lanes (four quads per wave at a time), the family's codes in LDS.  The
// reference scans the family once per rotation until one matches; here ONE scan serves the four rotations (every lane takes
// every 16th code; per rotation the first code with the smallest distance = the smallest (distance, index) pair), then the
// rotations are looked at in the reference's order -- the same (index, rotation).
__device__ bool best_tag_row(const TailArgs &a, const u64 *codes, u64 bits, int l, int &idx, int &rot)
{
    const int n_codes = a.n_codes, edge = a.edge;
    const uint32_t hamming = (uint32_t)a.hamming;
    const u64 b0 = bits, b1 = rotate_bits_row(b0, edge, l), b2 = rotate_bits_row(b1, edge, l), b3 = rotate_bits_row(b2, edge, l);
    uint32_t m0 = 0xffffffffu, m1 = 0xffffffffu, m2 = 0xffffffffu, m3 = 0xffffffffu;
#pragma unroll 2
    for (int i = l; i < n_codes; i += 16) {
        const u64 code = codes[i];
        const uint32_t k0 = (uint32_t)__popcll(code ^ b0) << 16 | (uint32_t)i, k1 = (uint32_t)__popcll(code ^ b1) << 16 | (uint32_t)i;
        const uint32_t k2 = (uint32_t)__popcll(code ^ b2) << 16 | (uint32_t)i, k3 = (uint32_t)__popcll(code ^ b3) << 16 | (uint32_t)i;
        m0 = k0 < m0 ? k0 : m0;
        m1 = k1 < m1 ? k1 : m1;
        m2 = k2 < m2 ? k2 : m2;
        m3 = k3 < m3 ? k3 : m3;
    }
    m0 = row_min_u(m0);
    m1 = row_min_u(m1);
    m2 = row_min_u(m2);
    m3 = row_min_u(m3);
    const uint32_t ms[4] = {m0, m1, m2, m3};
#pragma unroll
    for (int rotated = 0; rotated < 4; ++rotated)
        if ((ms[rotated] >> 16) < hamming) {
            idx = (int)(ms[rotated] & 0xffffu);
            rot = rotated;
            return true;
        }
    return false;
}

// ---- init_quads (detector.rs:543-586) for the seed s0, by one wave: the candidate quads into the wave's list, in the
// reference's order; returns how many
__device__ int init_quads_w(const Ctx &c, uint8_t *wv, int s0, int lane, uint32_t &status, unsigned long long *tk)
{
    AGX_TT(unsigned long long t_last = wall_clock64();)
#define TKS(i) AGX_TT(do { const unsigned long long t_now = wall_clock64(); tk[i] += t_now - t_last; t_last = t_now; } while (0))
    const u64 below = (1ull << lane) - 1ull;
    u64 *cand = reinterpret_cast<u64 *>(wv + WV_CAND);
    uint16_t *pairs = reinterpret_cast<uint16_t *>(wv + WV_PAIRS);
    uint16_t *same = reinterpret_cast<uint16_t *>(wv + WV_SMALL), *diff = same + 64, *s1ok = same + 128;
    float *a0s = reinterpret_cast<float *>(wv + WV_SMALL + 384), *a2s = a0s + 64, *a3tab = reinterpret_cast<float *>(wv + WV_A3);
    const int n = c.n;
    const float *sx = c.sx, *sy = c.sy, *st = c.st;
    const float s0x = sx[s0], s0y = sy[s0], s0t = st[s0];
    int nc = 0;
    // 50 nearest: the distance keys, sorted.  Of more than 128 saddles only those are sorted that can be among the 50: the 50th
    // smallest of the 64 lanes' minima (each lane's own keys: every 64th saddle) is one of 64 keys of the set, so the set's 50th
    // smallest is not above it -- usually 50 to 80 keys are left.
    int P = 64;
    u64 *keys = cand;
    wsync();  // (the previous seed's candidates are done with)
    bool pruned = false;
    if (n > 128) {
        u64 v = ~0ull;
        for (int i = lane; i < n; i += 64) {
            const u64 k = dist_key(s0x, s0y, sx[i], sy[i], (uint32_t)i);
            v = k < v ? k : v;
        }
        for (int k2 = 2; k2 <= 64; k2 <<= 1)  // the 64 minima sorted across the lanes (ascending with the lane)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                const u64 o = shfl_xor_u64(v, j);
                const bool keep_min = ((lane & j) == 0) == ((lane & k2) == 0);
                v = (o < v) == keep_min ? o : v;
            }
        const u64 bound = shfl_u64(v, 49);
        int cnt = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const u64 k = i < n ? dist_key(s0x, s0y, sx[i], sy[i], (uint32_t)i) : ~0ull;
            const bool keep = i < n && k <= bound;
            const u64 mk = __ballot(keep);
            const int at = cnt + __popcll(mk & below);
            if (keep && at < 128) keys[at] = k;
            cnt += __popcll(mk);
        }
        if (cnt <= 128) {
            pruned = true;
            P = cnt <= 64 ? 64 : 128;
            for (int i = cnt + lane; i < P; i += 64) keys[i] = ~0ull;
        }
    }
    if (!pruned) {
        while (P < n) P <<= 1;
        for (int i = lane; i < P; i += 64) keys[i] = i < n ? dist_key(s0x, s0y, sx[i], sy[i], (uint32_t)i) : ~0ull;
    }
    wsync();
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
            wsync();
        }
    TKS(2);
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
    wsync();  // (the keys are dead from here: cand is written below)
    int n_pairs = 0;
    if (ns > 0 && nd >= 2) {
        // the (d0, d1) combinations in the reference's order (itertools combinations(2)) that pass saddle.rs:18-21
        for (int p = 0; p < nd; ++p) {
            const bool ok = lane > p && lane < nd && !(theta_dist(st[diff[p]], st[diff[lane < nd ? lane : 0]]) > 5.0f);
            const u64 mk = __ballot(ok);
            if (ok) pairs[n_pairs + __popcll(mk & below)] = (uint16_t)(p | lane << 8);
            n_pairs += __popcll(mk);
        }
        // the white-block test depends on (s0, s1) only: once per s1 (0 fails, 1 passes, 2 undecided here)
        if (lane < ns) s1ok[lane] = (uint16_t)white_block(s0t, sx[same[lane]] - s0x, sy[same[lane]] - s0y);
        wsync();
        // a3 = angle(v30, v01) (saddle.rs:59) depends on (d0, d1) only: once per pair (its coarse value; the exact one on demand)
        for (int p = lane; p < n_pairs && p < TA3; p += 64) {
            const int d0 = diff[pairs[p] & 0xff], d1 = diff[pairs[p] >> 8];
            a3tab[p] = angle_coarse(s0x - sx[d1], s0y - sy[d1], sx[d0] - s0x, sy[d0] - s0y);
        }
    }
    wsync();
    TKS(3);
    if (n_pairs > 0) {
        for (int si = 0; si < ns; ++si) {
            const int wb = s1ok[si];
            if (wb == 0) continue;
            const int s1 = same[si];
            const float s1x = sx[s1], s1y = sy[s1];
            const float v02x = s1x - s0x, v02y = s1y - s0y;
            // a0 = angle(v01, v12) depends on (s1, d0), a2 = angle(v23, v30) on (s1, d1) (saddle.rs:56-58): once per d
            wsync();
            if (lane < nd) {
                const float dx = sx[diff[lane]], dy = sy[diff[lane]];
                a0s[lane] = angle_coarse(dx - s0x, dy - s0y, s1x - dx, s1y - dy);
                a2s[lane] = angle_coarse(dx - s1x, dy - s1y, s0x - dx, s0y - dy);
            }
            wsync();
            for (int base = 0; base < n_pairs; base += 64) {
                const int p = base + lane;
                bool ok = false;
                u64 q = 0;
                if (p < n_pairs) {
                    const int pa = pairs[p] & 0xff, pb = pairs[p] >> 8;
                    const int d0 = diff[pa], d1 = diff[pb];
                    // is_valid_quad(s0, d0, s1, d1) without the white-block test and :18-21 (the pair list), from the tables
                    {
                        const float d0x = sx[d0], d0y = sy[d0], d1x = sx[d1], d1y = sy[d1];
                        const float v01x = d0x - s0x, v01y = d0y - s0y, v03x = d1x - s0x, v03y = d1y - s0y;
                        const float v12x = s1x - d0x, v12y = s1y - d0y, v23x = d1x - s1x, v23y = d1y - s1y;
                        ok = !(cross2(v01x, v01y, v02x, v02y) * cross2(v02x, v02y, v03x, v03y) < 0.0f) &&
                             !(cross2(v01x, v01y, v12x, v12y) * cross2(v12x, v12y, v23x, v23y) < 0.0f) &&
                             !(dot2(v01x, v01y, v02x, v02y) < 0.0f || dot2(v03x, v03y, v02x, v02y) < 0.0f);
                        if (ok) {  // a0 against a2 (:55-61): the tables' coarse values, the exact angles where those are too close to say
                            const float v30x = s0x - d1x, v30y = s0y - d1y;
                            int r = differ10_coarse(a0s[pa], a2s[pb]);
                            ok = !(r > 0 || (r < 0 && fabsf(angle_degree(v01x, v01y, v12x, v12y) - angle_degree(v23x, v23y, v30x, v30y)) > 10.0f));
                            if (ok) {  // a1 against a3
                                r = differ10_coarse(angle_coarse(v12x, v12y, v23x, v23y), p < TA3 ? a3tab[p] : angle_coarse(v30x, v30y, v01x, v01y));
                                ok = !(r > 0 || (r < 0 && fabsf(angle_degree(v12x, v12y, v23x, v23y) - angle_degree(v30x, v30y, v01x, v01y)) > 10.0f));
                            }
                        }
                    }
                    if (ok && wb == 2) {  // a quad hangs on the undecided test
                        status |= TAIL_UNCERTAIN;
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
            status |= TAIL_CAPACITY;
            nc = TCAND;
        }
    }
    wsync();
    TKS(4);
    return nc;
}

__global__ void __launch_bounds__(64 * TW) k_board_tail(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const u64 below = (1ull << lane) - 1ull;
    float *sx = reinterpret_cast<float *>(lds + OFF_SX), *sy = reinterpret_cast<float *>(lds + OFF_SY), *st = reinterpret_cast<float *>(lds + OFF_ST);
    float *gx = reinterpret_cast<float *>(lds + OFF_GX), *gy = reinterpret_cast<float *>(lds + OFF_GY);
    uint16_t *gi = reinterpret_cast<uint16_t *>(lds + OFF_GI), *gstart = reinterpret_cast<uint16_t *>(lds + OFF_GSTART);
    uint16_t *seeds = reinterpret_cast<uint16_t *>(lds + OFF_SEEDS);
    u64 *quads = reinterpret_cast<u64 *>(lds + OFF_QUADS);
    uint32_t *tagids = reinterpret_cast<uint32_t *>(lds + OFF_TAGIDS);
    uint32_t *used = reinterpret_cast<uint32_t *>(lds + OFF_USED);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + OFF_HIST);
    uint32_t *sh = reinterpret_cast<uint32_t *>(lds + OFF_SHARED);  // [0] status, [1] n, [2] seeds, [3] saddles removed, [4] debug band, [5 .. 7] where the chosen board is kept, [8 ..) the groups' words (two sets)
    uint8_t *wv = lds + OFF_WAVES + wave * WV_BYTES;  // this wave's own

    uint32_t status = 0;  // per lane; merged through sh[0]
    int n_tags = 0;       // (wave 0)
    unsigned long long tk[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = wall_clock64();
    const unsigned long long t_start = t_last;
    int n_cands_total = 0, n_seeds_done = 0, n_boards = 0;
    unsigned long long ek[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dk[5] = {0, 0, 0, 0, 0};
    (void)n_cands_total; (void)n_boards; (void)dk;  // (only the timers' build reads them)
#define DK(i) AGX_TT(do { const unsigned long long t_now = wall_clock64(); dk[i] += t_now - t_last; t_last = t_now; } while (0))
#define TK(i) AGX_TT(do { const unsigned long long t_now = wall_clock64(); tk[i] += t_now - t_last; t_last = t_now; } while (0))
    const FrameCounters &fc = a.ctr[f];
    int n = (int)fc.n_out;
    const int n_first = n;
    const uint32_t cflags = fc.flags;
    if ((cflags & (FLAG_CAND_OVERFLOW | FLAG_ROOT_OVERFLOW | FLAG_OUT_OVERFLOW)) || n > TN) {  // (the whole workgroup alike)
        if (tid == 0) {
            a.table[4 * f] = 0;
            a.table[4 * f + 1] = n > TN ? TAIL_CAPACITY : TAIL_CHAIN;
            a.table[4 * f + 2] = 0;
            a.table[4 * f + 3] = 0;
        }
        return;
    }
    if (tid < 8 + 6 * TW) sh[tid] = tid == 4 ? __float_as_uint(a.debug_band) : 0u;  // ([4]: white_block's debug band)
    int group_no = 0;  // (counts the groups of seeds over all rounds)
    u64 *codes = reinterpret_cast<u64 *>(lds + OFF_CODES);
    for (int i = tid; i < a.n_codes && i < TCODES; i += 64 * TW) codes[i] = a.codes[i];
    if (a.n_codes > TCODES) status |= TAIL_CAPACITY;
    {
        const float *src = a.saddles + (size_t)fc.out_offset * 5;
        for (int i = tid; i < n; i += 64 * TW) {
            const float x = src[5 * i], y = src[5 * i + 1], t = src[5 * i + 3];
            sx[i] = x;
            sy[i] = y;
            st[i] = t;
            // coordinates of an image and half an atan2 in degrees; anything else (NaN included) is not this kernel's business
            if (!(fabsf(x) < 1e6f && fabsf(y) < 1e6f && t >= -180.0f && t <= 180.0f)) status |= TAIL_CAPACITY;
        }
    }
    const uint8_t *luma = a.luma + (size_t)f * (size_t)a.luma_frame_stride;
    const uint32_t tag_cap = a.tag_cap < (uint32_t)TTAGS ? a.tag_cap : (uint32_t)TTAGS;
    __syncthreads();
    if (status) atomicOr(&sh[0], status);
    __syncthreads();
    if (sh[0]) n = 0;  // nothing is searched; the status goes out below

    for (int round = 0; round < a.max_boards && n > 0; ++round) {
        // ---- the k-NN grid over this round's saddles (every wave derives the geometry, the first one fills the cells) ----
        Ctx c;
        c.sx = sx; c.sy = sy; c.st = st; c.gx = gx; c.gy = gy; c.gi = gi; c.gstart = gstart; c.n = n;
        c.memo_p = reinterpret_cast<u64 *>(lds + OFF_MEMO_P);
        c.memo_q = reinterpret_cast<u64 *>(lds + OFF_MEMO_Q);
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
        }
        if (wave == 0) {
            const int nx = c.nx, ncell = c.nx * c.ny;
            uint32_t *cnt = reinterpret_cast<uint32_t *>(lds + OFF_MEMO_P), *fill = cnt + TGC;  // (the memo tables are cleared below)
            for (int i = lane; i < ncell; i += 64) cnt[i] = 0;
            wsync();
            for (int i = lane; i < n; i += 64) atomicAdd(&cnt[cell_y(c, sy[i]) * nx + cell_x(c, sx[i])], 1u);
            wsync();
            uint32_t local = 0;  // exclusive scan: 16 consecutive cells per lane
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
            wsync();
            for (int i = lane; i < n; i += 64) {
                const uint32_t pos = atomicAdd(&fill[cell_y(c, sy[i]) * nx + cell_x(c, sx[i])], 1u);
                gi[pos] = (uint16_t)i;
                gx[pos] = sx[i];
                gy[pos] = sy[i];
            }
            wsync();
            TK(0);
            // ---- seeds: the most populated round(theta) bin (ties: the smallest angle), in index order ---------------
            for (int i = lane; i < 364; i += 64) hist[i] = 0;
            wsync();
            for (int i = lane; i < n; i += 64) atomicAdd(&hist[(int)round_half_away(st[i]) + 180], 1u);
            wsync();
            uint32_t bestk = 0;
            for (int i = lane; i < 361; i += 64) {
                const uint32_t k = hist[i] << 16 | (uint32_t)(0xffff - i);
                bestk = k > bestk ? k : bestk;
            }
            bestk = wave_max_u(bestk);
            const int best_angle = (int)(0xffff - (bestk & 0xffffu)) - 180;
            int ns_ = 0;
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                const bool is = i < n && (int)round_half_away(st[i]) == best_angle;
                const u64 m = __ballot(is);
                if (is) seeds[ns_ + __popcll(m & below)] = (uint16_t)i;
                ns_ += __popcll(m);
            }
            if (lane == 0) sh[2] = (uint32_t)ns_;
            TK(1);
        }
        __syncthreads();
        for (int i = tid; i < 2 * TMEMO; i += 64 * TW) c.memo_p[i] = 0ull;  // both memo tables: this round's saddle numbers
        __syncthreads();
        const int n_seeds = (int)sh[2];
        const int total = n_seeds < 30 ? n_seeds : 30;  // popped from the back, at most 30 (detector.rs:613)

        // ---- try_find_best_board's loop over the seeds (:613-629), TW seeds at a time: every wave lists the candidate quads
        // of one seed, then the waves take the group's boards one by one from a common counter (the boards of a seed that
        // finds the real board cost fifty times what the others cost) ----------------------------------------------------
        uint32_t best_score = 0, win_stamp = 0;
        u64 best_quad = 0;
        bool stop = false;
        // Seeds in groups of TW: every wave lists the candidate quads of one; then the boards.  The first group's boards in two
        // steps -- the first seed alone (it usually finds the board and ends the loop: its neighbours in the list would find
        // it again, fifty times the work), then the other seven, whose lists are there already.
        for (int base = 0; base < total && !stop; base += TW, ++group_no) {
            const int gw = total - base < TW ? total - base : TW;
            // the group's words: [0 .. TW) a seed's list published (bit 31) with its length, [TW .. 2 TW) the seed's best board,
            // [2 TW .. 3 TW) its candidates handed out.  Two sets, used alternately: a wave clears its words of the other set
            // when it has built its last board of this group, behind the barrier at which that set was last read
            uint32_t *gs = sh + 8 + 3 * TW * (group_no & 1), *gs_other = sh + 8 + 3 * TW * ((group_no & 1) ^ 1);
            const int k = base + wave;
            int nc_mine = 0;
            if (wave < gw && k < total) {
                nc_mine = init_quads_w(c, wv, seeds[n_seeds - 1 - k], lane, status, tk);
                n_cands_total += nc_mine;
                ++n_seeds_done;
            }
            // the list is complete: published with its length (bit 31).  No barrier: a wave takes boards of the seeds that are
            // listed while others still list theirs -- seed by seed in the reference's order, candidates from a counter per seed
            if (lane == 0) __hip_atomic_store(&gs[wave], (uint32_t)nc_mine | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int n_steps = (base == 0 && gw > 1) ? 2 : 1;
            for (int step = 0; step < n_steps && !stop; ++step) {
            const int w_from = (base == 0 && step == 1) ? 1 : 0, w_to = (base == 0 && step == 0) ? 1 : gw;
            uint32_t kept_score = 0;  // (what this wave kept in earlier steps is either the chosen board, recorded in sh[5], or beaten)
            int kept_w = 0, kept_ci = 0, kept_cells = 0;
            for (int w = w_from; w < w_to; ++w) {
                uint32_t pub;
                while (!((pub = __hip_atomic_load(&gs[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) & 0x80000000u)) __builtin_amdgcn_s_sleep(2);
                const int nc_w = (int)(pub & 0x7fffffffu);
            for (;;) {
                int ci = lane == 0 ? (int)atomicAdd(&gs[2 * TW + w], 1u) : 0;
                ci = __shfl(ci, 0);
                if (ci >= nc_w) break;
                const u64 q = reinterpret_cast<const u64 *>(lds + OFF_WAVES + w * WV_BYTES + WV_CAND)[ci];
                int cells;
                AGX_TT(const unsigned long long tb0 = wall_clock64();)
                const uint32_t score = (uint32_t)build_board_w(c, wv + WV_SLOT, q, lane, cells, status, ek);
                AGX_TT(tk[8] += wall_clock64() - tb0; tk[9] += (unsigned long long)cells;)
                ++n_boards;
                // the seed's best score and the FIRST candidate that reaches it (what the sequential loop is left with, :616-622)
                if (lane == 0) atomicMax(&gs[TW + w], score << 16 | (uint32_t)(0xffff - ci));
                // The board the merge below may choose is kept (its cells), so that it need not be grown again for
                // try_fix_missing: one that beats the best of the earlier groups and, in the merge's order (score, then seed,
                // then candidate), everything this wave has kept in this group -- whatever the merge chooses is the best in that
                // order of all the group's boards, so the wave that grew it still holds it.
                if (score > best_score && (score > kept_score || (score == kept_score && (w < kept_w || (w == kept_w && ci < kept_ci))))) {
                    kept_score = score;
                    kept_w = w;
                    kept_ci = ci;
                    kept_cells = cells;
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(wv + WV_SLOT);
                    uint32_t *dst = reinterpret_cast<uint32_t *>(wv + WV_KEEP);
                    for (int i = lane; i < SL_GRID / 4; i += 64) dst[i] = src[i];
                    wsync();
                }
            }
            }
            status = wave_or_u(status);
            if (lane == 0) {
                if (status) atomicOr(&sh[0], status);
                gs_other[wave] = 0;
                gs_other[TW + wave] = 0;
                gs_other[2 * TW + wave] = 0;
            }
            __syncthreads();
            int win_w = -1, win_ci = 0;
            for (int w = w_from; w < w_to; ++w) {  // the reference's order
                const uint32_t key = gs[TW + w];
                if ((key >> 16) > best_score) {
                    best_score = key >> 16;
                    win_w = w;
                    win_ci = (int)(0xffff - (key & 0xffffu));
                    best_quad = reinterpret_cast<const u64 *>(lds + OFF_WAVES + w * WV_BYTES + WV_CAND)[win_ci];
                }
                if (best_score >= 36) {
                    stop = true;
                    break;
                }
            }
            if (sh[0]) stop = true;  // the frame goes to the host anyway
            // where the chosen board's cells are kept: sh[5] = the wave + 1 (0: nowhere -- it is grown again), sh[6] = its cells
            if (win_w >= 0) {  // (every thread alike) the group that chose: round and first seed
                win_stamp = ((uint32_t)round + 1u) << 8 | (uint32_t)(base + w_from + 1);
                if (lane == 0 && kept_score == best_score && kept_w == win_w && kept_ci == win_ci) {
                    sh[5] = (uint32_t)wave + 1u;
                    sh[6] = (uint32_t)kept_cells;
                    sh[7] = win_stamp;
                }
            }
            __syncthreads();
            }
        }
        TK(5);
        if (sh[0] || best_score == 0) break;  // (None: the remaining rounds would find nothing either)

        if (wave == 0) {
            // the chosen board again (a board is a function of the saddles and its seed quad), then try_fix_missing +
            // all_tag_indexes (board.rs:49-112), cells in insertion order
            uint8_t *slot = wv + WV_SLOT;
            int best_cells = 0;
            if (sh[7] == win_stamp && sh[5] != 0) {  // kept by the wave that grew it: its cells, and the grid from them
                const uint32_t *src = reinterpret_cast<const uint32_t *>(lds + OFF_WAVES + (sh[5] - 1u) * WV_BYTES + WV_KEEP);
                uint32_t *dst = reinterpret_cast<uint32_t *>(slot);
                best_cells = (int)sh[6];
                for (int i = lane; i < SL_GRID / 4; i += 64) dst[i] = src[i];
                for (int i = lane; i < (SL_ACTIVE - SL_GRID) / 4; i += 64) reinterpret_cast<uint32_t *>(slot + SL_GRID)[i] = 0xffffffffu;
                wsync();
                const int8_t *xy = reinterpret_cast<const int8_t *>(slot + SL_XY);
                for (int i = lane; i < best_cells; i += 64)
                    slot[SL_GRID + (xy[2 * i + 1] + BGR) * BGN + (xy[2 * i] + BGR)] = (uint8_t)(i | (slot[SL_FOUND + i] ? 0x80 : 0));
                wsync();
            } else {
                (void)build_board_w(c, slot, best_quad, lane, best_cells, status, ek);
            }
            int n_quads = 0;
            {
                const int8_t *xy = reinterpret_cast<const int8_t *>(slot + SL_XY);
                uint8_t *found = slot + SL_FOUND;
                auto find = [&](int x, int y) -> int {
                    if (x < -BGR || x > BGR || y < -BGR || y > BGR) return -1;
                    const int e = slot[SL_GRID + (y + BGR) * BGN + (x + BGR)];
                    return e == 0xff ? -1 : (e & 0x7f);
                };
                uint8_t *fa = slot + SL_STACK, *fb = fa + BCELLS;  // the fix list: the two found neighbours ...
                uint8_t *fm = reinterpret_cast<uint8_t *>(wv + WV_PAIRS);  // ... and the cell between them
                int n_fix = 0;
                for (int base = 0; base < best_cells; base += 64) {  // (a cell per lane; the list in the cells' order)
                    const int i = base + lane;
                    bool want = false;
                    int ca = 0, cb = 0;
                    if (i < best_cells && !found[i]) {
                        const int x = xy[2 * i], y = xy[2 * i + 1];
                        const int c0 = find(x + 1, y), c1 = find(x - 1, y);
                        if (c0 >= 0 && c1 >= 0) {
                            want = found[c0] && found[c1];
                            ca = c0;
                            cb = c1;
                        } else {
                            const int c2 = find(x, y + 1), c3 = find(x, y - 1);
                            want = c2 >= 0 && c3 >= 0 && found[c2] && found[c3];
                            ca = c2;
                            cb = c3;
                        }
                    }
                    const u64 mk = __ballot(want);
                    if (want) {
                        const int at = n_fix + __popcll(mk & below);
                        fa[at] = (uint8_t)ca;
                        fb[at] = (uint8_t)cb;
                        fm[at] = (uint8_t)i;
                    }
                    n_fix += __popcll(mk);
                }
                if (n_fix) {  // (rare)
                    wsync();
                    if (lane == 0)
                        for (int k = 0; k < n_fix; ++k) {
                            const u64 q0 = slot_quad(slot, fa[k]), q1 = slot_quad(slot, fb[k]);
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
                            if (v == 2) status |= TAIL_UNCERTAIN;
                            if (v == 1) {  // the cell between the two is the missing one ((b0 + b1) / 2, :100): it exists, not found
                                slot_set_quad(slot, fm[k], q_make(mid[0], mid[1], mid[2], mid[3]));
                                found[fm[k]] = 1;
                            }
                        }
                    wsync();
                }
                for (int base = 0; base < best_cells; base += 64) {  // all_tag_indexes: the found cells' quads, in the cells' order
                    const int i = base + lane;
                    const bool f = i < best_cells && found[i];
                    const u64 mk = __ballot(f);
                    if (f) quads[n_quads + __popcll(mk & below)] = slot_quad(slot, i);
                    n_quads += __popcll(mk);
                }
            }
            wsync();
            TK(6);

            // ---- decode the board's quads (detector.rs:514-527); results in the candidates' space ---------------------
            float *dec_xy = reinterpret_cast<float *>(wv + WV_CAND);           // [BCELLS][8]
            int *dec_id = reinterpret_cast<int *>(wv + WV_PAIRS + BCELLS * 8);    // [BCELLS]: tag id or -1
            u64 *dec_bits = reinterpret_cast<u64 *>(wv + WV_PAIRS);  // [BCELLS] (the pair list is dead)
            for (int base = 0; base < n_quads; base += 8) {  // the sample bits: eight lanes per quad
                const int qi = base + (lane >> 3);
                const bool mine = qi < n_quads;
                const u64 q = quads[mine ? qi : 0];
                float qxy[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qxy[2 * i] = sx[q_at(q, i)];
                    qxy[2 * i + 1] = sy[q_at(q, i)];
                }
                u64 bits = 0;
                const bool have = quad_bits8(a, luma, qxy, lane & 7, bits);
                if (mine && (lane & 7) == 0) {
                    dec_id[qi] = have ? -2 : -1;
                    dec_bits[qi] = bits;
                }
            }
            wsync();
            DK(0);
            for (int base = 0; base < n_quads; base += 4) {  // the family's codes: a row of 16 lanes per quad
                const int qi = base + (lane >> 4);
                if (qi < n_quads && dec_id[qi] == -2) {
                    int id = -1, rot = 0;
                    const bool hit = best_tag_row(a, codes, dec_bits[qi], lane & 15, id, rot);
                    if ((lane & 15) == 0) {
                        dec_id[qi] = hit ? id : -1;
                        if (hit) {
                            const u64 q = quads[qi];
                            for (int i = 0; i < 4; ++i) {  // rotate_left(rot) then reverse, :468-469
                                const int src = q_at(q, ((3 - i) + rot) & 3);
                                dec_xy[8 * qi + 2 * i] = sx[src];
                                dec_xy[8 * qi + 2 * i + 1] = sy[src];
                            }
                        }
                    }
                }
            }
            wsync();
            DK(1);
            for (int i = lane; i < TN / 32; i += 64) used[i] = 0;
            wsync();
            // detected_tags.insert(tag_id, corners) (:520) in the quads' order: a tag keeps the place of its first insertion and the
            // corners of its last; the saddles of every decoded quad are marked (:521-523).  A quad per lane.
            int n_used = 0;
            for (int base = 0; base < n_quads; base += 64) {
                const int qi = base + lane, in_pass = n_quads - base < 64 ? n_quads - base : 64;
                const int id = qi < n_quads ? dec_id[qi] : -1;
                int at = -1;  // its place, if an earlier round or pass inserted it
                if (id >= 0)
                    for (int t = 0; t < n_tags; ++t)
                        if (tagids[t] == (uint32_t)id) at = t;
                int first = lane, last = lane;  // the first and the last quad of this pass with the same id (nearly always itself)
                for (int k = 0; k < in_pass; ++k) {
                    const int other = __shfl(id, k);
                    if (id >= 0 && other == id) {
                        first = k < first ? k : first;
                        last = k > last ? k : last;
                    }
                }
                const bool is_new = id >= 0 && at < 0 && first == lane;
                const u64 mnew = __ballot(is_new);
                if (is_new) at = n_tags + __popcll(mnew & below);
                const int at_first = __shfl(at, first);
                if (id >= 0 && at < 0) at = at_first;
                if ((uint32_t)(n_tags + __popcll(mnew)) > tag_cap) {  // (every lane alike)
                    status |= TAIL_CAPACITY;
                    break;
                }
                if (is_new) tagids[at] = (uint32_t)id;
                if (id >= 0) {
                    if (last == lane) {
                        agx_tag *o = a.tags + (size_t)f * a.tag_stride + at;
                        o->id = (uint32_t)id;
                        for (int i = 0; i < 8; ++i) o->xy[i] = dec_xy[8 * qi + i];
                    }
                    const u64 q = quads[qi];
                    for (int i = 0; i < 4; ++i) {
                        const int sdl = q_at(q, i);
                        const uint32_t bit = 1u << (sdl & 31);
                        n_used += !(atomicOr(&used[sdl >> 5], bit) & bit);
                    }
                }
                n_tags += __popcll(mnew);
                wsync();
            }
            for (int o = 32; o; o >>= 1) n_used += __shfl_xor(n_used, o);
            wsync();
            DK(2);
            TK(7);
            // the saddles of decoded quads leave the list (:528-538), order kept
            int kept = 0;
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                const bool keep = i < n && !((used[i >> 5] >> (i & 31)) & 1u);
                const float x = i < n ? sx[i] : 0.0f, y = i < n ? sy[i] : 0.0f, t = i < n ? st[i] : 0.0f;
                const u64 mk = __ballot(keep);
                wsync();
                if (keep) {
                    const int at = kept + __popcll(mk & below);
                    sx[at] = x;
                    sy[at] = y;
                    st[at] = t;
                }
                kept += __popcll(mk);
                wsync();
            }
            status = wave_or_u(status);
            if (lane == 0) {
                sh[1] = (uint32_t)kept;
                sh[3] = (uint32_t)n_used;
                if (status) atomicOr(&sh[0], status);
            }
        }
        __syncthreads();
        if (sh[0]) break;
        if (sh[3] == 0) break;  // nothing removed: the next round would repeat this one
        n = (int)sh[1];
        __syncthreads();  // (before wave 0 writes the shared words of the next round)
    }
    __syncthreads();
    const uint32_t st_all = sh[0];
#ifdef AGX_TAIL_TIMERS
    if (a.debug >= 2 && f == a.debug_frame && tid == 0)
        printf("tail frame (wave 0): ticks grid %llu seeds %llu seed loop %llu (sort50 %llu lists %llu cands %llu boards %llu) fix %llu decode %llu; seeds %d cands %d; boards built by this wave %d, their cells %llu\n", tk[0], tk[1], tk[5],
               tk[2], tk[3], tk[4], tk[8], tk[6], tk[7], n_seeds_done, n_cands_total, n_boards, tk[9]);
    if (a.debug >= 2 && f == a.debug_frame && tid == 0) printf("  decode: sample bits %llu best_tag %llu tag map + used %llu\n", dk[0], dk[1], dk[2]);
    if (a.debug >= 2 && f == a.debug_frame && tid == 0)
        printf("  expand_one: scan %llu reduce+filter %llu broadcast %llu combos %llu; calls %llu, with all four lists %llu; queries that missed the memo %llu of %llu\n", ek[0], ek[1], ek[2], ek[3], ek[5], ek[6], ek[7], 4 * ek[5]);
#endif
    if (tid == 0) {
        a.table[4 * f] = st_all ? 0u : (uint32_t)n_tags;
        a.table[4 * f + 1] = st_all;
        a.table[4 * f + 2] = (uint32_t)(wall_clock64() - t_start);  // 100 MHz ticks this frame took
        a.table[4 * f + 3] = (uint32_t)n_first | (uint32_t)n_seeds_done << 16;  // saddles; seeds wave 0 listed quads for
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
    hipLaunchKernelGGL(k_board_tail, dim3((unsigned)t.n_frames), dim3(64 * TW), LDS_BYTES, (hipStream_t)stream, t);
    return (int)hipGetLastError();
}

}  // namespace agx
