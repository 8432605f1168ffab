// host_tail.cpp -- board search and tag decode on the host (see host_tail.hpp).
//
// The reference keeps this part on the CPU as well: it is pointer-chasing over a few hundred
// saddles per frame.  Decisions here depend on f32 comparisons, so the f32 expressions keep
// the reference's operand order and this file is compiled with -ffp-contract=off.
#include "host_tail.hpp"
#include "libm_f32.h"

#ifdef AGX_TAIL_PROFILE
#include <chrono>
#endif

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <unordered_map>

namespace agx {

#ifdef AGX_TAIL_PROFILE  // tools/tail_profile: phase timers, compiled in only for that tool
double g_tail_prof[12] = {0};
long g_tail_cnt[12] = {0};
struct TailTimer {
    int slot;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit TailTimer(int s, long n = 1) : slot(s) { g_tail_cnt[s] += n; }
    ~TailTimer() { g_tail_prof[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
#define AGX_TAIL_TIME(slot) TailTimer agx_tail_timer_##slot(slot)
#define AGX_TAIL_COUNT(slot, n) (g_tail_cnt[slot] += (long)(n))
#else
#define AGX_TAIL_TIME(slot) do { } while (0)
#define AGX_TAIL_COUNT(slot, n) do { } while (0)
#endif

#include "tag_families_data.inc"

bool family_info(int family, FamilyInfo &out)
{
    switch (family) {  // src/detector.rs:369-405
    case AGX_T16H5: out = {4, 2, 1, kT16H5, 30}; return true;
    case AGX_T25H7: out = {5, 2, 2, kT25H7, 242}; return true;
    case AGX_T25H9: out = {5, 2, 2, kT25H9, 35}; return true;
    case AGX_T36H11: out = {6, 2, 3, kT36H11, 587}; return true;
    case AGX_T36H11B1: out = {6, 1, 3, kT36H11, 587}; return true;
    default: return false;
    }
}

static const float kPi = 3.14159274101257324219f;

// f32::round (half away from zero) without the call into libm the baseline x86-64 target makes of std::round (no SSE4.1
// rounding instruction): x - trunc(x) is exact in binary32, so the comparison with 0.5 decides exactly what roundf decides.
// Values of magnitude >= 2^23 (and NaN / inf) are their own rounding.  (The sign of a zero result may differ from roundf's --
// every use below converts the result to an integer.  Compared with roundf on all 2^32 bit patterns: equal.)
static inline float round_half_away(float x)
{
    if (!(std::fabs(x) < 8388608.0f)) return x;
    float t = (float)(int32_t)x;
    const float d = x - t;
    if (d >= 0.5f) t += 1.0f;
    else if (d <= -0.5f) t -= 1.0f;
    return t;
}

float theta_distance_degree(float t0, float t1)
{
    float d = t0 - t1 + 90.0f;
    if (d < 0.0f) d += 180.0f;
    else if (d > 180.0f) d -= 180.0f;
    return d > 90.0f ? d - 90.0f : 90.0f - d;
}

static inline float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }
static inline float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }

float angle_degree(float v0x, float v0y, float v1x, float v1y)
{
    AGX_TAIL_COUNT(10, 1);
    return std::atan2(v1y * v0x - v1x * v0y, v0x * v1x + v0y * v1y) * 180.0f / kPi;
}

// The board search compares angles with fixed thresholds (|a0 - a2| > 10, 60 <= |a| <= 120) tens of
// thousands of times per frame, and atan2f is a third of its time.  Almost every comparison is far
// from its threshold, so it is decided from an approximation with a KNOWN error bound; only inside the
// guard band around the threshold is the reference's own expression (angle_degree) evaluated, and it
// alone decides there.  The decisions are therefore exactly the reference's.
//   approximation: atan(z) on [0, 1] by an odd polynomial of degree 13 in binary64 (max error 1.5e-5
//   degrees, tools/fit_atan.py), octant reduction, one rounding to f32;  angle_degree itself is within
//   1e-4 degrees of the true angle (atan2f <= 1 ulp, two f32 roundings, kPi);  so |approximation -
//   angle_degree| < 2e-4 degrees.  kAngleBand is 25 times that; agx_debug_angle_pair + the CPU suite
//   check the bound on millions of inputs.
constexpr float kAngleBand = 0.005f;
// Round 5: a coarser first level in front of it.  The search evaluates ~16 000 of these angles per frame and nearly all
// comparisons are degrees away from their threshold, so most are decided from a three-term odd polynomial in float (max error
// 0.035 degrees against the true angle, tools/fit_atan.py; + 1e-4 of angle_degree's own rounding: < 0.04 degrees from
// angle_degree).  kCoarseBand is 2.5 times that.  Inside the coarse band the fine approximation above decides, inside its band
// the reference's expression -- the decisions are exactly the reference's at every level.
constexpr float kCoarseBand = 0.1f;
struct LazyAngle {
    float ax, ay, bx, by;  // angle_degree(ax, ay, bx, by)
    float yf, xf;          // what angle_degree hands to atan2f
    // (no member initialisers: init_quads keeps ~100 of these per s1 on its stack and must not pay for clearing them;
    // set() writes every field the levels read, "already set?" is the caller's bit mask)
    float coarse, approx, exact_v;
    uint8_t has_coarse, has_approx, fine_done, has_exact;
    void set(float v0x, float v0y, float v1x, float v1y)
    {
        ax = v0x; ay = v0y; bx = v1x; by = v1y;
        coarse = approx = exact_v = 0.0f;
        has_exact = 0;
        fine_done = 0;
        has_approx = 0;
        yf = v1y * v0x - v1x * v0y;
        xf = v0x * v1x + v0y * v1y;
        const float ya = std::fabs(yf), xa = std::fabs(xf);
        const float mx = xa > ya ? xa : ya, mn = xa > ya ? ya : xa;
        // zero, infinite or NaN operands and the sign-of-zero cases of atan2: the exact expression only
        has_coarse = (mx > 0.0f && mx < 3.0e38f && yf != 0.0f) ? 1 : 0;
        if (!has_coarse) return;
        const float z = mn / mx, z2 = z * z;
        float a = z * (0.9953585f + z2 * (-0.2886936f + z2 * 0.07934251f));
        if (ya > xa) a = 1.5707964f - a;
        if (xf < 0.0f) a = 3.1415927f - a;
        if (yf < 0.0f) a = -a;
        coarse = a * 57.29578f;
    }
    // the fine approximation (binary64 polynomial, one rounding); false where it is not used
    bool fine()
    {
        if (!fine_done) {
            fine_done = 1;
            const double ya = std::fabs((double)yf), xa = std::fabs((double)xf);
            const double mx = xa > ya ? xa : ya, mn = xa > ya ? ya : xa;
            has_approx = (mx > 0.0 && mx < 1e300 && yf != 0.0f) ? 1 : 0;
            if (has_approx) {
                const double z = mn / mx, z2 = z * z;
                double a = z * (0.9999961115936159 + z2 * (-0.3331736811416821 + z2 * (0.19807815786497726 + z2 * (-0.13233342317278815 +
                           z2 * (0.07962366987276416 + z2 * (-0.03360421491419842 + z2 * 0.006811790682567682))))));
                if (ya > xa) a = 1.5707963267948966 - a;
                if (xf < 0.0f) a = 3.141592653589793 - a;
                if (yf < 0.0f) a = -a;
                approx = (float)(a * 57.29577951308232);
            }
        }
        return has_approx != 0;
    }
    float exact()
    {
        if (!has_exact) {
            exact_v = angle_degree(ax, ay, bx, by);
            has_exact = 1;
        }
        return exact_v;
    }
};
// fabs(p - q) > limit, p and q being angle_degree values
static inline bool angles_differ_by_more_than(LazyAngle &p, LazyAngle &q, float limit)
{
    if (p.has_coarse && q.has_coarse) {
        const float d = std::fabs(p.coarse - q.coarse);
        if (d > limit + 2.0f * kCoarseBand) return true;
        if (d < limit - 2.0f * kCoarseBand) return false;
    }
    if (p.fine() && q.fine()) {
        const float d = std::fabs(p.approx - q.approx);
        if (d > limit + 2.0f * kAngleBand) return true;
        if (d < limit - 2.0f * kAngleBand) return false;
    }
    return std::fabs(p.exact() - q.exact()) > limit;
}
// lo <= fabs(p) <= hi
static inline bool abs_angle_within(LazyAngle &p, float lo, float hi)
{
    if (p.has_coarse) {
        const float a = std::fabs(p.coarse);
        if (a > lo + kCoarseBand && a < hi - kCoarseBand) return true;
        if (a < lo - kCoarseBand || a > hi + kCoarseBand) return false;
    }
    if (p.fine()) {
        const float a = std::fabs(p.approx);
        if (a > lo + kAngleBand && a < hi - kAngleBand) return true;
        if (a < lo - kAngleBand || a > hi + kAngleBand) return false;
    }
    const float a = std::fabs(p.exact());
    return a >= lo && a <= hi;
}

// is_valid_quad (saddle.rs:17-67) in three parts, so that init_quads can hoist the two that do not
// depend on all four saddles out of its inner loop: the conjunction is the same, every part
// evaluates the reference's expressions unchanged.
//   part 0 (:18-21)   d0 and d1 have the same orientation           -- depends on (d0, d1)
//   part 1 (:26-38)   "filter white block": the diagonal s0 -> s1 is roughly perpendicular to s0's
//                     saddle axis                                     -- depends on (s0, s1)
//   rest   (:40-66)   winding, opposite angles, both d on s1's side  -- all four
static inline bool quad_part0(const agx_saddle &d0, const agx_saddle &d1)
{
    return !(theta_distance_degree(d0.theta, d1.theta) > 5.0f);
}
static inline bool quad_part1_dir(const agx_saddle &s0, const agx_saddle &s1, float cos_th, float sin_th)
{
    LazyAngle ang;  // (cos_th, sin_th) = (cos, sin)(s0.theta / 180 * pi), the same floats for every s1
    ang.set(s1.x - s0.x, s1.y - s0.y, cos_th, sin_th);
    return abs_angle_within(ang, 60.0f, 120.0f);
}
static inline bool quad_part1(const agx_saddle &s0, const agx_saddle &s1)
{
    const float th = s0.theta / 180.0f * kPi;
    return quad_part1_dir(s0, s1, std::cos(th), std::sin(th));
}
static inline bool quad_rest(const agx_saddle &s0, const agx_saddle &d0, const agx_saddle &s1, const agx_saddle &d1)
{
    const float v01x = d0.x - s0.x, v01y = d0.y - s0.y;
    const float v03x = d1.x - s0.x, v03y = d1.y - s0.y;
    const float v02x = s1.x - s0.x, v02y = s1.y - s0.y;
    if (cross2(v01x, v01y, v02x, v02y) * cross2(v02x, v02y, v03x, v03y) < 0.0f) return false;
    const float v12x = s1.x - d0.x, v12y = s1.y - d0.y;
    const float v23x = d1.x - s1.x, v23y = d1.y - s1.y;
    if (cross2(v01x, v01y, v12x, v12y) * cross2(v12x, v12y, v23x, v23y) < 0.0f) return false;
    const float v30x = s0.x - d1.x, v30y = s0.y - d1.y;
    LazyAngle a0, a1, a2, a3;
    a0.set(v01x, v01y, v12x, v12y);
    a2.set(v23x, v23y, v30x, v30y);
    if (angles_differ_by_more_than(a0, a2, 10.0f)) return false;
    a1.set(v12x, v12y, v23x, v23y);
    a3.set(v30x, v30y, v01x, v01y);
    if (angles_differ_by_more_than(a1, a3, 10.0f)) return false;
    if (dot2(v01x, v01y, v02x, v02y) < 0.0f || dot2(v03x, v03y, v02x, v02y) < 0.0f) return false;
    return true;
}

// How many of n pseudo-random operand pairs (all bit patterns, image-sized products, ratios around the reduction's
// interval ends, the special cases) this process's atan2f and libm_f32.h's restatement of glibc's routine disagree on.
// The device tail (tail_kernels.hip) evaluates angle_degree with the restatement: it is offered only where this is 0.
uint64_t libm_atan2f_mismatches(uint64_t n, uint64_t seed)
{
    uint64_t state = seed * 0x9E3779B97F4A7C15ull + 0x243F6A8885A308D3ull, bad = 0;
    auto next = [&]() {  // splitmix64
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto differ = [](float y, float x) {
        const float a = std::atan2(y, x), b = fdlibm_atan2f(y, x);
        return f32_bits(a) != f32_bits(b) && !(a != a && b != b);
    };
    const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-40f, -1e-40f, 3e38f, 0.4375f, 0.6875f, 1.1875f, 2.4375f};
    for (float a : special)
        for (float b : special) bad += differ(a, b);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t r = next(), q = next();
        float y, x;
        switch (i & 3) {
        case 0: y = f32_from_bits((uint32_t)r); x = f32_from_bits((uint32_t)(r >> 32)); break;  // any two floats
        case 1:  // cross and dot products of vectors between image points
        case 2: {
            const float ax = (float)(int32_t)(r & 0xfff) - 2048.0f + (float)((r >> 12) & 0xffff) / 65536.0f;
            const float ay = (float)(int32_t)((r >> 28) & 0xfff) - 2048.0f + (float)((r >> 40) & 0xffff) / 65536.0f;
            const float bx = (float)(int32_t)(q & 0xfff) - 2048.0f + (float)((q >> 12) & 0xffff) / 65536.0f;
            const float by = (float)(int32_t)((q >> 28) & 0xfff) - 2048.0f + (float)((q >> 40) & 0xffff) / 65536.0f;
            y = by * ax - bx * ay;
            x = ax * bx + ay * by;
            break;
        }
        default: {  // ratios near the ends of the reduction intervals
            const float ends[] = {0.4375f, 0.6875f, 1.1875f, 2.4375f, 1.0f};
            x = 1.0f + (float)(r & 0xffffff) / 16777216.0f;
            y = x * ends[(r >> 24) % 5] * (1.0f + ((float)(int32_t)(q & 0xff) - 128.0f) * 5.9604645e-8f);
            if (q >> 63) y = -y;
            if ((q >> 62) & 1) x = -x;
            break;
        }
        }
        bad += differ(y, x);
    }
    return bad;
}

// The white-block angle (saddle.rs:26-38) of n (theta, v02x, v02y) triples as the reference evaluates it (binary32, this process's
// cosf / sinf / atan2f) and as the device tail's decisive evaluation does (binary64 from the binary32 theta, the kernel's own
// conversion constant): the kernel decides from the latter only outside a band around 60 / 120 that must exceed their difference.
void debug_white_block_angles(const float *t, size_t n, float *reference, double *binary64)
{
    const double deg = 180.0 / (double)kPi;
    for (size_t i = 0; i < n; ++i) {
        const float theta = t[3 * i], v02x = t[3 * i + 1], v02y = t[3 * i + 2];
        const float th = theta / 180.0f * kPi;
        reference[i] = std::fabs(angle_degree(v02x, v02y, std::cos(th), std::sin(th)));
        const double sd = std::sin((double)th), cd = std::cos((double)th);
        binary64[i] = std::fabs(std::atan2(sd * (double)v02x - cd * (double)v02y, (double)v02x * cd + (double)v02y * sd)) * deg;
    }
}

void debug_angle_pairs(const float *v, size_t n, float *exact, float *approx, uint8_t *has_approx, float *coarse, uint8_t *has_coarse)
{
    for (size_t i = 0; i < n; ++i) {
        LazyAngle a;
        a.set(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        if (exact) exact[i] = a.exact();
        if (approx) {
            has_approx[i] = a.fine() ? 1 : 0;
            approx[i] = a.approx;
        }
        if (coarse) {
            coarse[i] = a.coarse;
            has_coarse[i] = a.has_coarse;
        }
    }
}

bool is_valid_quad(const agx_saddle &s0, const agx_saddle &d0, const agx_saddle &s1, const agx_saddle &d1)
{
    return quad_part0(d0, d1) && quad_part1(s0, s1) && quad_rest(s0, d0, s1, d1);
}

namespace {

// Nearest-neighbour queries over the saddle set.  The reference uses kdtree 0.8.0's nearest():
// ascending squared distance (f32, folded from 0.0); exact-distance ties are resolved here by
// ascending index.  Implementation: uniform grid buckets, rings of cells around the query are
// examined until k hits are known that are strictly closer than anything outside the examined
// block can be -- so the result is exactly what an exhaustive search returns.
class SaddleIndex {
public:
    struct Hit {
        float d2;
        int idx;
        bool operator<(const Hit &o) const { return d2 < o.d2 || (d2 == o.d2 && idx < o.idx); }
    };
    SaddleIndex() {}
    explicit SaddleIndex(const std::vector<agx_saddle> &pts) { reset(pts); }
    // (Re)build over another saddle set.  The object keeps its storage -- cell lists, candidate vector, the two
    // memo tables -- between frames (detect_tail's per-thread scratch): only the slots the previous set used are cleared.
    void reset(const std::vector<agx_saddle> &pts_in)
    {
        pts_p_ = &pts_in;
        const std::vector<agx_saddle> &pts = pts_in;
        clear_tables();
        const int n = (int)pts.size();
        float x0 = 0, x1 = 1, y0 = 0, y1 = 1;
        if (n) {
            x0 = x1 = pts[0].x;
            y0 = y1 = pts[0].y;
            for (const agx_saddle &p : pts) {
                x0 = std::min(x0, p.x); x1 = std::max(x1, p.x);
                y0 = std::min(y0, p.y); y1 = std::max(y1, p.y);
            }
        }
        ox_ = x0;
        oy_ = y0;
        const double w = std::max(1e-3, (double)x1 - x0), h = std::max(1e-3, (double)y1 - y0);
#ifndef AGX_TAIL_CELL_FACTOR
#define AGX_TAIL_CELL_FACTOR 1.0
#endif
        cell_ = std::max(1.0, std::sqrt(w * h / std::max(1, n)) * AGX_TAIL_CELL_FACTOR);  // ~1 point per cell: a 3-NN query settles on its first 3 x 3 block of ~9 points (1.5: ~20 points; 674 -> 769 tails per second and thread, profiles/r5_host_tail_speed.txt)
        nx_ = std::max(1, (int)std::floor(w / cell_) + 1);
        ny_ = std::max(1, (int)std::floor(h / cell_) + 1);
        start_.assign((size_t)nx_ * ny_ + 1, 0);
        cell_of_.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            cell_of_[i] = cell_y(pts[i].y) * nx_ + cell_x(pts[i].x);
            start_[cell_of_[i] + 1]++;
        }
        for (size_t c = 0; c < (size_t)nx_ * ny_; ++c) start_[c + 1] += start_[c];
        items_.resize((size_t)n);
        fill_.assign(start_.begin(), start_.end() - 1);
        for (int i = 0; i < n; ++i) items_[fill_[cell_of_[i]]++] = i;  // ascending index inside a cell
        // the coordinates again in cell order: a row of cells is one contiguous run of floats (nearest_within)
        px_.resize((size_t)n);
        py_.resize((size_t)n);
        for (int t = 0; t < n; ++t) {
            px_[t] = pts[items_[t]].x;
            py_[t] = pts[items_[t]].y;
        }
    }

    // k nearest, ascending; returns how many were found (min(k, n))
    int nearest(float qx, float qy, int k, Hit *out)
    {
        const std::vector<agx_saddle> &pts_ = *pts_p_;
        const int n = (int)pts_.size();
        const int want = std::min(k, n);
        if (want <= 0) return 0;
        if (want <= 3) return nearest_small(qx, qy, want, out);  // the board search's 1- and 3-NN queries
        if (n <= 1024) {
            // init_quads' 50-NN on a frame's few hundred saddles: every distance (one pass over the contiguous coordinates),
            // then a selection -- cheaper than growing rings of cells with a partial sort per ring (2.0 -> 1.1 us at n = 270).
            // (distance, index) is a total order: the k smallest in order are the same list whichever way they are found
            cand_.resize((size_t)n);
            for (int t = 0; t < n; ++t) {
                const float dx = qx - px_[t], dy = qy - py_[t];
                cand_[(size_t)t] = Hit{(0.0f + dx * dx) + dy * dy, items_[t]};
            }
            if (want < n) std::nth_element(cand_.begin(), cand_.begin() + (want - 1), cand_.end());
            std::sort(cand_.begin(), cand_.begin() + want);
            std::copy(cand_.begin(), cand_.begin() + want, out);
            return want;
        }
        const int cx = cell_x(qx), cy = cell_y(qy);
        cand_.clear();
        int xlo = cx, xhi = cx, ylo = cy, yhi = cy;  // examined block of cells (inclusive)
        scan(xlo, xhi, ylo, yhi, qx, qy);
        for (;;) {
            const bool all = xlo == 0 && ylo == 0 && xhi == nx_ - 1 && yhi == ny_ - 1;
            if ((int)cand_.size() >= want) {
                std::partial_sort(cand_.begin(), cand_.begin() + want, cand_.end());
                if (all) break;
                // distance from the query to the nearest still unexamined region; sides of the
                // block that coincide with the grid border have nothing beyond them
                double gap = 1e300;
                if (xlo > 0) gap = std::min(gap, (double)qx - (ox_ + xlo * cell_));
                if (xhi < nx_ - 1) gap = std::min(gap, (ox_ + (xhi + 1) * cell_) - (double)qx);
                if (ylo > 0) gap = std::min(gap, (double)qy - (oy_ + ylo * cell_));
                if (yhi < ny_ - 1) gap = std::min(gap, (oy_ + (yhi + 1) * cell_) - (double)qy);
                if (gap > 0 && (double)cand_[want - 1].d2 < gap * gap * (1.0 - 1e-6)) break;
            } else if (all) {
                std::sort(cand_.begin(), cand_.end());
                break;
            }
            // grow the block by one ring and scan only the new cells
            const int nxlo = std::max(0, xlo - 1), nxhi = std::min(nx_ - 1, xhi + 1);
            const int nylo = std::max(0, ylo - 1), nyhi = std::min(ny_ - 1, yhi + 1);
            if (nylo < ylo) scan(nxlo, nxhi, nylo, nylo, qx, qy);
            if (nyhi > yhi) scan(nxlo, nxhi, nyhi, nyhi, qx, qy);
            if (nxlo < xlo) scan(nxlo, nxlo, ylo, yhi, qx, qy);
            if (nxhi > xhi) scan(nxhi, nxhi, ylo, yhi, qx, qy);
            xlo = nxlo; xhi = nxhi; ylo = nylo; yhi = nyhi;
        }
        const int m = std::min(want, (int)cand_.size());
        std::copy(cand_.begin(), cand_.begin() + m, out);
        return m;
    }

    // The board search asks the same questions over and over: every query point of
    // find_closest_potential_saddle_idxs (board.rs:177-233) is a function of an ordered pair of
    // saddles, and a board is built for every candidate quad of up to 30 seeds.  The candidates of
    // both query points of a pair -- 3 nearest neighbours, filtered by the radius and the saddle
    // angle, which do not depend on the board -- are memoised per (i0, i1) in an open-addressing
    // table; only the board's own "still unused" test is applied per call.
    struct PairCands {
        uint64_t key = 0;  // 0 = empty slot
        int idx[2][3];
        int n[2];
    };
    const PairCands &pair_candidates(int i0, int i1, float spacing_ratio)
    {
        const std::vector<agx_saddle> &pts_ = *pts_p_;
        AGX_TAIL_COUNT(7, 1);
        const uint64_t key = 1ull + ((uint64_t)(uint32_t)i0 << 32 | (uint64_t)(uint32_t)i1);
        if (pair_tab_.empty()) pair_tab_.resize(1u << 12);
        for (;;) {
            const size_t mask = pair_tab_.size() - 1;
            size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & mask;
            while (pair_tab_[h].key != 0ull && pair_tab_[h].key != key) h = (h + 1) & mask;
            if (pair_tab_[h].key == key) return pair_tab_[h];
            if (pair_used_ * 2 >= pair_tab_.size()) {  // grow and re-insert, then look the slot up again
                std::vector<PairCands> bigger(pair_tab_.size() * 2);
                pair_touched_.clear();
                for (const PairCands &e : pair_tab_)
                    if (e.key) {
                        size_t g = (size_t)((e.key * 0x9E3779B97F4A7C15ull) >> 32) & (bigger.size() - 1);
                        while (bigger[g].key) g = (g + 1) & (bigger.size() - 1);
                        bigger[g] = e;
                        pair_touched_.push_back((uint32_t)g);
                    }
                pair_tab_.swap(bigger);
                continue;
            }
            AGX_TAIL_COUNT(8, 1);
            PairCands e;
            e.key = key;
            const agx_saddle &s0 = pts_[i0], &s1 = pts_[i1];
            const float ratio0 = 1.0f + spacing_ratio;
            const float ex = s0.x - s1.x, ey = s0.y - s1.y;
            const float radius_sq = 0.5f * (ex * ex + ey * ey);
            const float v10x = s1.x - s0.x, v10y = s1.y - s0.y;
            e.n[0] = e.n[1] = 0;
            for (int side = 0; side < 2; ++side) {
                // (no candidate next to s0: try_expand_one's loops over both lists are empty whatever s1's holds -- its query is skipped)
                if (side == 1 && e.n[0] == 0) break;
                const agx_saddle &anchor = side ? s1 : s0;
                Hit hits[3];
                const int m = nearest_small(anchor.x + v10x * ratio0, anchor.y + v10y * ratio0, std::min(3, (int)pts_.size()), hits, radius_sq);
                e.n[side] = 0;
                for (int i = 0; i < m; ++i)
                    if (hits[i].d2 <= radius_sq && theta_distance_degree(anchor.theta, pts_[hits[i].idx].theta) < 5.0f)
                        e.idx[side][e.n[side]++] = hits[i].idx;
            }
            pair_tab_[h] = e;
            pair_touched_.push_back((uint32_t)h);
            ++pair_used_;
            return pair_tab_[h];
        }
    }

    // is_valid_quad is a pure function of four saddles (five atan2f, a sincos): boards grown from
    // different seed quads test the same index quadruples again and again.  Open-addressing table
    // keyed by the four indices (16 bits each; larger sets are evaluated directly).
    bool valid_quad(int i0, int i1, int i2, int i3)
    {
        const std::vector<agx_saddle> &pts_ = *pts_p_;
        if (pts_.size() >= 65535u) return is_valid_quad(pts_[i0], pts_[i1], pts_[i2], pts_[i3]);
        const uint64_t key = 1ull + ((uint64_t)i0 | ((uint64_t)i1 << 16) | ((uint64_t)i2 << 32) | ((uint64_t)i3 << 48));  // != 0
        if (quad_keys_.empty()) {
            quad_keys_.assign(1u << 14, 0ull);
            quad_vals_.assign(1u << 14, 0);
        }
        for (;;) {
            const size_t mask = quad_keys_.size() - 1;
            size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & mask;
            for (;;) {
                const uint64_t k = quad_keys_[h];
                if (k == key) return quad_vals_[h] != 0;
                if (k == 0ull) break;
                h = (h + 1) & mask;
            }
            if (quad_used_ * 2 >= quad_keys_.size()) {  // grow, re-insert, look the slot up again
                std::vector<uint64_t> ok(quad_keys_.size() * 2, 0ull);
                std::vector<uint8_t> ov(ok.size(), 0);
                quad_touched_.clear();
                for (size_t i = 0; i < quad_keys_.size(); ++i)
                    if (quad_keys_[i]) {
                        size_t g = (size_t)((quad_keys_[i] * 0x9E3779B97F4A7C15ull) >> 32) & (ok.size() - 1);
                        while (ok[g]) g = (g + 1) & (ok.size() - 1);
                        ok[g] = quad_keys_[i];
                        ov[g] = quad_vals_[i];
                        quad_touched_.push_back((uint32_t)g);
                    }
                quad_keys_.swap(ok);
                quad_vals_.swap(ov);
                continue;
            }
            AGX_TAIL_COUNT(9, 1);
            const bool v = is_valid_quad(pts_[i0], pts_[i1], pts_[i2], pts_[i3]);
            quad_keys_[h] = key;
            quad_vals_[h] = v ? 1 : 0;
            quad_touched_.push_back((uint32_t)h);
            ++quad_used_;
            return v;
        }
    }

private:
    // nearest() for want <= 3 without the candidate vector and its sorts: blocks of cells around the query are
    // examined under the same stopping rule, the best `want` hits are kept in order (ascending distance, ties by
    // index: a total order, so the result does not depend on the order of examination) while scanning -- the
    // result is the one partial_sort gives.
    // max_d2 >= 0: the caller keeps only hits with d2 <= max_d2 (find_closest_potential_saddle_idxs,
    // board.rs:207-213), so the search also stops once everything unexamined lies beyond that radius --
    // the hits within the radius, and their order, are the same as those of the unbounded search.
    int nearest_small(float qx, float qy, int want, Hit *out, float max_d2 = -1.0f)
    {
        // the best three as 64-bit keys (distance bits : index) -- for the non-negative distances of finite points the
        // integer order of the keys IS the (distance, index) order of Hit::operator< --, kept sorted by a three-element
        // min / max network: no data-dependent branch per point
        uint64_t k0 = ~0ull, k1 = ~0ull, k2 = ~0ull;
        long seen = 0;
        // a block of cells row by row: the cells xa .. xb of a row are one contiguous run of coordinates (px_ / py_ in cell order)
        auto scan_small = [&](int xa, int xb, int ya, int yb) {
            for (int y = ya; y <= yb; ++y) {
                const int t0 = start_[(size_t)y * nx_ + xa], t1 = start_[(size_t)y * nx_ + xb + 1];
                seen += t1 - t0;
                for (int t = t0; t < t1; ++t) {
                    const float dx = qx - px_[t], dy = qy - py_[t];
                    const float d2 = (0.0f + dx * dx) + dy * dy;
                    uint32_t bits;
                    std::memcpy(&bits, &d2, 4);
                    uint64_t x = (uint64_t)bits << 32 | (uint32_t)items_[t];
                    uint64_t lo = std::min(k0, x);
                    x = std::max(k0, x);
                    k0 = lo;
                    lo = std::min(k1, x);
                    x = std::max(k1, x);
                    k1 = lo;
                    k2 = std::min(k2, x);
                }
            }
        };
        auto key_d2 = [](uint64_t k) {
            const uint32_t bits = (uint32_t)(k >> 32);
            float d2;
            std::memcpy(&d2, &bits, 4);
            return d2;
        };
        // the first block is the query's cell with its eight neighbours at once (a cell holds ~1 point: the single cell
        // almost never settles a 3-NN query); the stopping rules below hold for any examined block
        const int cx = cell_x(qx), cy = cell_y(qy);
        int xlo = std::max(0, cx - 1), xhi = std::min(nx_ - 1, cx + 1), ylo = std::max(0, cy - 1), yhi = std::min(ny_ - 1, cy + 1);
        scan_small(xlo, xhi, ylo, yhi);
        for (;;) {
            const bool all = xlo == 0 && ylo == 0 && xhi == nx_ - 1 && yhi == ny_ - 1;
            if (all) break;
            if (seen >= want || max_d2 >= 0.0f) {
                double gap = 1e300;
                if (xlo > 0) gap = std::min(gap, (double)qx - (ox_ + xlo * cell_));
                if (xhi < nx_ - 1) gap = std::min(gap, (ox_ + (xhi + 1) * cell_) - (double)qx);
                if (ylo > 0) gap = std::min(gap, (double)qy - (oy_ + ylo * cell_));
                if (yhi < ny_ - 1) gap = std::min(gap, (oy_ + (yhi + 1) * cell_) - (double)qy);
                const float worst = key_d2(want == 1 ? k0 : (want == 2 ? k1 : k2));  // the want-th best so far
                if (seen >= want && gap > 0 && (double)worst < gap * gap * (1.0 - 1e-6)) break;
                if (max_d2 >= 0.0f && gap > 0 && (double)max_d2 < gap * gap * (1.0 - 1e-6)) break;  // nothing left within the radius
            }
            const int nxlo = std::max(0, xlo - 1), nxhi = std::min(nx_ - 1, xhi + 1);
            const int nylo = std::max(0, ylo - 1), nyhi = std::min(ny_ - 1, yhi + 1);
            if (nylo < ylo) scan_small(nxlo, nxhi, nylo, nylo);
            if (nyhi > yhi) scan_small(nxlo, nxhi, nyhi, nyhi);
            if (nxlo < xlo) scan_small(nxlo, nxlo, ylo, yhi);
            if (nxhi > xhi) scan_small(nxhi, nxhi, ylo, yhi);
            xlo = nxlo; xhi = nxhi; ylo = nylo; yhi = nyhi;
        }
        AGX_TAIL_COUNT(11, seen);
        const int nb = (int)std::min<long>(seen, want);
        const uint64_t ks[3] = {k0, k1, k2};
        for (int i = 0; i < nb; ++i) out[i] = Hit{key_d2(ks[i]), (int)(uint32_t)ks[i]};
        return nb;
    }

    // floor(t) clamped to [0, n - 1] without a call into libm (the baseline x86-64 target has no rounding instruction):
    // truncation is floor for t > 0, anything else (NaN included) is cell 0, and the clamp comes before the conversion
    static int clamped_cell(double t, int n) { return !(t > 0.0) ? 0 : (t >= (double)n ? n - 1 : (int)t); }
    int cell_x(float x) const { return clamped_cell(((double)x - ox_) / cell_, nx_); }
    int cell_y(float y) const { return clamped_cell(((double)y - oy_) / cell_, ny_); }
    void scan(int xa, int xb, int ya, int yb, float qx, float qy)
    {
        const std::vector<agx_saddle> &pts_ = *pts_p_;
        for (int y = ya; y <= yb; ++y)
            for (int x = xa; x <= xb; ++x) {
                const size_t c = (size_t)y * nx_ + x;
                for (int t = start_[c]; t < start_[c + 1]; ++t) {
                    const int i = items_[t];
                    const float dx = qx - pts_[i].x, dy = qy - pts_[i].y;
                    cand_.push_back({(0.0f + dx * dx) + dy * dy, i});
                }
            }
    }

    // the memo tables of the previous saddle set: emptied slot by slot while few were used, wholesale otherwise
    void clear_tables()
    {
        if (!pair_tab_.empty()) {
            if (pair_touched_.size() * 8 < pair_tab_.size())
                for (uint32_t h : pair_touched_) pair_tab_[h].key = 0ull;
            else
                for (PairCands &e : pair_tab_) e.key = 0ull;
        }
        pair_touched_.clear();
        pair_used_ = 0;
        if (!quad_keys_.empty()) {
            if (quad_touched_.size() * 8 < quad_keys_.size())
                for (uint32_t h : quad_touched_) quad_keys_[h] = 0ull;
            else
                std::fill(quad_keys_.begin(), quad_keys_.end(), 0ull);
        }
        quad_touched_.clear();
        quad_used_ = 0;
    }

    const std::vector<agx_saddle> *pts_p_ = nullptr;
    double ox_ = 0, oy_ = 0, cell_ = 1;
    int nx_ = 1, ny_ = 1;
    std::vector<int> start_, items_, cell_of_, fill_;
    std::vector<float> px_, py_;  // pts_[items_[t]].x / .y
    std::vector<uint32_t> pair_touched_, quad_touched_;  // occupied slots of the two tables
    std::vector<Hit> cand_;
    std::vector<PairCands> pair_tab_;
    size_t pair_used_ = 0;
    std::vector<uint64_t> quad_keys_;
    std::vector<uint8_t> quad_vals_;
    size_t quad_used_ = 0;
};

struct CellKey {
    int x, y;
    bool operator==(const CellKey &o) const { return x == o.x && y == o.y; }
};
struct CellKeyHash {
    size_t operator()(const CellKey &k) const
    {
        return (size_t)((uint32_t)k.x * 73856093u) ^ (size_t)((uint32_t)k.y * 19349663u);
    }
};

struct BoardCell {
    CellKey key;
    bool found;
    Quad quad;
};

// What a board owns.  try_find_best_board builds a board for every candidate quad of up to 30 seeds
// (several hundred per frame, most of them dead after a step or two), so the storage is recycled: a
// board undoes exactly what the previous user of the storage touched instead of allocating and
// clearing the used-saddle mask and the cell grid again.
struct BoardStorage {
    static constexpr int kGridR = 24, kGridN = 2 * kGridR + 1;
    std::vector<uint8_t> active;    // board.rs:30-37 active_idxs: 1 = saddle not used by this board yet
    std::vector<int> deactivated;   // indices cleared in `active`
    std::vector<int16_t> grid;      // cell coordinates near the seed -> index into cells (-1 = none)
    std::vector<BoardCell> cells;   // insertion order
    std::unordered_map<CellKey, size_t, CellKeyHash> lookup;  // cells farther than kGridR from the seed

    static bool in_grid(const CellKey &k) { return k.x >= -kGridR && k.x <= kGridR && k.y >= -kGridR && k.y <= kGridR; }
    static size_t grid_at(const CellKey &k) { return (size_t)((k.y + kGridR) * kGridN + (k.x + kGridR)); }
    void begin(size_t n_saddles)
    {
        if (active.size() != n_saddles) active.assign(n_saddles, 1);
        else
            for (int i : deactivated) active[i] = 1;
        deactivated.clear();
        if (grid.empty()) grid.assign((size_t)kGridN * kGridN, (int16_t)-1);
        else
            for (const BoardCell &c : cells)
                if (in_grid(c.key)) grid[grid_at(c.key)] = (int16_t)-1;
        cells.clear();
        if (!lookup.empty()) lookup.clear();
    }
    void use(int saddle)
    {
        active[saddle] = 0;
        deactivated.push_back(saddle);
    }
};

// board::Board, src/board.rs:18-235.  Cells are kept in insertion order (the reference's
// HashMap iteration order is unspecified).
class Board {
public:
    Board(const std::vector<agx_saddle> &refined, SaddleIndex &index, const Quad &seed, float spacing_ratio, BoardStorage &st)
        : refined_(refined), index_(index), st_(st), spacing_ratio_(spacing_ratio)
    {
        st_.begin(refined.size());
        for (int i = 1; i < 4; ++i) st_.use(seed[i]);  // board.rs:35-37
        put({0, 0}, true, seed);
        expand({0, 0});
    }
    // a finished board again, from the storage it was built in (for try_fix_missing / all_tag_indexes)
    Board(const std::vector<agx_saddle> &refined, SaddleIndex &index, BoardStorage &st, unsigned score)
        : refined_(refined), index_(index), st_(st), spacing_ratio_(0.0f), score_(score)
    {
    }
    unsigned score() const { return score_; }

    void collect(std::vector<Quad> &out) const  // all_tag_indexes, board.rs:49-51
    {
        for (const Cell &c : st_.cells)
            if (c.found) out.push_back(c.quad);
    }

    void fix_missing()  // try_fix_missing, board.rs:52-112
    {
        struct Fix { CellKey a, b; };
        std::vector<Fix> fixes;
        for (const Cell &c : st_.cells) {
            if (c.found) continue;
            const CellKey b0{c.key.x + 1, c.key.y}, b1{c.key.x - 1, c.key.y};
            const CellKey b2{c.key.x, c.key.y + 1}, b3{c.key.x, c.key.y - 1};
            const Cell *c0 = get(b0), *c1 = get(b1);
            if (c0 && c1) {
                if (c0->found && c1->found) fixes.push_back({b0, b1});
            } else {
                const Cell *c2 = get(b2), *c3 = get(b3);
                if (c2 && c3 && c2->found && c3->found) fixes.push_back({b2, b3});
            }
        }
        for (const Fix &f : fixes) {
            const Quad q0 = get(f.a)->quad, q1 = get(f.b)->quad;
            Quad mid;
            for (int i = 0; i < 4; ++i) {
                const float x = (refined_[q0[i]].x + refined_[q1[i]].x) / 2.0f;
                const float y = (refined_[q0[i]].y + refined_[q1[i]].y) / 2.0f;
                SaddleIndex::Hit h;
                index_.nearest(x, y, 1, &h);
                mid[i] = h.idx;
            }
            if (index_.valid_quad(mid[0], mid[1], mid[2], mid[3]))
                put({(f.a.x + f.b.x) / 2, (f.a.y + f.b.y) / 2}, true, mid);
        }
    }

private:
    typedef BoardCell Cell;

    // cell coordinates near the seed live in a direct-mapped grid, anything farther in the map
    int find_cell(const CellKey &k) const
    {
        if (BoardStorage::in_grid(k)) return st_.grid[BoardStorage::grid_at(k)];
        auto it = st_.lookup.find(k);
        return it == st_.lookup.end() ? -1 : (int)it->second;
    }
    const Cell *get(const CellKey &k) const
    {
        const int i = find_cell(k);
        return i < 0 ? nullptr : &st_.cells[i];
    }
    void put(const CellKey &k, bool found, const Quad &q)
    {
        const int i = find_cell(k);
        if (i < 0) {
            if (BoardStorage::in_grid(k)) st_.grid[BoardStorage::grid_at(k)] = (int16_t)st_.cells.size();
            else st_.lookup.emplace(k, st_.cells.size());
            st_.cells.push_back({k, found, q});
        } else {
            st_.cells[i].found = found;
            st_.cells[i].quad = q;
        }
    }

    // find_closest_potential_saddle_idxs, board.rs:177-233: the memoised candidates of the pair,
    // minus the saddles this board has already used
    void closest_pair(int i0, int i1, int o0[3], int &n0, int o1[3], int &n1)
    {
        const SaddleIndex::PairCands &pc = index_.pair_candidates(i0, i1, spacing_ratio_);
        n0 = n1 = 0;
        for (int i = 0; i < pc.n[0]; ++i)
            if (st_.active[pc.idx[0][i]]) o0[n0++] = pc.idx[0][i];
        for (int i = 0; i < pc.n[1]; ++i)
            if (st_.active[pc.idx[1][i]]) o1[n1++] = pc.idx[1][i];
    }

    bool expand_one(const Quad &q, Quad &out)  // try_expand_one, board.rs:153-176
    {
        int c0[3], c1[3], c2[3], c3[3], n0, n1, n2, n3;
        closest_pair(q[0], q[1], c0, n0, c1, n1);
        if (n0 == 0 || n1 == 0) return false;  // (the loops below are empty: the other pair need not be looked up)
        closest_pair(q[3], q[2], c3, n3, c2, n2);
        for (int i0 = 0; i0 < n0; ++i0)
            for (int i1 = 0; i1 < n1; ++i1)
                for (int i2 = 0; i2 < n2; ++i2)
                    for (int i3 = 0; i3 < n3; ++i3)
                        if (index_.valid_quad(c0[i0], c1[i1], c2[i2], c3[i3])) {
                            out = {c0[i0], c1[i1], c2[i2], c3[i3]};
                            return true;
                        }
        return false;
    }

    void expand(const CellKey &at)  // try_expand, board.rs:114-152
    {
        const Cell *start = get(at);
        if (!start || !start->found) return;
        const Quad quad = start->quad;
        static const int kDx[4] = {1, 0, -1, 0}, kDy[4] = {0, -1, 0, 1};
        for (int i = 0; i < 4; ++i) {
            Quad qs;
            for (int j = 0; j < 4; ++j) qs[j] = quad[(j + i) & 3];
            const CellKey next{at.x + kDx[i], at.y + kDy[i]};
            const Cell *existing = get(next);
            if (existing && existing->found) continue;
            Quad nq;
            if (expand_one(qs, nq)) {
                Quad v;
                for (int j = 0; j < 4; ++j) v[(j + i) & 3] = nq[j];
                for (int j = 0; j < 4; ++j) st_.use(v[j]);
                score_ += 1;
                put(next, true, v);
                expand(next);
            } else {
                put(next, false, Quad{0, 0, 0, 0});
            }
        }
    }

    const std::vector<agx_saddle> &refined_;
    SaddleIndex &index_;
    BoardStorage &st_;
    float spacing_ratio_;
    unsigned score_ = 1;
};

// init_quads, src/detector.rs:543-586.  The reference tests every (s1, d0, d1) combination with
// is_valid_quad; here the same conjunction is evaluated from tables, because most of its terms do
// not depend on all of s1, d0, d1 (s0 is fixed):
//   (d0, d1)   orientation test (part 0), angle a3 = angle(v30, v01)
//   (s1)       white-block test (part 1)
//   (s1, d)    cross(v0d, v02), cross(v02, v0d), dot(v0d, v02) >= 0, cross(v01, v12), a0 = angle(v01, v12)
//              for d as d0, a2 = angle(v23, v30) for d as d1
//   all four   cross(v12, v23), a1 = angle(v12, v23)
// Every entry is computed by the reference's expression on the reference's operands, so each test
// sees the same floats; only the order of the (side-effect free) tests differs, and three of the
// four atan2 per combination become table look-ups.  Quads are emitted in the reference's order.
void init_quads(const std::vector<agx_saddle> &refined, SaddleIndex &index, int s0_idx, std::vector<Quad> &out)
{
    out.clear();
    const agx_saddle &s0 = refined[s0_idx];
    SaddleIndex::Hit near[50];
    int m;
    {
        AGX_TAIL_TIME(6);
        m = index.nearest(s0.x, s0.y, 50, near);
    }
    int same[50], diff[50];
    int ns = 0, nd = 0;
    for (int i = 1; i < m; ++i) {
        const int idx = near[i].idx;
        const float td = theta_distance_degree(s0.theta, refined[idx].theta);
        if (td < 5.0f) same[ns++] = idx;
        else if (td > 80.0f) diff[nd++] = idx;
    }
    if (!ns || nd < 2) return;
    // (d): v0d = d - s0 (v01 / v03), v30 = s0 - d;  (d0, d1): part 0, a3 on demand
    float v0x[50], v0y[50], v30x[50], v30y[50], dpx[50], dpy[50];  // (dpx / dpy: the d saddles' coordinates side by side for the loops over s1)
    for (int d = 0; d < nd; ++d) {
        const agx_saddle &p = refined[diff[d]];
        dpx[d] = p.x; dpy[d] = p.y;
        v0x[d] = p.x - s0.x; v0y[d] = p.y - s0.y;
        v30x[d] = s0.x - p.x; v30y[d] = s0.y - p.y;
    }
    // the (d0, d1) combinations that pass part 0 as one bit row per a (bit b > a set: the pair is a candidate), and their
    // running number in the reference's order (a ascending, then b).  The loops over s1
    // below visit only the pairs whose two saddles also lie on s1's side of s0 (a second bit row per s1), in that same order.
    uint64_t p0[50];
    static thread_local uint16_t pair_no[50 * 50];         // the running number of pair (a, b), written where the bit is set
    static thread_local std::vector<LazyAngle> a3_store;  // a3 = angle(v30, v01) per listed pair, on demand ...
    static thread_local std::vector<uint64_t> a3_set;      // ... one bit per listed pair: evaluated yet?
    int n_pairs = 0;
    for (int a = 0; a < nd; ++a) {
        uint64_t row = 0;
        for (int b = a + 1; b < nd; ++b)
            if (quad_part0(refined[diff[a]], refined[diff[b]])) {
                row |= 1ull << b;
                pair_no[a * 50 + b] = (uint16_t)n_pairs++;
            }
        p0[a] = row;
    }
    if (!n_pairs) return;
    if (a3_store.size() < (size_t)n_pairs) a3_store.resize((size_t)n_pairs);
    a3_set.assign(((size_t)n_pairs + 63) / 64, 0ull);
    LazyAngle *a3v = a3_store.data();
    const float th0 = s0.theta / 180.0f * kPi;
    const float cos0 = std::cos(th0), sin0 = std::sin(th0);  // part 1's direction of s0, once instead of per s1
    for (int si = 0; si < ns; ++si) {
        const int s1_idx = same[si];
        const agx_saddle &s1 = refined[s1_idx];
        if (!quad_part1_dir(s0, s1, cos0, sin0)) continue;
        const float v02x = s1.x - s0.x, v02y = s1.y - s0.y;
        float cA[50], cB[50], c01[50], v12x[50], v12y[50], v23x[50], v23y[50];
        LazyAngle a0v[50], a2v[50];
        uint64_t a0_set = 0, a2_set = 0;  // which of them have been evaluated for this s1
        uint64_t ok = 0;  // bit d: dot(v0d, v02) >= 0 (saddle.rs:62-64), needed of both d0 and d1
        // the winding test cA[a] * cB[b] < 0 (saddle.rs:44-46) as bit rows: with both factors at least 2^-60 in magnitude the
        // product is a normal number of the factors' signs -- no underflow to a zero that would fail the `< 0` --, so opposite
        // signs decide it; smaller factors (never seen on image coordinates) keep the explicit product below
        uint64_t b_pos = 0, b_neg = 0;
        constexpr float kTiny = 8.67361737988403547e-19f;  // 2^-60
        float dt[50];
        const float s1x = s1.x, s1y = s1.y;
        for (int d = 0; d < nd; ++d) {  // (plain arrays in, plain arrays out: the compiler vectorises this loop)
            cA[d] = v0x[d] * v02y - v0y[d] * v02x;        // cross(v0d, v02): c0 with d as d0 (also the winding test)
            cB[d] = v02x * v0y[d] - v02y * v0x[d];        // cross(v02, v0d): c1 with d as d1
            dt[d] = v0x[d] * v02x + v0y[d] * v02y;        // dot(v0d, v02)
            v12x[d] = s1x - dpx[d]; v12y[d] = s1y - dpy[d];   // d as d0
            v23x[d] = dpx[d] - s1x; v23y[d] = dpy[d] - s1y;   // d as d1
            c01[d] = v0x[d] * v12y[d] - v0y[d] * v12x[d];  // cross(v01, v12)
        }
        for (int d = 0; d < nd; ++d) {
            if (cB[d] >= kTiny) b_pos |= 1ull << d;
            else if (cB[d] <= -kTiny) b_neg |= 1ull << d;
            if (!(dt[d] < 0.0f)) ok |= 1ull << d;
        }
        for (uint64_t as = ok; as;) {
            const int a = __builtin_ctzll(as);
            as &= as - 1;
            const uint64_t wound_wrong = cA[a] >= kTiny ? b_neg : (cA[a] <= -kTiny ? b_pos : 0ull);  // cA[a] * cB[b] < 0 for certain
            for (uint64_t bs = p0[a] & ok & ~wound_wrong; bs;) {
                const int b = __builtin_ctzll(bs);
                bs &= bs - 1;
                if (cA[a] * cB[b] < 0.0f) continue;
                if (c01[a] * cross2(v12x[a], v12y[a], v23x[b], v23y[b]) < 0.0f) continue;
                if (!(a0_set >> a & 1ull)) {
                    a0v[a].set(v0x[a], v0y[a], v12x[a], v12y[a]);
                    a0_set |= 1ull << a;
                }
                if (!(a2_set >> b & 1ull)) {
                    a2v[b].set(v23x[b], v23y[b], v30x[b], v30y[b]);
                    a2_set |= 1ull << b;
                }
                if (angles_differ_by_more_than(a0v[a], a2v[b], 10.0f)) continue;
                const int pi = pair_no[a * 50 + b];
                if (!(a3_set[(size_t)pi >> 6] >> (pi & 63) & 1ull)) {
                    a3v[pi].set(v30x[b], v30y[b], v0x[a], v0y[a]);
                    a3_set[(size_t)pi >> 6] |= 1ull << (pi & 63);
                }
                LazyAngle a1;
                a1.set(v12x[a], v12y[a], v23x[b], v23y[b]);
                if (angles_differ_by_more_than(a1, a3v[pi], 10.0f)) continue;
                if (cA[a] > 0.0f) out.push_back({s0_idx, diff[a], s1_idx, diff[b]});
                else out.push_back({s0_idx, diff[b], s1_idx, diff[a]});
            }
        }
    }
}

inline uint32_t f32_as_u32(float v)  // Rust `as u32`: saturating, NaN -> 0
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

}  // namespace

namespace {

// The seeds' boards on several workers.  Seeds are independent given the running best score: within a
// seed the sequential loop ends up with the FIRST quad that reaches the seed's maximum score if that
// beats the best so far (strictly), so a worker returns (maximum, first quad reaching it) per seed and
// the merge walks the seeds in the reference's order with the reference's rules (strict improvement,
// stop after the first seed that leaves the best at >= 36, at most 30 seeds).  The first seed is taken
// alone -- a board found by it, the common case, then costs what it costs the sequential loop -- and the
// rest in waves of `workers`.  The chosen board is rebuilt from its seed quad for try_fix_missing (a board is a
// deterministic function of the saddles and its seed quad).
bool find_best_board_parallel(const std::vector<agx_saddle> &refined, std::vector<int> &seeds, SaddleIndex &index,
                              std::vector<Quad> &quads, TailWorkers &workers)
{
    struct Ctx {
        SaddleIndex index;
        BoardStorage storage;
        std::vector<Quad> cand;
        explicit Ctx(const std::vector<agx_saddle> &r) : index(r) {}
    };
    struct SeedResult {
        unsigned score = 0;
        Quad quad{};
    };
    const int W = std::max(1, workers.size());
    std::vector<std::unique_ptr<Ctx>> ctx((size_t)W);
    const int n_seeds = (int)std::min<size_t>(seeds.size(), 30);  // popped from the back, at most 30
    unsigned best_score = 0;
    Quad best_quad{};
    bool have = false, stop = false;
    // the first seed alone (on the calling thread): it usually settles the search
    for (int base = 0, width = 1; base < n_seeds && !stop; base += width, width = W) {
        const int n = std::min(width, n_seeds - base);
        std::vector<SeedResult> res((size_t)n);
        workers.run(n, [&](int t) {
            if (!ctx[(size_t)t]) ctx[(size_t)t].reset(new Ctx(refined));
            Ctx &c = *ctx[(size_t)t];
            const int s0 = seeds[seeds.size() - 1 - (size_t)(base + t)];
            init_quads(refined, c.index, s0, c.cand);
            SeedResult r;
            for (const Quad &q : c.cand) {
                const Board b(refined, c.index, q, 0.3f, c.storage);
                if (b.score() > r.score) {
                    r.score = b.score();
                    r.quad = q;
                }
            }
            res[(size_t)t] = r;
        });
        for (int t = 0; t < n; ++t) {  // the reference's order
            if (res[(size_t)t].score > best_score) {
                best_score = res[(size_t)t].score;
                best_quad = res[(size_t)t].quad;
                have = true;
            }
            if (best_score >= 36) {
                stop = true;
                break;
            }
        }
    }
    if (!have) return false;
    BoardStorage storage;
    Board best(refined, index, best_quad, 0.3f, storage);
    best.fix_missing();
    best.collect(quads);
    return true;
}

}  // namespace

namespace {
// What one host thread keeps between frames: the tail of a frame allocates nothing once the thread has seen a frame
// of that size (index, memo tables, board storage, the work lists).  agx_detect_batch runs thousands of tails per
// second on a pool of threads; without this every tail built and zeroed ~0.6 MB of tables through the allocator.
struct TailScratch {
    SaddleIndex index;
    BoardStorage storage[2];  // the best board so far keeps one, the next candidate is built in the other
    std::vector<Quad> cand, quads;
    std::vector<int> seeds;
    std::vector<int> hist;    // round(theta) bins, -180 .. 180
    std::vector<agx_saddle> refined;
    std::vector<uint8_t> used;
    std::vector<agx_tag> tags;
};
TailScratch &tail_scratch()
{
    static thread_local TailScratch s;
    return s;
}
}  // namespace

bool try_find_best_board(const std::vector<agx_saddle> &refined, std::vector<Quad> &quads, TailWorkers *workers)
{
    quads.clear();
    if (refined.empty()) return false;
    TailScratch &sc = tail_scratch();
    {
        AGX_TAIL_TIME(0);
        sc.index.reset(refined);
    }
    SaddleIndex &index = sc.index;
    // seeds: the most populated round(theta) bin (ties -> smallest angle; the reference's
    // HashMap order makes its own tie-break arbitrary), popped from the back
    int best_angle = 0, best_len = -1;
    bool small_angles = true;  // theta is half an atan2 in degrees: (-90, 90]; anything else (a caller's own list) takes the map
    for (const agx_saddle &s : refined)
        if (!(s.theta >= -180.0f && s.theta <= 180.0f)) {
            small_angles = false;
            break;
        }
    if (small_angles) {
        sc.hist.assign(361, 0);
        for (const agx_saddle &s : refined) sc.hist[(size_t)((int)round_half_away(s.theta) + 180)]++;
        for (int a = 0; a < 361; ++a)
            if (sc.hist[(size_t)a] > best_len) {  // ascending angle: the first of equal counts is the smallest
                best_len = sc.hist[(size_t)a];
                best_angle = a - 180;
            }
    } else {
        std::unordered_map<int, int> hist;
        for (const agx_saddle &s : refined) hist[(int)round_half_away(s.theta)]++;
        for (const auto &kv : hist)
            if (kv.second > best_len || (kv.second == best_len && kv.first < best_angle)) {
                best_len = kv.second;
                best_angle = kv.first;
            }
    }
    std::vector<int> &seeds = sc.seeds;
    seeds.clear();
    for (size_t i = 0; i < refined.size(); ++i)
        if ((int)round_half_away(refined[i].theta) == best_angle) seeds.push_back((int)i);

    if (workers && workers->size() > 1 && seeds.size() > 1) return find_best_board_parallel(refined, seeds, index, quads, *workers);

    unsigned best_score = 0;
    BoardStorage *storage = sc.storage;
    int cur = 0, best_at = -1;
    std::vector<Quad> &cand = sc.cand;
    int count = 0;
    while (!seeds.empty() && count < 30) {
        const int s0 = seeds.back();
        seeds.pop_back();
        {
            AGX_TAIL_TIME(1);
            init_quads(refined, index, s0, cand);
        }
        AGX_TAIL_TIME(2);
        AGX_TAIL_COUNT(5, cand.size());
        for (const Quad &q : cand) {
            const Board b(refined, index, q, 0.3f, storage[cur]);
            if (b.score() > best_score) {
                best_score = b.score();
                best_at = cur;
                cur ^= 1;
            }
        }
        if (best_score >= 36) break;
        ++count;
    }
    if (best_at < 0) return false;
    AGX_TAIL_TIME(3);
    Board best(refined, index, storage[best_at], best_score);
    best.fix_missing();
    best.collect(quads);
    return true;
}

void tag_affine(const float quad_xy[8], int side_bits, float margin, float h[6])
{
    // Least-squares affine map from the tag's corner grid (-m,-m), (-m,S), (S,S), (S,-m) to
    // the image quad.  The source points are the corners of an axis-aligned square, so the
    // normal equations diagonalise about its centre.  Evaluated in binary64, rounded once.
    const double S = (double)((float)side_bits - 1.0f + margin), m = (double)margin;
    const double c = 0.5 * (S - m), a = 0.5 * (S + m);
    const double su[4] = {-a, -a, a, a}, sv[4] = {-a, a, a, -a};
    for (int axis = 0; axis < 2; ++axis) {
        double gu = 0, gv = 0, mean = 0;
        for (int p = 0; p < 4; ++p) {
            const double t = quad_xy[2 * p + axis];
            gu += su[p] * t;
            gv += sv[p] * t;
            mean += t;
        }
        const double hu = gu / (4.0 * a * a), hv = gv / (4.0 * a * a);
        h[3 * axis + 0] = (float)hu;
        h[3 * axis + 1] = (float)hv;
        h[3 * axis + 2] = (float)(mean / 4.0 - hu * c - hv * c);
    }
}

uint64_t rotate_bits(uint64_t bits, int edge_bits)
{
    uint64_t out = 0;
    int count = 0;
    for (int r = edge_bits - 1; r >= 0; --r)
        for (int c = 0; c < edge_bits; ++c, ++count) out |= ((bits >> (r + c * edge_bits)) & 1ull) << count;
    return out;
}

// best_tag, src/detector.rs:142-169: up to four rotations x every code of the family (587 for T36H11), one XOR + population
// count each.  The library is built for baseline x86-64, which has no POPCNT instruction (the builtin becomes ~15 instructions
// of bit arithmetic); the same loop compiled for POPCNT is taken where the CPU has it -- every x86-64 since 2008 -- : the decode
// of a quad 2.9 -> 1.9 us, 5 % of the host tail.  Same result either way.
template <typename Popcount>
static inline bool best_tag_impl(uint64_t bits, int thres, const uint64_t *codes, int n_codes, int edge_bits, int &idx, int &rot, Popcount pop)
{
    for (int rotated = 0; rotated < 4; ++rotated) {
        int best = 0;
        unsigned best_score = pop(codes[0] ^ bits);
        for (int i = 1; i < n_codes; ++i) {
            const unsigned s = pop(codes[i] ^ bits);
            if (s < best_score) {
                best_score = s;
                best = i;
            }
        }
        if (best_score < (unsigned)thres) {
            idx = best;
            rot = rotated;
            return true;
        }
        if (rotated == 3) break;
        bits = rotate_bits(bits, edge_bits);
    }
    return false;
}
#if defined(__x86_64__)
__attribute__((target("popcnt"))) static bool best_tag_popcnt(uint64_t bits, int thres, const uint64_t *codes, int n_codes, int edge_bits,
                                                              int &idx, int &rot)
{
    return best_tag_impl(bits, thres, codes, n_codes, edge_bits, idx, rot, [](uint64_t v) __attribute__((target("popcnt"))) { return (unsigned)__builtin_popcountll(v); });
}
#endif
bool best_tag(uint64_t bits, int thres, const uint64_t *codes, int n_codes, int edge_bits, int &idx, int &rot)
{
#if defined(__x86_64__)
    static const bool has_popcnt = __builtin_cpu_supports("popcnt") != 0;
    if (has_popcnt) return best_tag_popcnt(bits, thres, codes, n_codes, edge_bits, idx, rot);
#endif
    return best_tag_impl(bits, thres, codes, n_codes, edge_bits, idx, rot, [](uint64_t v) { return (unsigned)__builtin_popcountll(v); });
}

namespace {

// try_decode_quad, src/detector.rs:448-476 (decode_positions :42-72, bit_code :74-122)
bool decode_quad(const FamilyInfo &fam, const uint8_t *luma8, uint32_t w, uint32_t h, size_t stride,
                 const float quad_xy[8], int &tag_id, float corners[8])
{
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = f32_as_u32(round_half_away(quad_xy[2 * i])), y = f32_as_u32(round_half_away(quad_xy[2 * i + 1]));
        if (x >= w || y >= h) return false;
    }
    float aff[6];
    tag_affine(quad_xy, fam.border * 2 + fam.edge, 0.5f, aff);
    uint8_t samples[64];
    int n = 0;
    for (int gx = fam.border; gx < fam.border + fam.edge; ++gx)
        for (int gy = fam.border; gy < fam.border + fam.edge; ++gy) {
            const float fx = (float)gx, fy = (float)gy;
            const float px = aff[0] * fx + aff[1] * fy + aff[2] * 1.0f;
            const float py = aff[3] * fx + aff[4] * fy + aff[5] * 1.0f;
            const uint32_t ix = f32_as_u32(round_half_away(px)), iy = f32_as_u32(round_half_away(py));
            if (ix >= w || iy >= h) return false;
            samples[n++] = luma8[(size_t)iy * stride + ix];
        }
    int lo = 255, hi = 0;
    for (int i = 0; i < n; ++i) {
        lo = std::min<int>(lo, samples[i]);
        hi = std::max<int>(hi, samples[i]);
    }
    if (hi - lo < 50) return false;
    const int mid = (int)(uint8_t)f32_as_u32(round_half_away(((float)lo + (float)hi) / 2.0f));
    uint64_t bits = 0;
    uint32_t invalid = 0;
    for (int i = 0; i < n; ++i) {  // first sample is the most significant bit
        const int b = samples[n - 1 - i];
        if (std::abs(mid - b) < 10) ++invalid;
        if (b > mid) bits |= 1ull << i;
    }
    if (invalid > 3) return false;
    int idx, rot;
    if (!best_tag(bits, fam.hamming, fam.codes, fam.n_codes, fam.edge, idx, rot)) return false;
    // rotate_left(rot) then reverse, :468-469
    for (int i = 0; i < 4; ++i) {
        const int src = ((3 - i) + rot) & 3;
        corners[2 * i] = quad_xy[2 * src];
        corners[2 * i + 1] = quad_xy[2 * src + 1];
    }
    tag_id = idx;
    return true;
}

}  // namespace

namespace {
// detect's loop body (detector.rs:510-539) over a working copy of the saddle list (consumed: the saddles of decoded
// quads are removed between the rounds)
void detect_tail_rounds(const FamilyInfo &fam, int max_num_of_boards, std::vector<agx_saddle> &refined,
                        const uint8_t *luma8, int width, int height, size_t row_stride, std::vector<agx_tag> &tags,
                        TailWorkers *workers)
{
    TailScratch &sc = tail_scratch();
    tags.clear();
    std::vector<Quad> &quads = sc.quads;
    for (int round = 0; round < max_num_of_boards; ++round) {
        if (!try_find_best_board(refined, quads, workers)) continue;
        std::vector<uint8_t> &used = sc.used;
        used.assign(refined.size(), 0);
        for (const Quad &q : quads) {
            float qxy[8];
            for (int i = 0; i < 4; ++i) {
                qxy[2 * i] = refined[q[i]].x;
                qxy[2 * i + 1] = refined[q[i]].y;
            }
            int id;
            float corners[8];
            AGX_TAIL_TIME(4);
            if (!decode_quad(fam, luma8, (uint32_t)width, (uint32_t)height, row_stride, qxy, id, corners)) continue;
            auto it = std::find_if(tags.begin(), tags.end(), [&](const agx_tag &t) { return t.id == (uint32_t)id; });
            if (it == tags.end()) {
                tags.push_back(agx_tag{});
                it = tags.end() - 1;
                it->id = (uint32_t)id;
            }
            std::memcpy(it->xy, corners, sizeof(corners));
            for (int i = 0; i < 4; ++i) used[q[i]] = 1;
        }
        size_t keep = 0;
        for (size_t i = 0; i < refined.size(); ++i)
            if (!used[i]) refined[keep++] = refined[i];
        refined.resize(keep);
    }
}
}  // namespace

void detect_tail(const FamilyInfo &fam, int max_num_of_boards, std::vector<agx_saddle> refined,
                 const uint8_t *luma8, int width, int height, size_t row_stride, std::vector<agx_tag> &tags,
                 TailWorkers *workers)
{
    detect_tail_rounds(fam, max_num_of_boards, refined, luma8, width, height, row_stride, tags, workers);
}

const std::vector<agx_tag> &detect_tail_scratch(const FamilyInfo &fam, int max_num_of_boards, const agx_saddle *saddles, size_t n_saddles,
                                                const uint8_t *luma8, int width, int height, size_t row_stride)
{
    TailScratch &sc = tail_scratch();
    sc.refined.assign(saddles, saddles + n_saddles);
    detect_tail_rounds(fam, max_num_of_boards, sc.refined, luma8, width, height, row_stride, sc.tags, nullptr);
    return sc.tags;
}

int luma8(const void *pixels, int width, int height, size_t row_stride, int format, uint8_t *out)
{
    for (int y = 0; y < height; ++y) {
        const uint8_t *row = (const uint8_t *)pixels + (size_t)y * row_stride;
        uint8_t *o = out + (size_t)y * width;
        switch (format) {
        case AGX_L8: std::memcpy(o, row, (size_t)width); break;
        case AGX_L16: {
            const uint16_t *r16 = (const uint16_t *)row;
            for (int x = 0; x < width; ++x) o[x] = (uint8_t)(((uint32_t)r16[x] + 128u) / 257u);
            break;
        }
        case AGX_RGB8:
            for (int x = 0; x < width; ++x)
                o[x] = (uint8_t)((2126u * row[3 * x] + 7152u * row[3 * x + 1] + 722u * row[3 * x + 2]) / 10000u);
            break;
        default: return AGX_ERR_FORMAT;
        }
    }
    return AGX_OK;
}

}  // namespace agx
