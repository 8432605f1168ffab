// host_tail.hpp -- the irregular tail of TagDetector::detect that stays on the host:
// board search (reference src/detector.rs:543-639, src/board.rs, src/saddle.rs:17-67) and tag
// decode (src/detector.rs:42-169,448-476, src/image_util.rs:39-70).
#pragma once
#include <array>
#include <functional>
#include <cstdint>
#include <vector>

#include "../../include/aprilgrid_amd.h"

namespace agx {

struct FamilyInfo {
    int edge, border, hamming;
    const uint64_t *codes;
    int n_codes;
};

// TagDetector::new's table, src/detector.rs:369-405.  Returns false for an unknown family.
bool family_info(int family, FamilyInfo &out);

using Quad = std::array<int, 4>;

// math_util.rs:15-33
float theta_distance_degree(float t0, float t1);
float angle_degree(float v0x, float v0y, float v1x, float v1y);
// test hook: angle_degree next to the bounded approximation the board search decides with outside
// its guard bands (has_approx 0: the approximation is not used for these operands); v = n x 4 floats
// (coarse / has_coarse, optional: the cheaper first-level approximation and where it is used)
void debug_angle_pairs(const float *v, size_t n, float *exact, float *approx, uint8_t *has_approx, float *coarse = nullptr,
                       uint8_t *has_coarse = nullptr);
// this process's atan2f against libm_f32.h's restatement of glibc's routine on n pseudo-random operand pairs + the special
// cases: the number of disagreements (the device tail is offered only where it is 0)
uint64_t libm_atan2f_mismatches(uint64_t n, uint64_t seed);
// test hook: the white-block angle as the reference evaluates it and as the device tail's binary64 evaluation does (t = n x (theta, v02x, v02y))
void debug_white_block_angles(const float *t, size_t n, float *reference, double *binary64);
// saddle.rs:17-67
bool is_valid_quad(const agx_saddle &s0, const agx_saddle &d0, const agx_saddle &s1, const agx_saddle &d1);

// Workers for the board search of ONE frame (option "tail_threads"; agx_detect_batch parallelises over
// frames instead).  run(n, f) calls f(task) for task = 0..n-1 on the workers and returns when all are done.
struct TailWorkers {
    virtual ~TailWorkers() {}
    virtual int size() const = 0;
    virtual void run(int n, const std::function<void(int)> &f) = 0;
};

// detector.rs:588-639: quads (saddle indices) of the best board, or false (None).  With workers, the
// seeds are taken in waves of size() in the reference's order and merged in that order, so the board
// chosen is the one the sequential loop chooses.
bool try_find_best_board(const std::vector<agx_saddle> &refined, std::vector<Quad> &quads, TailWorkers *workers = nullptr);

// image_util.rs:39-70 (h = 2x3 affine, row-major)
void tag_affine(const float quad_xy[8], int side_bits, float margin, float h[6]);
// detector.rs:124-140, 142-169
uint64_t rotate_bits(uint64_t bits, int edge_bits);
bool best_tag(uint64_t bits, int thres, const uint64_t *codes, int n_codes, int edge_bits, int &idx, int &rot);

// detector.rs:510-539: board search + decode over a saddle list and the u8 luma plane.
// Tags in first-insertion order; a repeated id replaces the earlier corners.
void detect_tail(const FamilyInfo &fam, int max_num_of_boards, std::vector<agx_saddle> refined,
                 const uint8_t *luma8, int width, int height, size_t row_stride, std::vector<agx_tag> &tags,
                 TailWorkers *workers = nullptr);

// The same for a pool thread of agx_detect_batch: the list is copied into the calling thread's scratch, the tags are
// returned in it (valid until the thread's next tail) -- no allocation once the thread has seen a frame of that size.
const std::vector<agx_tag> &detect_tail_scratch(const FamilyInfo &fam, int max_num_of_boards, const agx_saddle *saddles, size_t n_saddles,
                                                const uint8_t *luma8, int width, int height, size_t row_stride);

// image 0.25.9 to_luma8 (call site detector.rs:507)
int luma8(const void *pixels, int width, int height, size_t row_stride, int format, uint8_t *out);

}  // namespace agx
