// tail_kernels.h -- host-visible interface of the device board search + tag decode (tail_kernels.hip).
// Internal to the library; the public boundary is include/aprilgrid_amd.h.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/aprilgrid_amd.h"
#include "chain_kernels.h"

namespace agx {

// Per-frame status word of the device tail (frame row [1]).
enum : uint32_t {
    TAIL_OK = 0u,
    TAIL_UNCERTAIN = 1u,  // an angle comparison fell inside the guard band of its threshold: only the host's own
                          // expression (libm's atan2f / cosf / sinf, as the reference) decides there -- the frame goes to the host tail
    TAIL_CAPACITY = 2u,   // more saddles / candidate quads / board cells / tags than the kernel's fixed lists hold: host tail
    TAIL_CHAIN = 4u       // the chain itself reported an overflow for the frame (no saddle list): reported as such
};

constexpr int TAIL_MAX_SADDLES = 1024;  // saddles of a frame the device tail takes (more: TAIL_CAPACITY)

struct TailArgs {
    // the chain's results of the batch (device): compact agx_saddle array + per-frame counters (n_out, out_offset, flags)
    const float *saddles;
    const FrameCounters *ctr;
    int n_frames;
    // u8 luma for the decode (to_luma8): rows luma_row_stride bytes apart, frames luma_frame_stride bytes apart
    const uint8_t *luma;
    long long luma_frame_stride;
    int luma_row_stride;
    int W, H;
    // TagDetector's family fields (src/detector.rs:17-23) and max_num_of_boards
    int edge, border, hamming, n_codes;
    const uint64_t *codes;  // device copy of the family's code list
    int max_boards;
    // results: tags[f][tag_stride] (at most tag_cap of them written: more is TAIL_CAPACITY), table[f] = {count, status,
    // ticks (100 MHz) the frame took, saddles | seeds << 16}
    agx_tag *tags;
    uint32_t *table;
    uint32_t tag_cap, tag_stride;
    // option "tail_debug_band" (tests of the hand-back path): a white-block angle within this many degrees of 60 / 120 is
    // reported undecided whatever the exact evaluation would say (0 = off: only the kernel's own guard band)
    float debug_band;
    int debug;  // AGX_TAIL_DEBUG >= 2 (and a build with -DAGX_TAIL_TIMERS): frame debug_frame's first wave prints where its time went (100 MHz ticks)
    int debug_frame;  // AGX_TAIL_DEBUG_FRAME (default 0)
};

// Enqueue the device tail of the batch on `stream`; hipError_t.
int launch_board_tail(const TailArgs &t, void *stream);
int init_tail_kernels();  // per-device kernel attributes (current device); hipError_t

}  // namespace agx
