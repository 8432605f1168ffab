// detector_internal.h -- library-internal accessors of a detector handle (not part of the ABI).
#pragma once
#include "../../include/aprilgrid_amd.h"

extern "C" {
// hipStream_t the detector's kernels are enqueued on (its own stream, or the caller's after
// agx_detector_set_stream) and its device ordinal.
void *agx_internal_stream(agx_detector *det);
int agx_internal_device(const agx_detector *det);
// agx_detect_batch: the detector's worker pool (created on first use, resized when n_threads
// changes), its family table (agx::FamilyInfo), max_num_of_boards and a device staging buffer of
// at least `bytes` (grown on demand).
void *agx_internal_pool(agx_detector *det, int n_threads);
const void *agx_internal_family(const agx_detector *det);
int agx_internal_max_boards(const agx_detector *det);
void *agx_internal_stage(agx_detector *det, size_t bytes);
// Wait for whatever is enqueued and forget it (an error path between enqueue and fetch).
void agx_internal_abandon_batch(agx_detector *det);
#define AGX_UPLOAD_STREAMS 3
int agx_internal_upload_streams(agx_detector *det, void **streams /* [AGX_UPLOAD_STREAMS] */);  // hipError_t
// u8 luma of a chunk of L16 / RGB8 device frames, computed on the device and copied to pinned host
// memory behind the detector's stream ([n_frames][H][W] at *h_out once the stream has been waited for);
// the staging is a ring of n_slots chunks of chunk_capacity_frames
int agx_internal_chunk_luma8(agx_detector *det, const void *d_frames, int n_frames, int width, int height, size_t row_stride,
                             size_t frame_stride, int format, int slot, int n_slots, size_t chunk_capacity_frames,
                             const uint8_t **h_out, const uint8_t **d_out);  // (h_out null: no copy to the host; d_out: where it is on the device)
// the last batch's compact list in the detector's pinned host mirror (valid until the next enqueue) + per-frame
// counts / offsets / status; waits for the device
int agx_internal_fetch_compact(agx_detector *det, const agx_saddle **records, uint32_t *counts, uint32_t *offsets, int *status);
// option "device_tail": board search + decode of the enqueued batch on the device (tail_kernels.hip); the results in mapped
// pinned host memory after agx_internal_fetch_tail: tags[f * *tag_cap ..] (the rows' stride, >= the cap that was enqueued), table[4 f] = count, table[4 f + 1] = agx::TAIL_* status (+ 2: the frame's 100 MHz ticks, + 3: saddles | seeds << 16)
int agx_internal_device_tail(agx_detector *det);
int agx_internal_tail_prepare(agx_detector *det);  // one-time set-up on the handle's device (code list, kernel attributes): all or nothing
int agx_internal_tail_debug(const agx_detector *det);  // AGX_TAIL_DEBUG as read when the handle was created
int agx_internal_enqueue_tail(agx_detector *det, const void *d_luma, size_t luma_row_stride, size_t luma_frame_stride, uint32_t tag_cap);
int agx_internal_fetch_tail(agx_detector *det, const agx_tag **tags, const uint32_t **table, uint32_t *tag_cap);
void agx_internal_tail_stats(agx_detector *det, int frames, int fallbacks, int uncertain);
}
namespace agx {
void destroy_worker_pool(void *pool);
void *create_worker_pool(int n_threads);
struct TailWorkers;  // host_tail.hpp
TailWorkers *create_tail_workers(int n_threads);  // nullptr for n_threads <= 1
void destroy_tail_workers(TailWorkers *w);
}
