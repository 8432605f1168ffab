// detector_internal.h -- library-internal accessors of a detector handle (not part of the ABI).
#pragma once
#include "../../include/aprilgrid_amd.h"

extern "C" {
// hipStream_t the detector's kernels are enqueued on (its own stream, or the caller's after
// agx_detector_set_stream) and its device ordinal.
void *agx_internal_stream(agx_detector *det);
int agx_internal_device(const agx_detector *det);
}
