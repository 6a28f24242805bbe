// internal.h -- what the kernel files of the library share with each other on the host side
// (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

namespace fa_internal {

// The meeting counter of stream `st` on the current device (a device address, 256-byte line of
// its own) and the tag of the launch that is about to use it; NULL: unavailable, launch without
// meetings.  Defined in fedagg.hip, which owns the counters.
unsigned* meeting_word(hipStream_t st, unsigned* epoch);

// Number of compute units of the current device.
int cu_count();

}  // namespace fa_internal
