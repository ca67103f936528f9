// TEST INFRASTRUCTURE ONLY: lane-emulator stand-in for csrc/svb_glds.h (the copy completes immediately; the hardware's
// asynchrony -- vmcnt, then a barrier -- is only exercised on the GPU).
#pragma once
#include <hip/hip_runtime.h>
static inline void svb_glds16(const void* gsrc, void* lds, unsigned byte_off) {
    memcpy((char*)lds + byte_off + 16 * emu_lane_id(), gsrc, 16);
}
typedef float svb_glds_f32x16 __attribute__((ext_vector_type(16)));
static inline void svb_opaque16(svb_glds_f32x16&) {}
