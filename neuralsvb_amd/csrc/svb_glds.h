// svb_glds.h -- LDS-DMA (global -> LDS without VGPRs) as an instruction the compiler does not track.
//
// hipcc waits `vmcnt(0)` in front of every ds_read that follows a `__builtin_amdgcn_global_load_lds` it cannot prove
// disjoint from the read (measured in conv1d_tw.hip: the wait landed behind the second MFMA of every K phase and
// drained the phase's whole prefetch).  Issued from an asm statement the copy is invisible to that bookkeeping: the kernel
// counts it itself (s_waitcnt vmcnt + barrier before the first read of the destination; cdna_hip_programming.md 5.7).
// Included as <svb_glds.h>: tests/emu/include/ holds the lane emulator's memcpy stand-in.
#pragma once
#include <hip/hip_runtime.h>

// lane L of the wave copies the 16 bytes at its own `gsrc` to byte `byte_off` + 16 L of the workgroup's LDS array `lds`
// (`lds`: the __shared__ array itself, `byte_off`: wave-uniform)
__device__ __forceinline__ void svb_glds16(const void* gsrc, void* lds, unsigned byte_off) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + byte_off);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// makes the 16 registers of an accumulator opaque to the optimiser at this point (no instruction is emitted)
typedef float svb_glds_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void svb_opaque16(svb_glds_f32x16& v) { asm volatile("" : "+v"(v)); }
