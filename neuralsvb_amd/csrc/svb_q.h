// svb_q.h -- the pre-split activation layout "Q" consumed by the bf16x3 conv kernels, and the fp32 -> bf16 hi/lo split.
//
// An fp32 activation tensor [B][C][T] has a Q image  [B][ceil(C/16)][T][32 bf16]:  for (b, 16-channel chunk, position)
// one 64-byte row = 16 hi values then 16 lo values (v = hi + lo, hi = rne_bf16(v), lo = rne_bf16(v - hi)); channels
// beyond C in the last chunk are zero.  A producer writes it once; every consuming conv block then stages its x tile with
// plain 16-byte copies (no conversion, 4x fewer load instructions) instead of re-splitting the tile per consumer.
#pragma once
#include "svb_common.h"

typedef __bf16 svbq_bf2 __attribute__((ext_vector_type(2)));
typedef float svbq_f2 __attribute__((ext_vector_type(2)));

// two fp32 values -> packed bf16 pairs  hi = rne(v), lo = rne(v - hi)   (v_cvt_pk_bf16_f32 x2 + 3 VALU ops on gfx950)
__device__ __forceinline__ void svbq_split2(float v0, float v1, unsigned& hi, unsigned& lo) {
    const svbq_f2 v = {v0, v1};
    const svbq_bf2 h = __builtin_convertvector(v, svbq_bf2);
    const svbq_f2 hf = __builtin_convertvector(h, svbq_f2);
    const svbq_bf2 l = __builtin_convertvector(v - hf, svbq_bf2);
    __builtin_memcpy(&hi, &h, 4);
    __builtin_memcpy(&lo, &l, 4);
}

// One thread's Q rows for VEC consecutive positions of one (batch, chunk): filled two channels at a time, then stored as
// VEC contiguous 64-byte rows.
template <int VEC>
struct SvbQRows {
    unsigned hi[VEC][8], lo[VEC][8];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int p = 0; p < VEC; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) { hi[p][e] = 0u; lo[p][e] = 0u; }
    }
    // channels 2e and 2e+1 of the chunk, at position p
    __device__ __forceinline__ void put(int p, int e, float v0, float v1) { svbq_split2(v0, v1, hi[p][e], lo[p][e]); }
    // rows of positions t0 .. t0+VEC-1 of (b, chunk); q points at the tensor's Q image
    __device__ __forceinline__ void store(unsigned short* q, size_t row0) const {
        uint4* dst = reinterpret_cast<uint4*>(q) + row0 * 4;
#pragma unroll
        for (int p = 0; p < VEC; ++p) {
            dst[p * 4 + 0] = make_uint4(hi[p][0], hi[p][1], hi[p][2], hi[p][3]);
            dst[p * 4 + 1] = make_uint4(hi[p][4], hi[p][5], hi[p][6], hi[p][7]);
            dst[p * 4 + 2] = make_uint4(lo[p][0], lo[p][1], lo[p][2], lo[p][3]);
            dst[p * 4 + 3] = make_uint4(lo[p][4], lo[p][5], lo[p][6], lo[p][7]);
        }
    }
};

__device__ __forceinline__ void svbq_ldv(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void svbq_ldv(const float* p, float (&v)[1]) { v[0] = p[0]; }
__device__ __forceinline__ void svbq_stv(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void svbq_stv(float* p, const float (&v)[1]) { p[0] = v[0]; }
