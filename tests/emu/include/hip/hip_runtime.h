// TEST INFRASTRUCTURE ONLY -- a lane-level CPU emulator of the subset of HIP / gfx950 used by
// neuralsvb_amd/csrc.  It is NOT a product path and is never loaded by neuralsvb_amd: the build
// container has no GPU, so tests/emu compiles the *unmodified* kernel sources against this header
// (it shadows <hip/hip_runtime.h> via -I) to check index math, LDS staging, MFMA fragment maps and
// barriers before a kernel is sent to a real MI355X.  Every HIP thread is a ucontext fiber; a
// workgroup's fibers are scheduled round-robin and blocks run one after another.
//
// Emulated semantics (must match the gfx950 ISA; layouts from cdna_hip_programming.md §3):
//   __builtin_amdgcn_mfma_f32_32x32x2f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//        D reg r of lane l -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31; k-ordered fmaf chain.
//   __builtin_amdgcn_mfma_f32_16x16x4f32 : A[l&15][k=l>>4], B[k=l>>4][l&15],
//        D reg r of lane l -> row (l>>4)*4+r, col l&15.
//   __shfl / __shfl_xor / __shfl_down / __shfl_up / __ballot over 64-lane waves.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define SVB_HIP_EMU 1

struct emu_uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern emu_uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
#define hipDeviceAttributeMultiprocessorCount 0
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 4; return hipSuccess; }   // a 4-CU "device": persistent kernels walk several tiles
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));
typedef int int4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint2 __attribute__((ext_vector_type(2)));
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

// ---- runtime hooks (emu_runtime.cpp) ----
void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body);
void emu_syncthreads();
void emu_wave_exchange_begin(const void* src, size_t bytes);  // publish this lane's payload
const unsigned char* emu_wave_slot(int lane);                 // read another lane's payload
void emu_wave_exchange_end();
int emu_lane_id();

#define hipFuncAttributeMaxDynamicSharedMemorySize 0
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
extern unsigned char* emu_dyn_smem;
void emu_set_dyn_smem(size_t bytes);
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu_dyn_smem);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (emu_set_dyn_smem(shmem), emu_launch((grid), (block), [=]() { kernel(__VA_ARGS__); }))

static inline void __syncthreads() { emu_syncthreads(); }
static inline void __builtin_amdgcn_s_barrier_emu() { emu_syncthreads(); }
#define __builtin_amdgcn_s_barrier() emu_syncthreads()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
void emu_wave_sync();                                             // all lanes of the calling wave arrive before any continues
#define __builtin_amdgcn_wave_barrier() emu_wave_sync()
#define __builtin_amdgcn_fence(order, scope) ((void)0)             // (the emulator's lanes run one at a time: program order is memory order)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
// buffer descriptors: (base, bytes); loads outside the range return 0, stores outside are dropped
struct emu_buffer_rsrc { char* base; unsigned bytes; };
typedef emu_buffer_rsrc __amdgpu_buffer_rsrc_t;
static inline emu_buffer_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short, unsigned bytes, int) { return emu_buffer_rsrc{(char*)p, bytes}; }
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    unsigned v = 0;
    if ((unsigned long long)voff + soff + 4 <= r.bytes) memcpy(&v, r.base + voff + soff, 4);
    return v;
}
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    emu_u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned long long)voff + soff + 16 <= r.bytes) memcpy(&v, r.base + voff + soff, 16);
    return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    if ((unsigned long long)voff + soff + 16 <= r.bytes) memcpy(r.base + voff + soff, &v, 16);
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, emu_buffer_rsrc r, unsigned voff, unsigned soff, int) {
    if ((unsigned long long)voff + soff + 4 <= r.bytes) memcpy(r.base + voff + soff, &v, 4);
}
#define __builtin_readcyclecounter() 0ull

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

static inline emu_f32x16 emu_mfma_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    float ab[2] = {a, b};
    emu_wave_exchange_begin(ab, sizeof(ab));
    const int l = emu_lane_id();
    const int col = l & 31;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            const float av = ((const float*)emu_wave_slot(row + 32 * k))[0];   // A[i=row][k]
            const float bv = ((const float*)emu_wave_slot(col + 32 * k))[1];   // B[k][j=col]
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    emu_wave_exchange_end();
    return d;
}
static inline emu_f32x4 emu_mfma_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    float ab[2] = {a, b};
    emu_wave_exchange_begin(ab, sizeof(ab));
    const int l = emu_lane_id();
    const int col = l & 15;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            const float av = ((const float*)emu_wave_slot(row + 16 * k))[0];
            const float bv = ((const float*)emu_wave_slot(col + 16 * k))[1];
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    emu_wave_exchange_end();
    return d;
}
// v_mfma_f32_32x32x16_bf16: lane l supplies A[i=l&31][k=8*(l>>5)+e] and B[k=8*(l>>5)+e][j=l&31], e=0..7 (bf16);
// D layout as the f32 32x32 form.  Products are exact in fp32 (8-bit mantissas), accumulated k-ascending in fp32.
// (host clang has a native __bf16 with round-to-nearest-even conversions, the same as v_cvt_pk_bf16_f32)
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
static inline float emu_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline emu_f32x16 emu_mfma_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c, int, int, int) {
    unsigned short ab[16];
    memcpy(ab, &a, 16);
    memcpy(ab + 8, &b, 16);
    emu_wave_exchange_begin(ab, sizeof(ab));
    const int l = emu_lane_id();
    const int col = l & 31;
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const unsigned short* pa = (const unsigned short*)emu_wave_slot(row + 32 * (k >> 3));
            const unsigned short* pb = (const unsigned short*)emu_wave_slot(col + 32 * (k >> 3));
            acc = fmaf(emu_bf16_to_f32(pa[k & 7]), emu_bf16_to_f32(pb[8 + (k & 7)]), acc);
        }
        d[r] = acc;
    }
    emu_wave_exchange_end();
    return d;
}
// v_mfma_f32_16x16x32_bf16: lane l supplies A[i=l&15][k=8*(l>>4)+e] and B[k=8*(l>>4)+e][j=l&15], e=0..7; D: col = l&15,
// row = 4*(l>>4) + reg (cdna_hip_programming.md, fragment layout).
static inline emu_f32x4 emu_mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
    unsigned short ab[16];
    memcpy(ab, &a, 16);
    memcpy(ab + 8, &b, 16);
    emu_wave_exchange_begin(ab, sizeof(ab));
    const int l = emu_lane_id();
    const int col = l & 15;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            const unsigned short* pa = (const unsigned short*)emu_wave_slot(row + 16 * (k >> 3));
            const unsigned short* pb = (const unsigned short*)emu_wave_slot(col + 16 * (k >> 3));
            acc = fmaf(emu_bf16_to_f32(pa[k & 7]), emu_bf16_to_f32(pb[8 + (k & 7)]), acc);
        }
        d[r] = acc;
    }
    emu_wave_exchange_end();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu_mfma_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_16x16x4f32

template <typename T>
static inline T emu_shfl_from(T v, int src_lane) {
    emu_wave_exchange_begin(&v, sizeof(T));
    T out;
    memcpy(&out, emu_wave_slot(src_lane & 63), sizeof(T));
    emu_wave_exchange_end();
    return out;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    const int l = emu_lane_id();
    return emu_shfl_from(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    const int l = emu_lane_id();
    const int s = l ^ mask;
    return emu_shfl_from(v, ((s & ~(width - 1)) == (l & ~(width - 1))) ? s : l);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int l = emu_lane_id();
    const int s = l + (int)d;
    return emu_shfl_from(v, ((s & ~(width - 1)) == (l & ~(width - 1))) ? s : l);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int l = emu_lane_id();
    const int s = l - (int)d;
    return emu_shfl_from(v, (s >= 0 && (s & ~(width - 1)) == (l & ~(width - 1))) ? s : l);
}
// __ballot: bit l of the result = predicate of lane l (all 64 lanes take part in the exchange)
static inline unsigned long long __ballot(int pred) {
    int p = pred ? 1 : 0;
    emu_wave_exchange_begin(&p, sizeof(int));
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        int q;
        memcpy(&q, emu_wave_slot(l), sizeof(int));
        if (q) m |= 1ull << l;
    }
    emu_wave_exchange_end();
    return m;
}
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline float __builtin_amdgcn_readfirstlane_f(float v) { return emu_shfl_from(v, 0); }
static inline int __builtin_amdgcn_readfirstlane_emu(int v) { return emu_shfl_from(v, 0); }
#define __builtin_amdgcn_readfirstlane(v) __builtin_amdgcn_readfirstlane_emu(v)

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }

static inline float __fdividef(float a, float b) { return a / b; }
static inline double __dadd_rn(double a, double b) { volatile double r = a + b; return r; }   /* (volatile: no contraction) */
static inline double __dsub_rn(double a, double b) { volatile double r = a - b; return r; }
static inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
static inline double __ddiv_rn(double a, double b) { volatile double r = a / b; return r; }
#define __expf(x) expf(x)            /* (glibc declares but does not export __expf) */
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline void sincosf_emu(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
static inline float sinpif(float x) { return (float)sin(M_PI * (double)x); }
static inline float cospif(float x) { return (float)cos(M_PI * (double)x); }
static inline int __float2int_rn(float x) { return (int)nearbyintf(x); }
static inline float __int2float_rn(int x) { return (float)x; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

// ---- events / stream waits used by the host-side stack executor (wn_stack.hip): the emulator runs every launch to completion
// in program order on one "stream", so these are no-ops
typedef void* hipEvent_t;
#define hipEventDisableTiming 0
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
