// Shared device helpers for the svb HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#define SVB_OK 0
#define SVB_ERR_ARG (-1)
#define SVB_ERR_LAUNCH (-2)
#define SVB_ERR_UNSUPPORTED (-3)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Output activation codes shared by host wrappers and kernels.
enum { SVB_ACT_NONE = 0, SVB_ACT_RELU = 1, SVB_ACT_LRELU = 2, SVB_ACT_TANH = 3 };

__device__ __forceinline__ float svb_gate(float g, float slope) { return g > 0.f ? 1.f : slope; }

__device__ __forceinline__ float svb_apply_act(float v, int act, float slope) {
    if (act == SVB_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == SVB_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == SVB_ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float svb_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// 64-lane wave reductions (wavefront = 64 on CDNA).
__device__ __forceinline__ float svb_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float svb_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blocks of NT threads (NT multiple of 64, <= 1024). `red` is >= NT/64 floats of LDS.
template <int NT>
__device__ __forceinline__ float svb_block_sum(float v, float* red) {
    v = svb_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += red[w];
    return t;
}

// (q, m) = (i / W, i % W) of the elements i = threadIdx.x + 256 k a thread of a 256-thread workgroup visits, without a division per
// element.  A wave64 VALU instruction issues in 4 cycles on CDNA and a runtime integer division is ~20 of them: in the stencil and
// plane kernels (ssim.hip, conv2d.hip) the index arithmetic was a third to a half of the instruction stream (round 4).
struct SvbDiv256 {
    int q, m, dq, dm, W;
    __device__ __forceinline__ explicit SvbDiv256(int W_) : W(W_) {
        q = (int)threadIdx.x / W_; m = (int)threadIdx.x - q * W_;
        dq = 256 / W_; dm = 256 - dq * W_;
    }
    __device__ __forceinline__ void next() {
        q += dq; m += dm;
        if (m >= W) { m -= W; ++q; }
    }
};

static inline int svb_cdiv(int a, int b) { return (a + b - 1) / b; }

// Experiment switches and cycle stamps exist only in the instrumentation build (`make instr` -> libsvb_hip_instr.so,
// compiled with -DSVB_INSTRUMENT, loaded by tools/ only); in the product library every switch is its default, a constant.
#ifdef SVB_INSTRUMENT
#include <stdlib.h>
#define SVB_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define SVB_ENV_LONG(name, dflt) (getenv(name) ? atol(getenv(name)) : (long)(dflt))
#define SVB_ENV_FLAG(name) (getenv(name) != nullptr)
#else
#define SVB_ENV_INT(name, dflt) (dflt)
#define SVB_ENV_LONG(name, dflt) ((long)(dflt))
#define SVB_ENV_FLAG(name) false
#endif

#define SVB_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return SVB_ERR_LAUNCH; \
    } while (0)
