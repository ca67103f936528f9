// loss_ops.hip -- the GAN feature-matching loss of the HifiGAN training step as multi-tensor launches (round 6).
//
// Reference modules/hifigan/hifigan.py:328-335 (`feature_loss`): loss = 2 * sum over every (real, generated) feature-map pair of
// mean(|r - g|) -- 54 pairs per generator pass (5 period discriminators x 6 maps + 3 scale discriminators x 8), i.e. per pair and
// direction sub, abs, mean and an add onto the running total: ~430 stock launches of a few microseconds each on the step's
// critical stream.  Here: ONE launch sums |a - b| of up to 32 pairs (descriptors in the kernel-argument segment: no device table,
// no upload), one per-block partial each, and a one-workgroup launch finishes them in a fixed order (deterministic, no atomics);
// the backward is ONE launch per batch of pairs that writes  d b = scale_p * g * sign(b - a)  (and d a = - d b where asked).
// mode 1 terms (same launches): sum (a - target)^2 -- the LS-GAN terms `mean((1 - D(x))^2)`, `mean(D(G)^2)` of discriminator_loss /
// generator_loss (hifigan.py:338-365) over the 8 discriminators' outputs: ~60 stock launches per call and as many in backward.
#include "svb_common.h"
#include "../../include/svb_hip.h"

#define L1_CHUNK 4096               /* elements per workgroup: 256 threads x 4 float4 */

struct SvbL1Batch {
    SvbL1Pair p[SVB_L1_MAX_PAIRS];
    int n;
};

// block -> (pair, first element): pairs own consecutive block ranges starting at p[i].block0
__device__ __forceinline__ int l1_find_pair(const SvbL1Batch& b, int blk) {
    int i = 0;
    for (int k = 1; k < b.n; ++k)
        if (b.p[k].block0 <= blk) i = k;
    return i;
}

__global__ __launch_bounds__(256) void svb_l1_pairs_fwd_kernel(SvbL1Batch b, float* partials) {
    __shared__ float red[8];
    const int blk = blockIdx.x;
    const int i = l1_find_pair(b, blk);
    const float* pa = b.p[i].a;
    const float* pb = b.p[i].b;
    const long n = b.p[i].n;
    const long e0 = (long)(blk - b.p[i].block0) * L1_CHUNK;
    float s = 0.f;
    if (b.p[i].mode == 1) {                          // squared distance to a constant (LS-GAN term): short tensors, scalar loads
        const float c = b.p[i].target;
        for (int j = 0; j < 16; ++j) {
            const long e = e0 + j * 256 + threadIdx.x;
            if (e < n) { const float d = pa[e] - c; s = fmaf(d, d, s); }
        }
    } else if ((n & 3) == 0 && (((size_t)pa | (size_t)pb) & 15) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long e = e0 + (long)(j * 256 + threadIdx.x) * 4;
            if (e < n) {
                const float4 x = *reinterpret_cast<const float4*>(pa + e), y = *reinterpret_cast<const float4*>(pb + e);
                s += fabsf(x.x - y.x) + fabsf(x.y - y.y) + fabsf(x.z - y.z) + fabsf(x.w - y.w);
            }
        }
    } else {
        for (int j = 0; j < 16; ++j) {
            const long e = e0 + j * 256 + threadIdx.x;
            if (e < n) s += fabsf(pa[e] - pb[e]);
        }
    }
    s = svb_block_sum<256>(s, red);
    if (threadIdx.x == 0) partials[blk] = s;
}

// out[0] (+)= sum_p scale_p * (sum of pair p's partials, in block order): one workgroup, fixed order
__global__ __launch_bounds__(256) void svb_l1_pairs_final_kernel(SvbL1Batch b, const float* partials, int total_blocks, float* out,
                                                                int accumulate) {
    __shared__ float red[8];
    float total = 0.f;
    for (int i = 0; i < b.n; ++i) {
        const int b0 = b.p[i].block0, b1 = i + 1 < b.n ? b.p[i + 1].block0 : total_blocks;
        float s = 0.f;
        for (int k = b0 + threadIdx.x; k < b1; k += 256) s += partials[k];
        s = svb_block_sum<256>(s, red);
        total += s * b.p[i].scale;
    }
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + total : total;
}

__global__ __launch_bounds__(256) void svb_l1_pairs_bwd_kernel(SvbL1Batch b, const float* gout) {
    const int blk = blockIdx.x;
    const int i = l1_find_pair(b, blk);
    const float* pa = b.p[i].a;
    const float* pb = b.p[i].b;
    float* da = b.p[i].da;
    float* db = b.p[i].db;
    const long n = b.p[i].n;
    const long e0 = (long)(blk - b.p[i].block0) * L1_CHUNK;
    const float g = gout[0] * b.p[i].scale;
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };       // torch's abs backward: sign(0) = 0
    if (b.p[i].mode == 1) {
        const float c = b.p[i].target, g2 = 2.f * g;
        for (int j = 0; j < 16; ++j) {
            const long e = e0 + j * 256 + threadIdx.x;
            if (e < n && da) da[e] = g2 * (pa[e] - c);
        }
        return;
    }
    if ((n & 3) == 0 && (((size_t)pa | (size_t)pb | (size_t)da | (size_t)db) & 15) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long e = e0 + (long)(j * 256 + threadIdx.x) * 4;
            if (e < n) {
                const float4 x = *reinterpret_cast<const float4*>(pa + e), y = *reinterpret_cast<const float4*>(pb + e);
                const float4 d = make_float4(g * sgn(y.x - x.x), g * sgn(y.y - x.y), g * sgn(y.z - x.z), g * sgn(y.w - x.w));
                if (db) *reinterpret_cast<float4*>(db + e) = d;
                if (da) *reinterpret_cast<float4*>(da + e) = make_float4(-d.x, -d.y, -d.z, -d.w);
            }
        }
    } else {
        for (int j = 0; j < 16; ++j) {
            const long e = e0 + j * 256 + threadIdx.x;
            if (e < n) {
                const float d = g * sgn(pb[e] - pa[e]);
                if (db) db[e] = d;
                if (da) da[e] = -d;
            }
        }
    }
}

static int l1_fill(SvbL1Batch& b, const SvbL1Pair* pairs, int n) {
    if (!pairs || n <= 0 || n > SVB_L1_MAX_PAIRS) return -1;
    int blocks = 0;
    b.n = n;
    for (int i = 0; i < n; ++i) {
        if (!pairs[i].a || (!pairs[i].b && pairs[i].mode != 1) || pairs[i].n <= 0 || (pairs[i].mode != 0 && pairs[i].mode != 1)) return -1;
        b.p[i] = pairs[i];
        b.p[i].block0 = blocks;
        const long nb = (pairs[i].n + L1_CHUNK - 1) / L1_CHUNK;
        if (nb > (1L << 24) || blocks + nb > (1L << 30)) return -1;
        blocks += (int)nb;
    }
    return blocks;
}

extern "C" int svb_l1_pairs_blocks(const SvbL1Pair* pairs, int n) {
    SvbL1Batch b;
    return l1_fill(b, pairs, n);
}

extern "C" int svb_l1_pairs_fwd(const SvbL1Pair* pairs, int n, float* partials, float* out, int accumulate, void* stream) {
    SvbL1Batch b;
    const int blocks = l1_fill(b, pairs, n);
    if (blocks <= 0 || !partials || !out) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_l1_pairs_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b, partials);
    hipLaunchKernelGGL(svb_l1_pairs_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, b, partials, blocks, out, accumulate);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_l1_pairs_bwd(const SvbL1Pair* pairs, int n, const float* gout, void* stream) {
    SvbL1Batch b;
    const int blocks = l1_fill(b, pairs, n);
    if (blocks <= 0 || !gout) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_l1_pairs_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b, gout);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ---- out = scale * (a + b [+ c]): the HifiGAN generator's mean over its parallel ResBlocks (reference modules/hifigan/hifigan.py:157-163:
// `xs = xs + resblocks[...](x)` twice, then `x = xs / num_kernels`: three stock launches over the widest tensors of the step) in one pass.
__global__ __launch_bounds__(256) void svb_sum_scale_kernel(const float* a, const float* b, const float* c, float scale, float* out, long n4,
                                                            long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c) z = reinterpret_cast<const float4*>(c)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4((x.x + y.x + z.x) * scale, (x.y + y.y + z.y) * scale, (x.z + y.z + z.z) * scale,
                                                        (x.w + y.w + z.w) * scale);
    } else {
        const long e = 4 * n4 + (i - n4);
        if (e < n) out[e] = (a[e] + b[e] + (c ? c[e] : 0.f)) * scale;
    }
}

extern "C" int svb_sum_scale(const float* a, const float* b, const float* c, float scale, float* out, long n, void* stream) {
    if (!a || !b || !out || n <= 0) return SVB_ERR_ARG;
    const bool vec = ((((size_t)a | (size_t)b | (size_t)c | (size_t)out) & 15) == 0);
    const long n4 = vec ? n / 4 : 0, items = n4 + (n - 4 * n4);
    if ((items + 255) / 256 > (1L << 30)) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_sum_scale_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, scale, out, n4, n);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

