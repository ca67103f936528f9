// vae_ops.hip -- small fused kernels of the VAE around the convolutions (gfx950).
//
// 1. Latent head of GlobalFVAEEncoder / TMPFVAE (reference modules/voice_conversion/vae_models.py:24-41,100-105):
//    time mean of the pooled features, split into (m_q, logs_q), re-parameterised draw z = m_q + eps * exp(logs_q), the
//    positivity guard on logs_q, KL(N(m, e^logs) || N(0,1)) and its masked mean per stacked call -- ~22 stock launches on
//    [N,128,1] tensors forward and ~20 backward become one call per direction.
// 2. GroupNorm + ReLU + residual of the pitch encoder's ConvBlock (reference modules/commons/common_layers.py:739-773 as
//    used by ConvStacks :688-707, res=True): y = x_res + relu(GroupNorm(h)), forward and backward, one workgroup per
//    (clip, group) -- the group's 16 x T slab is contiguous in the [B,C,T] layout.
// All reductions have a fixed order (deterministic, no atomics).
#include "svb_common.h"
#include "../../include/svb_hip.h"

// ------------------------------------------------------------------------------------------------------------------
// Latent head.  xp [N][2L][Tp] pooled features; eps [N][L]; mask [N][Tq] (x_mask_sqz); groups equal slices of N.
//   m = mean_t xp[n][l], lg = mean_t xp[n][L+l];  z = m + eps * exp(lg);  bad = !(exp(lg) > 0);  lgg = bad ? 0 : lg
//   kl_nl = 0.5 (exp(2 lgg) + m^2 - 1) - lgg
//   kl[g] = sum_{n in g} (sum_l kl_nl) (sum_t mask[n]) / (sum_{n in g} sum_t mask[n]) / L
// (the reference multiplies kl [N,L,1] with the mask [N,1,Tq] and sums the broadcast product, vae_models.py:36-38)
// stat [N][2] = (sum_l kl_nl, sum_t mask[n]) is kept for the backward pass.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void svb_vae_head_fwd_kernel(const float* xp, const float* eps, const float* mask, float* z,
                                                               float* mq, float* lq, float* stat, int L, int Tp, int Tq) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    float klsum = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) {
        const float* r0 = xp + ((size_t)n * 2 * L + l) * Tp;
        const float* r1 = xp + ((size_t)n * 2 * L + L + l) * Tp;
        float a = 0.f, b = 0.f;
        for (int t = 0; t < Tp; ++t) { a += r0[t]; b += r1[t]; }
        const float m = a / (float)Tp, lg = b / (float)Tp;
        const float e = expf(lg);
        const bool bad = !(e > 0.f);
        const float lgg = bad ? 0.f : lg;
        const size_t o = (size_t)n * L + l;
        z[o] = m + eps[o] * e;
        mq[o] = m;
        lq[o] = lgg;
        klsum += 0.5f * (expf(2.f * lgg) + m * m - 1.f) - lgg;
    }
    klsum = svb_block_sum<256>(klsum, red);
    float ms = 0.f;
    for (int t = threadIdx.x; t < Tq; t += 256) ms += mask[(size_t)n * Tq + t];
    ms = svb_block_sum<256>(ms, red);
    if (threadIdx.x == 0) { stat[2 * n] = klsum; stat[2 * n + 1] = ms; }
}

// kl[g] = sum_n stat[n][0] stat[n][1] / sum_n stat[n][1] / L   (one small workgroup; fixed order)
__global__ __launch_bounds__(64) void svb_vae_head_kl_kernel(const float* stat, float* kl, int N, int groups, int L) {
    const int g = threadIdx.x;
    if (g >= groups) return;
    const int per = N / groups;
    float num = 0.f, den = 0.f;
    for (int n = g * per; n < (g + 1) * per; ++n) { num += stat[2 * n] * stat[2 * n + 1]; den += stat[2 * n + 1]; }
    kl[g] = num / den / (float)L;
}

// dxp from the cotangents of (z, m_q, logs_q, kl); any of gz / gm / glq may be null (no gradient)
__global__ __launch_bounds__(256) void svb_vae_head_bwd_kernel(const float* xp, const float* eps, const float* stat,
                                                               const float* gz, const float* gm, const float* glq,
                                                               const float* gkl, float* dxp, int N, int groups, int L, int Tp) {
    __shared__ float s_c;
    const int n = blockIdx.x;
    if (threadIdx.x == 0) {
        const int per = N / groups, g = n / per;
        float den = 0.f;
        for (int q = g * per; q < (g + 1) * per; ++q) den += stat[2 * q + 1];
        s_c = gkl ? gkl[g] * stat[2 * n + 1] / den / (float)L : 0.f;
    }
    __syncthreads();
    const float c = s_c;
    for (int l = threadIdx.x; l < L; l += 256) {
        const float* r0 = xp + ((size_t)n * 2 * L + l) * Tp;
        const float* r1 = xp + ((size_t)n * 2 * L + L + l) * Tp;
        float a = 0.f, b = 0.f;
        for (int t = 0; t < Tp; ++t) { a += r0[t]; b += r1[t]; }
        const float m = a / (float)Tp, lg = b / (float)Tp;
        const float e = expf(lg);
        const bool bad = !(e > 0.f);
        const size_t o = (size_t)n * L + l;
        const float vz = gz ? gz[o] : 0.f;
        float dm = vz + (gm ? gm[o] : 0.f) + c * m;
        float dl = vz * eps[o] * e;
        if (!bad) dl += (glq ? glq[o] : 0.f) + c * (expf(2.f * lg) - 1.f);
        dm /= (float)Tp;
        dl /= (float)Tp;
        float* d0 = dxp + ((size_t)n * 2 * L + l) * Tp;
        float* d1 = dxp + ((size_t)n * 2 * L + L + l) * Tp;
        for (int t = 0; t < Tp; ++t) { d0[t] = dm; d1[t] = dl; }
    }
}

extern "C" int svb_vae_head_fwd(const float* xp, const float* eps, const float* mask, float* z, float* mq, float* lq, float* kl,
                                float* stat, int N, int groups, int L, int Tp, int Tq, void* stream) {
    if (!xp || !eps || !mask || !z || !mq || !lq || !kl || !stat || N <= 0 || groups <= 0 || groups > 64 || N % groups || L <= 0 ||
        Tp <= 0 || Tq <= 0)
        return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_vae_head_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, xp, eps, mask, z, mq, lq, stat, L, Tp,
                       Tq);
    hipLaunchKernelGGL(svb_vae_head_kl_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)stat, kl, N, groups, L);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_vae_head_bwd(const float* xp, const float* eps, const float* stat, const float* gz, const float* gm,
                                const float* glq, const float* gkl, float* dxp, int N, int groups, int L, int Tp, void* stream) {
    if (!xp || !eps || !stat || !dxp || N <= 0 || groups <= 0 || groups > 64 || N % groups || L <= 0 || Tp <= 0)
        return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_vae_head_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, xp, eps, stat, gz, gm, glq, gkl, dxp,
                       N, groups, L, Tp);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm + ReLU (+ residual).  h, res, y: [B][C][T] contiguous; groups of cg = C / G channels; one workgroup per
// (clip, group), whose slab of cg*T values is contiguous.  Two-pass statistics (mean, then centred variance, biased) as
// torch's native_group_norm.  stats [B*G][2] = (mean, rstd).
//   forward : y = (res ? res : 0) + relu((h - mean) rstd gamma[c] + beta[c])
//   backward: gy -> dh (GroupNorm backward through the ReLU gate), per-(clip, channel) partial sums dgb [2][B][C]
//             (sum_t g xhat, sum_t g) that the host sums over clips (dgamma, dbeta); d res = gy (the caller's, no kernel).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void svb_gn_relu_fwd_kernel(const float* h, const float* res, const float* gamma,
                                                              const float* beta, float* y, float* stats, int C, int T, int G,
                                                              float eps) {
    __shared__ float red[4];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cg = C / G, n = cg * T;
    const size_t base = ((size_t)b * C + (size_t)g * cg) * T;
    const float* hp = h + base;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += hp[i];
    const float mean = svb_block_sum<256>(s, red) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float d = hp[i] - mean; q += d * d; }
    const float var = svb_block_sum<256>(q, red) / (float)n;
    const float rstd = 1.f / sqrtf(var + eps);
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = g * cg + i / T;
        float v = (hp[i] - mean) * rstd * gamma[c] + beta[c];
        v = v > 0.f ? v : 0.f;
        y[base + i] = res ? res[base + i] + v : v;
    }
}

__global__ __launch_bounds__(256) void svb_gn_relu_bwd_kernel(const float* gy, const float* h, const float* gamma,
                                                              const float* beta, const float* stats, float* dh, float* dgb,
                                                              int B, int C, int T, int G) {
    __shared__ float red[4];
    __shared__ float ch_a[64], ch_b[64];           // per-channel sums of this group (cg <= 64)
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cg = C / G, n = cg * T;
    const size_t base = ((size_t)b * C + (size_t)g * cg) * T;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    const float* hp = h + base;
    const float* gp = gy + base;
    // per channel: A_c = sum_t g xhat, B_c = sum_t g  (g = gy gated by the ReLU); one wave per channel, round-robin
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int cl = wave; cl < cg; cl += 4) {
        const int c = g * cg + cl;
        const float gm = gamma[c], bt = beta[c];
        float a = 0.f, bb = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float xh = (hp[cl * T + t] - mean) * rstd;
            const float gg = (xh * gm + bt) > 0.f ? gp[cl * T + t] : 0.f;
            a += gg * xh;
            bb += gg;
        }
        a = svb_wave_sum(a);
        bb = svb_wave_sum(bb);
        if (lane == 0) {
            ch_a[cl] = a; ch_b[cl] = bb;
            dgb[(size_t)b * C + c] = a;
            dgb[(size_t)B * C + (size_t)b * C + c] = bb;
        }
    }
    __syncthreads();
    // group sums of the gradient w.r.t. xhat: S1 = sum_c gamma_c B_c, S2 = sum_c gamma_c A_c
    float s1 = 0.f, s2 = 0.f;
    for (int cl = 0; cl < cg; ++cl) { const float gm = gamma[g * cg + cl]; s1 += gm * ch_b[cl]; s2 += gm * ch_a[cl]; }
    const float inv_n = 1.f / (float)n;
    (void)red;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int cl = i / T, c = g * cg + cl;
        const float gm = gamma[c];
        const float xh = (hp[i] - mean) * rstd;
        const float gg = (xh * gm + beta[c]) > 0.f ? gp[i] : 0.f;
        dh[base + i] = rstd * (gg * gm - inv_n * (s1 + xh * s2));
    }
}

// Row-resident forms (round 3): one WAVE per channel of the group, the channel's row held in registers as KR 16-byte quads per
// lane -- h (and gy) are read from HBM exactly once, every load of a row is in flight at once, the channel's gamma / beta are
// wave-uniform scalars.  The streaming kernels above (scalar dependent-latency loops over a 72 KB slab, 2 workgroups per CU)
// ran at 0.7-1.2 TB/s: 88 / 91 us per call at [32, 256, 1124], G = 16 (tools/ewbench.py).  Need cg <= 16, T % 4 == 0,
// T / 4 <= 64 * KR and 16-byte aligned tensors; other shapes keep the kernels above.
template <int KR>
__global__ __launch_bounds__(1024) void svb_gn_relu_fwd_rows_kernel(const float* h, const float* res, const float* gamma,
                                                                   const float* beta, float* y, float* stats, int C, int T,
                                                                   int G, float eps) {
    __shared__ float red[16];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cg = C / G, n = cg * T, T4 = T >> 2;
    const int lane = threadIdx.x & 63, cl = threadIdx.x >> 6;          // blockDim = 64 * cg
    const int c = g * cg + cl;
    const size_t row = ((size_t)b * C + c) * T;
    const float4* hp = reinterpret_cast<const float4*>(h + row);
    float4 v[KR];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int i = lane + 64 * k;
        v[k] = i < T4 ? hp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    s = svb_wave_sum(s);
    if (lane == 0) red[cl] = s;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < cg; ++w) tot += red[w];
    const float mean = tot / (float)n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        if (lane + 64 * k < T4) {
            const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    q = svb_wave_sum(q);
    __syncthreads();
    if (lane == 0) red[cl] = q;
    __syncthreads();
    tot = 0.f;
    for (int w = 0; w < cg; ++w) tot += red[w];
    const float rstd = 1.f / sqrtf(tot / (float)n + eps);
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
    const float gm = gamma[c], bt = beta[c];
    const float4* rp = res ? reinterpret_cast<const float4*>(res + row) : nullptr;
    float4* yp = reinterpret_cast<float4*>(y + row);
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int i = lane + 64 * k;
        if (i < T4) {
            float4 o;
            o.x = fmaxf((v[k].x - mean) * rstd * gm + bt, 0.f);
            o.y = fmaxf((v[k].y - mean) * rstd * gm + bt, 0.f);
            o.z = fmaxf((v[k].z - mean) * rstd * gm + bt, 0.f);
            o.w = fmaxf((v[k].w - mean) * rstd * gm + bt, 0.f);
            if (rp) { const float4 r = rp[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
            yp[i] = o;
        }
    }
}

template <int KR>
__global__ __launch_bounds__(1024) void svb_gn_relu_bwd_rows_kernel(const float* gy, const float* h, const float* gamma,
                                                                   const float* beta, const float* stats, float* dh, float* dgb,
                                                                   int B, int C, int T, int G) {
    __shared__ float ch_a[16], ch_b[16];
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cg = C / G, n = cg * T, T4 = T >> 2;
    const int lane = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const int c = g * cg + cl;
    const size_t row = ((size_t)b * C + c) * T;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    const float gm = gamma[c], bt = beta[c];
    const float4* hp = reinterpret_cast<const float4*>(h + row);
    const float4* gp = reinterpret_cast<const float4*>(gy + row);
    float4 xh[KR], gg[KR];                     // xhat and the ReLU-gated gradient of this lane's quads
    float a = 0.f, bb = 0.f;
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int i = lane + 64 * k;
        const bool ok = i < T4;
        const float4 hv = ok ? hp[i] : make_float4(mean, mean, mean, mean);
        const float4 gv = ok ? gp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        xh[k] = make_float4((hv.x - mean) * rstd, (hv.y - mean) * rstd, (hv.z - mean) * rstd, (hv.w - mean) * rstd);
        gg[k].x = (xh[k].x * gm + bt) > 0.f ? gv.x : 0.f;
        gg[k].y = (xh[k].y * gm + bt) > 0.f ? gv.y : 0.f;
        gg[k].z = (xh[k].z * gm + bt) > 0.f ? gv.z : 0.f;
        gg[k].w = (xh[k].w * gm + bt) > 0.f ? gv.w : 0.f;
        a += (gg[k].x * xh[k].x + gg[k].y * xh[k].y) + (gg[k].z * xh[k].z + gg[k].w * xh[k].w);
        bb += (gg[k].x + gg[k].y) + (gg[k].z + gg[k].w);
    }
    a = svb_wave_sum(a);
    bb = svb_wave_sum(bb);
    if (lane == 0) {
        ch_a[cl] = a; ch_b[cl] = bb;
        dgb[(size_t)b * C + c] = a;
        dgb[(size_t)B * C + (size_t)b * C + c] = bb;
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    for (int w = 0; w < cg; ++w) { const float gw = gamma[g * cg + w]; s1 += gw * ch_b[w]; s2 += gw * ch_a[w]; }
    const float inv_n = 1.f / (float)n;
    float4* dp = reinterpret_cast<float4*>(dh + row);
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int i = lane + 64 * k;
        if (i < T4) {
            float4 o;
            o.x = rstd * (gg[k].x * gm - inv_n * (s1 + xh[k].x * s2));
            o.y = rstd * (gg[k].y * gm - inv_n * (s1 + xh[k].y * s2));
            o.z = rstd * (gg[k].z * gm - inv_n * (s1 + xh[k].z * s2));
            o.w = rstd * (gg[k].w * gm - inv_n * (s1 + xh[k].w * s2));
            dp[i] = o;
        }
    }
}

static const bool g_svb_gn_rows_off = SVB_ENV_FLAG("SVB_GN_NO_ROWS");       // A/B switch
// KR of the row-resident kernels for this shape, 0 = not eligible
static int svb_gn_rows_kr(int C, int T, int G, const void* p0, const void* p1, const void* p2) {
    if (g_svb_gn_rows_off || C / G > 16 || (T & 3)) return 0;
    if ((((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) != 0) return 0;
    const int need = ((T >> 2) + 63) / 64;
    return need <= 2 ? 2 : (need <= 5 ? 5 : (need <= 8 ? 8 : 0));
}

extern "C" int svb_gn_relu_fwd(const float* h, const float* res, const float* gamma, const float* beta, float* y, float* stats,
                               int B, int C, int T, int G, float eps, void* stream) {
    if (!h || !gamma || !beta || !y || !stats || B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G || C / G > 64) return SVB_ERR_ARG;
    const int kr = svb_gn_rows_kr(C, T, G, h, res, y);
    const dim3 rows_block(64 * (C / G));
    if (kr == 2)
        hipLaunchKernelGGL(svb_gn_relu_fwd_rows_kernel<2>, dim3(B * G), rows_block, 0, (hipStream_t)stream, h, res, gamma, beta, y, stats, C, T, G, eps);
    else if (kr == 5)
        hipLaunchKernelGGL(svb_gn_relu_fwd_rows_kernel<5>, dim3(B * G), rows_block, 0, (hipStream_t)stream, h, res, gamma, beta, y, stats, C, T, G, eps);
    else if (kr == 8)
        hipLaunchKernelGGL(svb_gn_relu_fwd_rows_kernel<8>, dim3(B * G), rows_block, 0, (hipStream_t)stream, h, res, gamma, beta, y, stats, C, T, G, eps);
    else
        hipLaunchKernelGGL(svb_gn_relu_fwd_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, h, res, gamma, beta, y, stats, C, T,
                           G, eps);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_gn_relu_bwd(const float* gy, const float* h, const float* gamma, const float* beta, const float* stats,
                               float* dh, float* dgb, int B, int C, int T, int G, void* stream) {
    if (!gy || !h || !gamma || !beta || !stats || !dh || !dgb || B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G || C / G > 64)
        return SVB_ERR_ARG;
    const int kr = svb_gn_rows_kr(C, T, G, gy, h, dh);
    const dim3 rows_block(64 * (C / G));
    if (kr == 2)
        hipLaunchKernelGGL(svb_gn_relu_bwd_rows_kernel<2>, dim3(B * G), rows_block, 0, (hipStream_t)stream, gy, h, gamma, beta, stats, dh, dgb, B, C, T, G);
    else if (kr == 5)
        hipLaunchKernelGGL(svb_gn_relu_bwd_rows_kernel<5>, dim3(B * G), rows_block, 0, (hipStream_t)stream, gy, h, gamma, beta, stats, dh, dgb, B, C, T, G);
    else if (kr == 8)
        hipLaunchKernelGGL(svb_gn_relu_bwd_rows_kernel<8>, dim3(B * G), rows_block, 0, (hipStream_t)stream, gy, h, gamma, beta, stats, dh, dgb, B, C, T, G);
    else
        hipLaunchKernelGGL(svb_gn_relu_bwd_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, gy, h, gamma, beta, stats, dh, dgb,
                           B, C, T, G);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
