// wn_ops.hip -- HBM-bound pieces of the gated (WaveNet-style) conv stack and LayerNorm.
//   gate:     reference modules/fastspeech/fs2_vae.py:10-16 (fused_add_tanh_sigmoid_multiply) + :75-81 (cond slice)
//   res/skip: reference modules/fastspeech/fs2_vae.py:83-89
//   layernorm: torch.nn.LayerNorm as used by reference modules/fastspeech/conformer/layers.py:160-170
// All are pure streaming kernels: 16-byte vector loads when rows are 16-byte aligned, grid-stride loops,
// 64-wide wave reductions.  Roofline: HBM (bytes listed per kernel in DESIGN.md).
#include "svb_common.h"
#include "../../include/svb_hip.h"

extern "C" int svb_abi_version(void) { return SVB_ABI_VERSION; }

// acts[b,c,t] = tanh(xin[b,c,t]+g[b,goff+c,t]) * sigmoid(xin[b,C+c,t]+g[b,goff+C+c,t])
template <int VEC>
__global__ __launch_bounds__(256) void svb_wn_gate_fwd_kernel(const float* xin, const float* g, float* acts, int B, int C,
                                                              int T, int gch, int goff) {
    const int TV = T / VEC;
    const long total = (long)B * C * TV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int tv = (int)(i % TV);
        const long bc = i / TV;
        const int c = (int)(bc % C), b = (int)(bc / C);
        const size_t xa = ((size_t)b * 2 * C + c) * T + (size_t)tv * VEC;
        const size_t xb = xa + (size_t)C * T;
        const size_t ga = ((size_t)b * gch + goff + c) * T + (size_t)tv * VEC;
        const size_t gb = ga + (size_t)C * T;
        const size_t oa = ((size_t)b * C + c) * T + (size_t)tv * VEC;
        if (VEC == 4) {
            float4 a = *reinterpret_cast<const float4*>(xin + xa);
            float4 s = *reinterpret_cast<const float4*>(xin + xb);
            if (g) {
                const float4 ga4 = *reinterpret_cast<const float4*>(g + ga);
                const float4 gb4 = *reinterpret_cast<const float4*>(g + gb);
                a.x += ga4.x; a.y += ga4.y; a.z += ga4.z; a.w += ga4.w;
                s.x += gb4.x; s.y += gb4.y; s.z += gb4.z; s.w += gb4.w;
            }
            float4 o;
            o.x = tanhf(a.x) * svb_sigmoid(s.x);
            o.y = tanhf(a.y) * svb_sigmoid(s.y);
            o.z = tanhf(a.z) * svb_sigmoid(s.z);
            o.w = tanhf(a.w) * svb_sigmoid(s.w);
            *reinterpret_cast<float4*>(acts + oa) = o;
        } else {
            float a = xin[xa], s = xin[xb];
            if (g) { a += g[ga]; s += g[gb]; }
            acts[oa] = tanhf(a) * svb_sigmoid(s);
        }
    }
}

// dxin[:, :C] = dacts * sig * (1 - tanh^2) ; dxin[:, C:] = dacts * tanh * sig * (1 - sig); optional copy into dg slice
__global__ __launch_bounds__(256) void svb_wn_gate_bwd_kernel(const float* xin, const float* g, const float* dacts,
                                                              float* dxin, float* dg, int B, int C, int T, int gch,
                                                              int goff) {
    const long total = (long)B * C * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const long bc = i / T;
        const int c = (int)(bc % C), b = (int)(bc / C);
        const size_t xa = ((size_t)b * 2 * C + c) * T + t;
        const size_t xb = xa + (size_t)C * T;
        const size_t ga = ((size_t)b * gch + goff + c) * T + t;
        const size_t gb = ga + (size_t)C * T;
        float a = xin[xa], s = xin[xb];
        if (g) { a += g[ga]; s += g[gb]; }
        const float th = tanhf(a), sg = svb_sigmoid(s);
        const float d = dacts[((size_t)b * C + c) * T + t];
        const float da = d * sg * (1.f - th * th);
        const float ds = d * th * sg * (1.f - sg);
        if (dxin) { dxin[xa] = da; dxin[xb] = ds; }
        if (dg) { dg[ga] = da; dg[gb] = ds; }
    }
}

// x_new = (x + rs[:, :C]) * mask ; out_new = out + rs[:, C:]     (last: out_new = out + rs)
__global__ __launch_bounds__(256) void svb_wn_res_skip_kernel(const float* x, const float* rs, const float* mask,
                                                              const float* out, float* x_new, float* out_new, int B, int C,
                                                              int T, int last) {
    const long total = (long)B * C * T;
    const int rc = last ? C : 2 * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const long bc = i / T;
        const int c = (int)(bc % C), b = (int)(bc / C);
        const size_t r0 = ((size_t)b * rc + c) * T + t;
        const float o = out ? out[i] : 0.f;
        if (last) {
            out_new[i] = o + rs[r0];
        } else {
            const float m = mask ? mask[(size_t)b * T + t] : 1.f;
            x_new[i] = (x[i] + rs[r0]) * m;
            out_new[i] = o + rs[r0 + (size_t)C * T];
        }
    }
}

// backward of res/skip: drs[:, :C] = dx_new * mask ; drs[:, C:] = dout ; dxm = dx_new * mask (contiguous copy)
__global__ __launch_bounds__(256) void svb_wn_res_skip_bwd_kernel(const float* dx_new, const float* dout, const float* mask,
                                                                  float* drs, float* dxm, int B, int C, int T) {
    const long total = (long)B * C * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const long bc = i / T;
        const int c = (int)(bc % C), b = (int)(bc / C);
        const size_t r0 = ((size_t)b * 2 * C + c) * T + t;
        const float m = mask ? mask[(size_t)b * T + t] : 1.f;
        const float v = dx_new ? dx_new[i] * m : 0.f;
        drs[r0] = v;
        drs[r0 + (size_t)C * T] = dout[i];
        if (dxm) dxm[i] = v;
    }
}

// ---- LayerNorm: one 64-lane wave per row ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void svb_layernorm_fwd_kernel(const float* x, const float* gamma, const float* beta,
                                                                float* y, float* mean, float* rstd, int rows, int C,
                                                                float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        const float mu = svb_wave_sum(s) / (float)C;
        float v = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; v += d * d; }
        const float rs = 1.f / sqrtf(svb_wave_sum(v) / (float)C + eps);
        float* yr = y + (size_t)row * C;
        for (int c = lane; c < C; c += 64) {
            float o = (xr[c] - mu) * rs;
            if (gamma) o *= gamma[c];
            if (beta) o += beta[c];
            yr[c] = o;
        }
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
    }
}

// dx = rstd * (dy*gamma - mean_c(dy*gamma) - xhat * mean_c(dy*gamma*xhat)); dgamma/dbeta partials per block
__global__ __launch_bounds__(256) void svb_layernorm_bwd_kernel(const float* x, const float* gamma, const float* dy,
                                                                const float* mean, const float* rstd, float* dx,
                                                                float* dgamma_part, float* dbeta_part, int rows, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-lane running sums for channels lane, lane+64, ... (C <= 1024)
    float dgs[16], dbs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { dgs[j] = 0.f; dbs[j] = 0.f; }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float* xr = x + (size_t)row * C;
        const float* dr = dy + (size_t)row * C;
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = lane + 64 * j;
            if (c < C) {
                const float xh = (xr[c] - mu) * rs;
                const float gd = dr[c] * (gamma ? gamma[c] : 1.f);
                s1 += gd;
                s2 += gd * xh;
                dgs[j] += dr[c] * xh;
                dbs[j] += dr[c];
            }
        }
        s1 = svb_wave_sum(s1) / (float)C;
        s2 = svb_wave_sum(s2) / (float)C;
        if (dx) {
            float* dxr = dx + (size_t)row * C;
            for (int c = lane; c < C; c += 64) {
                const float xh = (xr[c] - mu) * rs;
                const float gd = dr[c] * (gamma ? gamma[c] : 1.f);
                dxr[c] = rs * (gd - s1 - xh * s2);
            }
        }
    }
    __shared__ float red[2][4][1024];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = lane + 64 * j;
        if (c < C) { red[0][wave][c] = dgs[j]; red[1][wave][c] = dbs[j]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        if (dgamma_part) dgamma_part[(size_t)blockIdx.x * C + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        if (dbeta_part) dbeta_part[(size_t)blockIdx.x * C + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    }
}


// LayerNorm over the channel dim of an NCT tensor.  Block = 32 time columns x 8 channel groups (a wave covers two
// 128-byte row segments per load); each thread keeps its C/8 values in registers (all loads in flight at once, x is
// read ONCE), the groups combine their shifted moments through LDS.  1 read + 1 write of x.
#define SVB_LN_MAXV 48          /* register-resident values per thread: C <= 384 (larger C re-reads x) */
__global__ __launch_bounds__(256) void svb_layernorm_nct_fwd_kernel(const float* x, const float* gamma, const float* beta,
                                                                    float* y, int B, int C, int T, float eps) {
    __shared__ float s_sum[8][32];
    __shared__ float s_sq[8][32];
    const int tl = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int tiles = (T + 31) / 32;
    const int b = blockIdx.x / tiles, t = (blockIdx.x % tiles) * 32 + tl;
    const bool ok = t < T;
    const float* xc = x + (size_t)b * C * T + (ok ? t : 0);
    float* yc = y + (size_t)b * C * T + (ok ? t : 0);
    const float x0 = xc[0];                                   // shift (any finite value of the column works)
    const bool inreg = C <= 8 * SVB_LN_MAXV;
    float v[SVB_LN_MAXV];
    float s = 0.f, ss = 0.f;
    if (inreg) {
#pragma unroll
        for (int i = 0; i < SVB_LN_MAXV; ++i) {
            const int c = cg + 8 * i;
            v[i] = c < C ? xc[(size_t)c * T] : x0;
        }
#pragma unroll
        for (int i = 0; i < SVB_LN_MAXV; ++i) {
            const float d = v[i] - x0;                        // channels beyond C contribute d = 0
            s += d;
            ss = fmaf(d, d, ss);
        }
    } else {
        for (int c = cg; c < C; c += 8) {
            const float d = xc[(size_t)c * T] - x0;
            s += d;
            ss = fmaf(d, d, ss);
        }
    }
    s_sum[cg][tl] = s;
    s_sq[cg][tl] = ss;
    __syncthreads();
    s = 0.f; ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { s += s_sum[q][tl]; ss += s_sq[q][tl]; }
    const float md = s / (float)C;
    const float var = fmaxf(ss / (float)C - md * md, 0.f);
    const float mu = x0 + md;
    const float rs = 1.f / sqrtf(var + eps);
    if (!ok) return;
    if (inreg) {
#pragma unroll
        for (int i = 0; i < SVB_LN_MAXV; ++i) {
            const int c = cg + 8 * i;
            if (c < C) {
                float o = (v[i] - mu) * rs;
                if (gamma) o *= gamma[c];
                if (beta) o += beta[c];
                yc[(size_t)c * T] = o;
            }
        }
    } else {
        for (int c = cg; c < C; c += 8) {
            float o = (xc[(size_t)c * T] - mu) * rs;
            if (gamma) o *= gamma[c];
            if (beta) o += beta[c];
            yc[(size_t)c * T] = o;
        }
    }
}

static inline int ew_grid(long total) {
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int svb_wn_gate_fwd(const float* xin, const float* g, float* acts, int B, int C, int T, int g_channels,
                               int g_off, void* stream) {
    if (!xin || !acts || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    if (g && (g_off < 0 || g_off + 2 * C > g_channels)) return SVB_ERR_ARG;
    const bool v4 = (T % 4 == 0) && (((uintptr_t)xin | (uintptr_t)g | (uintptr_t)acts) % 16 == 0);
    if (v4)
        hipLaunchKernelGGL(svb_wn_gate_fwd_kernel<4>, dim3(ew_grid((long)B * C * (T / 4))), dim3(256), 0,
                           (hipStream_t)stream, xin, g, acts, B, C, T, g_channels, g_off);
    else
        hipLaunchKernelGGL(svb_wn_gate_fwd_kernel<1>, dim3(ew_grid((long)B * C * T)), dim3(256), 0, (hipStream_t)stream,
                           xin, g, acts, B, C, T, g_channels, g_off);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_wn_gate_bwd(const float* xin, const float* g, const float* dacts, float* dxin, float* dg, int B, int C,
                               int T, int g_channels, int g_off, void* stream) {
    if (!xin || !dacts || (!dxin && !dg) || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    if ((g || dg) && (g_off < 0 || g_off + 2 * C > g_channels)) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_wn_gate_bwd_kernel, dim3(ew_grid((long)B * C * T)), dim3(256), 0, (hipStream_t)stream, xin, g,
                       dacts, dxin, dg, B, C, T, g_channels, g_off);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_wn_res_skip(const float* x, const float* rs, const float* mask, const float* out, float* x_new,
                               float* out_new, int B, int C, int T, int last, void* stream) {
    if (!rs || !out_new || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    if (!last && (!x || !x_new)) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_wn_res_skip_kernel, dim3(ew_grid((long)B * C * T)), dim3(256), 0, (hipStream_t)stream, x, rs,
                       mask, out, x_new, out_new, B, C, T, last);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_wn_res_skip_bwd(const float* dx_new, const float* dout, const float* mask, float* drs, float* dxm,
                                   int B, int C, int T, void* stream) {
    if (!dout || !drs || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_wn_res_skip_bwd_kernel, dim3(ew_grid((long)B * C * T)), dim3(256), 0, (hipStream_t)stream,
                       dx_new, dout, mask, drs, dxm, B, C, T);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                 float* rstd, int rows, int C, float eps, void* stream) {
    if (!x || !y || rows <= 0 || C <= 0) return SVB_ERR_ARG;
    int grid = (rows + 3) / 4;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(svb_layernorm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean,
                       rstd, rows, C, eps);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* mean, const float* rstd,
                                 float* dx, float* dgamma_part, float* dbeta_part, int rows, int C, int n_part,
                                 void* stream) {
    if (!x || !dy || !mean || !rstd || rows <= 0 || C <= 0 || C > 1024 || n_part <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_layernorm_bwd_kernel, dim3(n_part), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, mean, rstd,
                       dx, dgamma_part, dbeta_part, rows, C);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_layernorm_nct_fwd(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T,
                                     float eps, void* stream) {
    if (!x || !y || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_layernorm_nct_fwd_kernel, dim3(B * ((T + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                       beta, y, B, C, T, eps);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
