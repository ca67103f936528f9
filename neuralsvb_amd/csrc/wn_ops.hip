// wn_ops.hip -- HBM-bound pieces of the gated (WaveNet-style) conv stack and LayerNorm.
//   gate:     reference modules/fastspeech/fs2_vae.py:10-16 (fused_add_tanh_sigmoid_multiply) + :75-81 (cond slice)
//   res/skip: reference modules/fastspeech/fs2_vae.py:83-89
//   layernorm: torch.nn.LayerNorm as used by reference modules/fastspeech/conformer/layers.py:160-170
// All are pure streaming kernels: 16-byte vector loads when rows are 16-byte aligned, grid-stride loops,
// 64-wide wave reductions.  Roofline: HBM (bytes listed per kernel in DESIGN.md).
#include "svb_common.h"
#include <initializer_list>
#include "svb_q.h"
#include "../../include/svb_hip.h"

extern "C" int svb_abi_version(void) { return SVB_ABI_VERSION; }

// All four elementwise kernels of the gated stack (and svb_split_q) share one thread mapping: a thread owns VEC consecutive
// positions of one (batch, 16-channel chunk) and walks the chunk's channels two at a time.  Loads / fp32 stores are
// coalesced along t across the lanes of a wave (16-byte vectors when VEC == 4); the optional Q image (svb_q.h) of the
// result leaves as VEC contiguous 64-byte rows per thread -- the consuming bf16x3 conv stages it with plain copies.
#define SVB_EW_LOOP(total_)                                                                              \
    for (long i_ = (long)blockIdx.x * 256 + threadIdx.x; i_ < (total_); i_ += (long)gridDim.x * 256)

// acts[b,c,t] = tanh(xin[b,c,t]+g[b,goff+c,t]) * sigmoid(xin[b,C+c,t]+g[b,goff+C+c,t])
template <int VEC>
__global__ __launch_bounds__(256) void svb_wn_gate_fwd_kernel(const float* xin, const float* g, float* acts,
                                                              unsigned short* acts_q, int B, int C, int T, int gch, int goff) {
    const int TV = T / VEC, kc = (C + 15) / 16;
    SVB_EW_LOOP((long)B * kc * TV) {
        const int tv = (int)(i_ % TV);
        const long bk = i_ / TV;
        const int k = (int)(bk % kc), b = (int)(bk / kc);
        const size_t t0 = (size_t)tv * VEC;
        SvbQRows<VEC> q;
        q.zero();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o[2][VEC];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = k * 16 + 2 * e + h;
#pragma unroll
                for (int p = 0; p < VEC; ++p) o[h][p] = 0.f;
                if (c < C) {
                    const size_t xa = ((size_t)b * 2 * C + c) * T + t0, xb = xa + (size_t)C * T;
                    float a[VEC], sv[VEC];
                    svbq_ldv(xin + xa, a);
                    svbq_ldv(xin + xb, sv);
                    if (g) {
                        const size_t ga = ((size_t)b * gch + goff + c) * T + t0, gb = ga + (size_t)C * T;
                        float ag[VEC], sg[VEC];
                        svbq_ldv(g + ga, ag);
                        svbq_ldv(g + gb, sg);
#pragma unroll
                        for (int p = 0; p < VEC; ++p) { a[p] += ag[p]; sv[p] += sg[p]; }
                    }
#pragma unroll
                    for (int p = 0; p < VEC; ++p) o[h][p] = tanhf(a[p]) * svb_sigmoid(sv[p]);
                    svbq_stv(acts + ((size_t)b * C + c) * T + t0, o[h]);
                }
            }
            if (acts_q) {
#pragma unroll
                for (int p = 0; p < VEC; ++p) q.put(p, e, o[0][p], o[1][p]);
            }
        }
        if (acts_q) q.store(acts_q, ((size_t)b * kc + k) * T + t0);
    }
}

// dxin[:, :C] = dacts * sig * (1 - tanh^2) ; dxin[:, C:] = dacts * tanh * sig * (1 - sig); optional copy into dg slice;
// optional Q image of dxin (2C channels; needs C % 16 == 0 so that both halves start on a chunk boundary)
template <int VEC>
__global__ __launch_bounds__(256) void svb_wn_gate_bwd_kernel(const float* xin, const float* g, const float* dacts,
                                                              float* dxin, float* dg, unsigned short* dxin_q, int B, int C,
                                                              int T, int gch, int goff) {
    const int TV = T / VEC, kc = (C + 15) / 16;
    SVB_EW_LOOP((long)B * kc * TV) {
        const int tv = (int)(i_ % TV);
        const long bk = i_ / TV;
        const int k = (int)(bk % kc), b = (int)(bk / kc);
        const size_t t0 = (size_t)tv * VEC;
        SvbQRows<VEC> qa, qs;
        qa.zero();
        qs.zero();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float da[2][VEC], ds[2][VEC];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = k * 16 + 2 * e + h;
#pragma unroll
                for (int p = 0; p < VEC; ++p) { da[h][p] = 0.f; ds[h][p] = 0.f; }
                if (c < C) {
                    const size_t xa = ((size_t)b * 2 * C + c) * T + t0, xb = xa + (size_t)C * T;
                    const size_t ga = ((size_t)b * gch + goff + c) * T + t0, gb = ga + (size_t)C * T;
                    float a[VEC], sv[VEC], d[VEC];
                    svbq_ldv(xin + xa, a);
                    svbq_ldv(xin + xb, sv);
                    svbq_ldv(dacts + ((size_t)b * C + c) * T + t0, d);
                    if (g) {
                        float ag[VEC], sg[VEC];
                        svbq_ldv(g + ga, ag);
                        svbq_ldv(g + gb, sg);
#pragma unroll
                        for (int p = 0; p < VEC; ++p) { a[p] += ag[p]; sv[p] += sg[p]; }
                    }
#pragma unroll
                    for (int p = 0; p < VEC; ++p) {
                        const float th = tanhf(a[p]), sg = svb_sigmoid(sv[p]);
                        da[h][p] = d[p] * sg * (1.f - th * th);
                        ds[h][p] = d[p] * th * sg * (1.f - sg);
                    }
                    if (dxin) { svbq_stv(dxin + xa, da[h]); svbq_stv(dxin + xb, ds[h]); }
                    if (dg) { svbq_stv(dg + ga, da[h]); svbq_stv(dg + gb, ds[h]); }
                }
            }
            if (dxin_q) {
#pragma unroll
                for (int p = 0; p < VEC; ++p) { qa.put(p, e, da[0][p], da[1][p]); qs.put(p, e, ds[0][p], ds[1][p]); }
            }
        }
        if (dxin_q) {
            qa.store(dxin_q, ((size_t)b * 2 * kc + k) * T + t0);
            qs.store(dxin_q, ((size_t)b * 2 * kc + kc + k) * T + t0);
        }
    }
}

// x_new = (x + rs[:, :C]) * mask ; out_new = out + rs[:, C:]     (last: out_new = out + rs); optional Q image of x_new
template <int VEC>
__global__ __launch_bounds__(256) void svb_wn_res_skip_kernel(const float* x, const float* rs, const float* mask,
                                                              const float* out, float* x_new, float* out_new,
                                                              unsigned short* x_new_q, int B, int C, int T, int last) {
    const int TV = T / VEC, kc = (C + 15) / 16;
    const int rc = last ? C : 2 * C;
    SVB_EW_LOOP((long)B * kc * TV) {
        const int tv = (int)(i_ % TV);
        const long bk = i_ / TV;
        const int k = (int)(bk % kc), b = (int)(bk / kc);
        const size_t t0 = (size_t)tv * VEC;
        float m[VEC];
#pragma unroll
        for (int p = 0; p < VEC; ++p) m[p] = 1.f;
        if (mask && !last) svbq_ldv(mask + (size_t)b * T + t0, m);
        SvbQRows<VEC> q;
        q.zero();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float xn[2][VEC];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = k * 16 + 2 * e + h;
#pragma unroll
                for (int p = 0; p < VEC; ++p) xn[h][p] = 0.f;
                if (c < C) {
                    const size_t i0 = ((size_t)b * C + c) * T + t0, r0 = ((size_t)b * rc + c) * T + t0;
                    float o[VEC], ra[VEC];
#pragma unroll
                    for (int p = 0; p < VEC; ++p) o[p] = 0.f;
                    if (out) svbq_ldv(out + i0, o);
                    svbq_ldv(rs + r0, ra);
                    if (last) {
#pragma unroll
                        for (int p = 0; p < VEC; ++p) o[p] += ra[p];
                        svbq_stv(out_new + i0, o);
                    } else {
                        float xv[VEC], rb[VEC];
                        svbq_ldv(x + i0, xv);
                        svbq_ldv(rs + r0 + (size_t)C * T, rb);
#pragma unroll
                        for (int p = 0; p < VEC; ++p) { xn[h][p] = (xv[p] + ra[p]) * m[p]; o[p] += rb[p]; }
                        svbq_stv(x_new + i0, xn[h]);
                        svbq_stv(out_new + i0, o);
                    }
                }
            }
            if (x_new_q) {
#pragma unroll
                for (int p = 0; p < VEC; ++p) q.put(p, e, xn[0][p], xn[1][p]);
            }
        }
        if (x_new_q) q.store(x_new_q, ((size_t)b * kc + k) * T + t0);
    }
}

// backward of res/skip: drs[:, :C] = dx_new * mask ; drs[:, C:] = dout ; dxm = dx_new * mask (contiguous copy);
// optional Q image of drs (2C channels, C % 16 == 0)
template <int VEC>
__global__ __launch_bounds__(256) void svb_wn_res_skip_bwd_kernel(const float* dx_new, const float* dout, const float* mask,
                                                                  float* drs, float* dxm, unsigned short* drs_q, int B,
                                                                  int C, int T) {
    const int TV = T / VEC, kc = (C + 15) / 16;
    SVB_EW_LOOP((long)B * kc * TV) {
        const int tv = (int)(i_ % TV);
        const long bk = i_ / TV;
        const int k = (int)(bk % kc), b = (int)(bk / kc);
        const size_t t0 = (size_t)tv * VEC;
        float m[VEC];
#pragma unroll
        for (int p = 0; p < VEC; ++p) m[p] = 1.f;
        if (mask) svbq_ldv(mask + (size_t)b * T + t0, m);
        SvbQRows<VEC> qa, qb;
        qa.zero();
        qb.zero();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float va[2][VEC], vb[2][VEC];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = k * 16 + 2 * e + h;
#pragma unroll
                for (int p = 0; p < VEC; ++p) { va[h][p] = 0.f; vb[h][p] = 0.f; }
                if (c < C) {
                    const size_t i0 = ((size_t)b * C + c) * T + t0, r0 = ((size_t)b * 2 * C + c) * T + t0;
                    if (dx_new) {
                        svbq_ldv(dx_new + i0, va[h]);
#pragma unroll
                        for (int p = 0; p < VEC; ++p) va[h][p] *= m[p];
                    }
                    svbq_ldv(dout + i0, vb[h]);
                    svbq_stv(drs + r0, va[h]);
                    svbq_stv(drs + r0 + (size_t)C * T, vb[h]);
                    if (dxm) svbq_stv(dxm + i0, va[h]);
                }
            }
            if (drs_q) {
#pragma unroll
                for (int p = 0; p < VEC; ++p) { qa.put(p, e, va[0][p], va[1][p]); qb.put(p, e, vb[0][p], vb[1][p]); }
            }
        }
        if (drs_q) {
            qa.store(drs_q, ((size_t)b * 2 * kc + k) * T + t0);
            qb.store(drs_q, ((size_t)b * 2 * kc + kc + k) * T + t0);
        }
    }
}

// Q image of an fp32 [B][C][T] tensor (optionally of x * mask[b,t])
template <int VEC>
__global__ __launch_bounds__(256) void svb_split_q_kernel(const float* x, const float* mask, unsigned short* xq, int B, int C,
                                                          int T) {
    const int TV = T / VEC, kc = (C + 15) / 16;
    SVB_EW_LOOP((long)B * kc * TV) {
        const int tv = (int)(i_ % TV);
        const long bk = i_ / TV;
        const int k = (int)(bk % kc), b = (int)(bk / kc);
        const size_t t0 = (size_t)tv * VEC;
        float m[VEC];
#pragma unroll
        for (int p = 0; p < VEC; ++p) m[p] = 1.f;
        if (mask) svbq_ldv(mask + (size_t)b * T + t0, m);
        SvbQRows<VEC> q;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v[2][VEC];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = k * 16 + 2 * e + h;
#pragma unroll
                for (int p = 0; p < VEC; ++p) v[h][p] = 0.f;
                if (c < C) {
                    svbq_ldv(x + ((size_t)b * C + c) * T + t0, v[h]);
#pragma unroll
                    for (int p = 0; p < VEC; ++p) v[h][p] *= m[p];
                }
            }
#pragma unroll
            for (int p = 0; p < VEC; ++p) q.put(p, e, v[0][p], v[1][p]);
        }
        q.store(xq, ((size_t)b * kc + k) * T + t0);
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
// (Round 6: 3.0 TB/s at C = 256, T = 1124.  A 64-column x 4-group block -- 256-byte runs per load instruction, 64 values per
// thread -- measured 2.47 TB/s, profiles/r06_streaming_kernels_2.log: not the run length; kept as is.)
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

static inline bool ew_vec4(int T, std::initializer_list<const void*> ptrs) {
    if (T % 4) return false;
    for (const void* p : ptrs)
        if ((uintptr_t)p % 16) return false;
    return true;
}
#define SVB_EW_LAUNCH(kern, total_v4, total_v1, v4, ...)                                                              \
    do {                                                                                                              \
        if (v4) hipLaunchKernelGGL(kern<4>, dim3(ew_grid(total_v4)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);  \
        else hipLaunchKernelGGL(kern<1>, dim3(ew_grid(total_v1)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);     \
    } while (0)

extern "C" int svb_wn_gate_fwd(const float* xin, const float* g, float* acts, unsigned short* acts_q, int B, int C, int T,
                               int g_channels, int g_off, void* stream) {
    if (!xin || !acts || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    if (g && (g_off < 0 || g_off + 2 * C > g_channels)) return SVB_ERR_ARG;
    const long kc = (C + 15) / 16;
    const bool v4 = ew_vec4(T, {xin, g, acts});
    SVB_EW_LAUNCH(svb_wn_gate_fwd_kernel, (long)B * kc * (T / 4), (long)B * kc * T, v4, xin, g, acts, acts_q, B, C, T,
                  g_channels, g_off);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_wn_gate_bwd(const float* xin, const float* g, const float* dacts, float* dxin, float* dg,
                               unsigned short* dxin_q, int B, int C, int T, int g_channels, int g_off, void* stream) {
    if (!xin || !dacts || (!dxin && !dg) || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    if ((g || dg) && (g_off < 0 || g_off + 2 * C > g_channels)) return SVB_ERR_ARG;
    if (dxin_q && C % 16) return SVB_ERR_ARG;
    const long kc = (C + 15) / 16;
    const bool v4 = ew_vec4(T, {xin, g, dacts, dxin, dg});
    SVB_EW_LAUNCH(svb_wn_gate_bwd_kernel, (long)B * kc * (T / 4), (long)B * kc * T, v4, xin, g, dacts, dxin, dg, dxin_q, B, C,
                  T, g_channels, g_off);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_wn_res_skip(const float* x, const float* rs, const float* mask, const float* out, float* x_new,
                               float* out_new, unsigned short* x_new_q, int B, int C, int T, int last, void* stream) {
    if (!rs || !out_new || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    if (!last && (!x || !x_new)) return SVB_ERR_ARG;
    if (last && x_new_q) return SVB_ERR_ARG;
    const long kc = (C + 15) / 16;
    const bool v4 = ew_vec4(T, {x, rs, mask, out, x_new, out_new});
    SVB_EW_LAUNCH(svb_wn_res_skip_kernel, (long)B * kc * (T / 4), (long)B * kc * T, v4, x, rs, mask, out, x_new, out_new,
                  x_new_q, B, C, T, last);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_wn_res_skip_bwd(const float* dx_new, const float* dout, const float* mask, float* drs, float* dxm,
                                   unsigned short* drs_q, int B, int C, int T, void* stream) {
    if (!dout || !drs || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    if (drs_q && C % 16) return SVB_ERR_ARG;
    const long kc = (C + 15) / 16;
    const bool v4 = ew_vec4(T, {dx_new, dout, mask, drs, dxm});
    SVB_EW_LAUNCH(svb_wn_res_skip_bwd_kernel, (long)B * kc * (T / 4), (long)B * kc * T, v4, dx_new, dout, mask, drs, dxm,
                  drs_q, B, C, T);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_split_q(const float* x, const float* mask, unsigned short* xq, int B, int C, int T, void* stream) {
    if (!x || !xq || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    const long kc = (C + 15) / 16;
    const bool v4 = ew_vec4(T, {x, mask});
    SVB_EW_LAUNCH(svb_split_q_kernel, (long)B * kc * (T / 4), (long)B * kc * T, v4, x, mask, xq, B, C, T);
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
