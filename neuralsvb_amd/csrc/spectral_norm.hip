// spectral_norm.hip -- torch.nn.utils.spectral_norm of a conv weight as 3 launches forward and 2 backward.
//
// The first scale discriminator of the NSF-HifiGAN MSD is spectrally normalised (reference modules/hifigan/hifigan.py:238-250:
// `norm_f = spectral_norm` for DiscriminatorS(use_spectral_norm=True)): every forward call of each of its 8 convs runs one power
// iteration on the [Cout, Cin/g * k] weight matrix -- v = normalize(W^T u), u = normalize(W v), sigma = u . (W v), W / sigma --
// which is 15 tiny torch launches per conv and call and ~12 more in backward: ~860 of the vocoder step's 2 250 launches.
//   K1  v_raw = W^T u            thread per column, rows streamed (coalesced); per-block partial of |v_raw|^2
//   K2  t = W v_raw / max(|v_raw|, eps)   wave per row; per-block partial of |t|^2          (t = W v for the normalised v)
//   K3  sigma = |t|^2 / max(|t|, eps)  (= u . (W v) for u = t / max(|t|, eps));  W_sn = W / sigma;  u, v written to the module's
//       buffers (in place, as torch does under no_grad) and to the copies the backward reads
//   eval mode (no power iteration): K2 with the stored v, K3 with sigma = u . (W v) for the stored u; nothing is written back.
//   backward: dW = dW_sn / sigma - (sum dW_sn * W) / sigma^2 * u v^T     (u, v are constants of the graph, as in torch)
// Partials are reduced by every workgroup in the same fixed order: deterministic, no atomics.
#include "svb_common.h"

#define SVB_SN_THREADS 256
#define SVB_SN_MAXPART 4096

struct SvbSnArgs {
    const float* w;
    const float* u_in;
    const float* v_in;
    float* u_buf;       // module buffers (training: updated in place)
    float* v_buf;
    float* u_save;      // copies for the backward
    float* v_save;
    float* w_sn;
    float* sigma_out;
    float* v_raw;       // workspace [C]
    float* t;           // workspace [R]
    float* part1;       // workspace [nb1]
    float* part2;       // workspace [nb2]
    int R, C, nb1, nb2, training;
    float eps;
};

__global__ __launch_bounds__(SVB_SN_THREADS) void svb_sn_wtu_kernel(SvbSnArgs a) {
    __shared__ float red[SVB_SN_THREADS / 64];
    const int c = blockIdx.x * SVB_SN_THREADS + threadIdx.x;
    float acc = 0.f;
    if (c < a.C) {
        const float* wc = a.w + c;
        int r = 0;
        for (; r + 4 <= a.R; r += 4) {
            const float w0 = wc[(size_t)r * a.C], w1 = wc[(size_t)(r + 1) * a.C], w2 = wc[(size_t)(r + 2) * a.C], w3 = wc[(size_t)(r + 3) * a.C];
            acc = fmaf(w0, a.u_in[r], acc);
            acc = fmaf(w1, a.u_in[r + 1], acc);
            acc = fmaf(w2, a.u_in[r + 2], acc);
            acc = fmaf(w3, a.u_in[r + 3], acc);
        }
        for (; r < a.R; ++r) acc = fmaf(wc[(size_t)r * a.C], a.u_in[r], acc);
        a.v_raw[c] = acc;
    }
    const float s = svb_block_sum<SVB_SN_THREADS>(c < a.C ? acc * acc : 0.f, red);
    if (threadIdx.x == 0) a.part1[blockIdx.x] = s;
}

// 4 rows per workgroup (one wave each)
__global__ __launch_bounds__(SVB_SN_THREADS) void svb_sn_wv_kernel(SvbSnArgs a) {
    __shared__ float red[SVB_SN_THREADS / 64];
    __shared__ float rowsq[SVB_SN_THREADS / 64];
    float inv = 1.f;
    const float* v = a.v_in;
    if (a.training) {
        float s = 0.f;
        for (int i = threadIdx.x; i < a.nb1; i += SVB_SN_THREADS) s += a.part1[i];
        s = svb_block_sum<SVB_SN_THREADS>(s, red);
        inv = 1.f / fmaxf(sqrtf(s), a.eps);
        v = a.v_raw;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * (SVB_SN_THREADS / 64) + wave;
    float acc = 0.f;
    if (r < a.R) {
        const float* wr = a.w + (size_t)r * a.C;
        for (int c = lane; c < a.C; c += 64) acc = fmaf(wr[c], v[c], acc);
    }
    acc = svb_wave_sum(acc) * inv;
    if (lane == 0) {
        if (r < a.R) a.t[r] = acc;
        rowsq[wave] = r < a.R ? acc * acc : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < SVB_SN_THREADS / 64; ++w) s += rowsq[w];
        a.part2[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(SVB_SN_THREADS) void svb_sn_scale_kernel(SvbSnArgs a) {
    __shared__ float red[SVB_SN_THREADS / 64];
    float sigma, uinv = 1.f, vinv = 1.f;
    if (a.training) {
        float s = 0.f;
        for (int i = threadIdx.x; i < a.nb2; i += SVB_SN_THREADS) s += a.part2[i];
        s = svb_block_sum<SVB_SN_THREADS>(s, red);
        uinv = 1.f / fmaxf(sqrtf(s), a.eps);
        sigma = s * uinv;                                   // u . (W v) with u = t * uinv, W v = t
        float s1 = 0.f;
        for (int i = threadIdx.x; i < a.nb1; i += SVB_SN_THREADS) s1 += a.part1[i];
        s1 = svb_block_sum<SVB_SN_THREADS>(s1, red);
        vinv = 1.f / fmaxf(sqrtf(s1), a.eps);
    } else {
        float s = 0.f;
        for (int i = threadIdx.x; i < a.R; i += SVB_SN_THREADS) s = fmaf(a.u_in[i], a.t[i], s);
        sigma = svb_block_sum<SVB_SN_THREADS>(s, red);
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) a.sigma_out[0] = sigma;
        for (int i = threadIdx.x; i < a.R; i += SVB_SN_THREADS) {
            const float uv = a.training ? a.t[i] * uinv : a.u_in[i];
            if (a.training) a.u_buf[i] = uv;
            a.u_save[i] = uv;
        }
        for (int i = threadIdx.x; i < a.C; i += SVB_SN_THREADS) {
            const float vv = a.training ? a.v_raw[i] * vinv : a.v_in[i];
            if (a.training) a.v_buf[i] = vv;
            a.v_save[i] = vv;
        }
    }
    const size_t n = (size_t)a.R * a.C;
    for (size_t i = (size_t)blockIdx.x * SVB_SN_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * SVB_SN_THREADS)
        a.w_sn[i] = a.w[i] / sigma;
}

extern "C" size_t svb_spectral_norm_workspace_floats(int R, int C) {
    if (R <= 0 || C <= 0) return 0;
    return (size_t)C + R + svb_cdiv(C, SVB_SN_THREADS) + svb_cdiv(R, SVB_SN_THREADS / 64) + 256;     // (+256: the backward's partials)
}

extern "C" int svb_spectral_norm_fwd(const float* w, float* u, float* v, float* w_sn, float* u_save, float* v_save, float* sigma_out,
                                     int R, int C, int training, float eps, float* workspace, void* stream) {
    if (!w || !u || !v || !w_sn || !u_save || !v_save || !sigma_out || !workspace || R <= 0 || C <= 0) return SVB_ERR_ARG;
    SvbSnArgs a;
    a.w = w; a.u_in = u; a.v_in = v; a.u_buf = u; a.v_buf = v; a.u_save = u_save; a.v_save = v_save; a.w_sn = w_sn;
    a.sigma_out = sigma_out; a.R = R; a.C = C; a.training = training ? 1 : 0; a.eps = eps;
    a.nb1 = svb_cdiv(C, SVB_SN_THREADS); a.nb2 = svb_cdiv(R, SVB_SN_THREADS / 64);
    if (a.nb1 > SVB_SN_MAXPART || a.nb2 > (1 << 20)) return SVB_ERR_UNSUPPORTED;
    a.v_raw = workspace; a.t = a.v_raw + C; a.part1 = a.t + R; a.part2 = a.part1 + a.nb1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a.training) {
        hipLaunchKernelGGL(svb_sn_wtu_kernel, dim3(a.nb1), dim3(SVB_SN_THREADS), 0, st, a);
        SVB_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(svb_sn_wv_kernel, dim3(a.nb2), dim3(SVB_SN_THREADS), 0, st, a);
    SVB_CHECK_LAUNCH();
    size_t nb = ((size_t)R * C + 4 * SVB_SN_THREADS - 1) / (4 * SVB_SN_THREADS);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(svb_sn_scale_kernel, dim3((unsigned)nb), dim3(SVB_SN_THREADS), 0, st, a);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

struct SvbSnBwdArgs {
    const float* dw_sn;
    const float* w;
    const float* u;
    const float* v;
    const float* sigma;
    float* dw;
    float* part;
    int R, C, nb;
};

__global__ __launch_bounds__(SVB_SN_THREADS) void svb_sn_bwd_dot_kernel(SvbSnBwdArgs a) {
    __shared__ float red[SVB_SN_THREADS / 64];
    const size_t n = (size_t)a.R * a.C;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * SVB_SN_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * SVB_SN_THREADS)
        s = fmaf(a.dw_sn[i], a.w[i], s);
    s = svb_block_sum<SVB_SN_THREADS>(s, red);
    if (threadIdx.x == 0) a.part[blockIdx.x] = s;
}

__global__ __launch_bounds__(SVB_SN_THREADS) void svb_sn_bwd_kernel(SvbSnBwdArgs a) {
    __shared__ float red[SVB_SN_THREADS / 64];
    float s = 0.f;
    for (int i = threadIdx.x; i < a.nb; i += SVB_SN_THREADS) s += a.part[i];
    s = svb_block_sum<SVB_SN_THREADS>(s, red);
    const float sg = a.sigma[0], isg = 1.f / sg, k = s * isg * isg;
    const size_t n = (size_t)a.R * a.C;
    for (size_t i = (size_t)blockIdx.x * SVB_SN_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * SVB_SN_THREADS) {
        const size_t r = i / a.C, c = i - r * a.C;
        a.dw[i] = a.dw_sn[i] * isg - k * a.u[r] * a.v[c];
    }
}

extern "C" int svb_spectral_norm_bwd(const float* dw_sn, const float* w, const float* u, const float* v, const float* sigma, float* dw,
                                     int R, int C, float* workspace, void* stream) {
    if (!dw_sn || !w || !u || !v || !sigma || !dw || !workspace || R <= 0 || C <= 0) return SVB_ERR_ARG;
    SvbSnBwdArgs a;
    a.dw_sn = dw_sn; a.w = w; a.u = u; a.v = v; a.sigma = sigma; a.dw = dw; a.part = workspace; a.R = R; a.C = C;
    size_t nb = ((size_t)R * C + 4 * SVB_SN_THREADS - 1) / (4 * SVB_SN_THREADS);
    if (nb > 256) nb = 256;                                  // (the workspace query reserves 256 floats for these partials)
    a.nb = (int)nb;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(svb_sn_bwd_dot_kernel, dim3((unsigned)nb), dim3(SVB_SN_THREADS), 0, st, a);
    SVB_CHECK_LAUNCH();
    size_t nb2 = ((size_t)R * C + 4 * SVB_SN_THREADS - 1) / (4 * SVB_SN_THREADS);
    if (nb2 > 1024) nb2 = 1024;
    hipLaunchKernelGGL(svb_sn_bwd_kernel, dim3((unsigned)nb2), dim3(SVB_SN_THREADS), 0, st, a);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
