// conv2d.hip -- im2col / col2im for the small strided 2-D convolutions of the multi-window mel critic
// (reference modules/fastspeech/multi_window_disc.py:14-31: Conv2d(c, 128, 3x3, stride 2, pad 1) on [B,c,T,80] windows).
//
// The GEMM itself runs on the implicit-GEMM conv kernel of conv1d.hip as a 1x1 conv over the column matrix:
//   cols[b, (ci,jh,jw), (ho,wo)] = x[b, ci, ho*sh + jh - ph, wo*sw + jw - pw]      (zero outside)
//   y[b, co, (ho,wo)]            = sum_k w[co, k] * cols[b, k, (ho,wo)]
// x and cols carry explicit batch / channel(row) element strides, so the column matrix can be laid out either per
// clip ([B][K][L]) or with the batch folded into the position axis ([K][B*L]) -- the latter turns the late critic
// layers (L = 40..160 positions per clip) into one wide GEMM instead of B skinny ones.
// Both kernels are pure streaming (HBM-bound): lanes run along the contiguous (ho,wo) axis.
#include "svb_common.h"
#include "../../include/svb_hip.h"

__global__ __launch_bounds__(256) void svb_im2col_kernel(const float* x, float* cols, int B, int C, int H, int W, int KH,
                                                         int KW, int SH, int SW, int PH, int PW, int Ho, int Wo, long x_sb, long x_sc,
                                                         long c_sb, long c_sk) {
    const long n_pos = (long)Ho * Wo;
    const long total = (long)B * C * KH * KW * n_pos;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pos = i % n_pos;
        long r = i / n_pos;
        const int jw = (int)(r % KW); r /= KW;
        const int jh = (int)(r % KH); r /= KH;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const int ho = (int)(pos / Wo), wo = (int)(pos - (long)ho * Wo);
        const int h = ho * SH + jh - PH, w = wo * SW + jw - PW;
        float v = 0.f;
        if (h >= 0 && h < H && w >= 0 && w < W) v = x[(long)b * x_sb + (long)c * x_sc + (long)h * W + w];
        cols[(long)b * c_sb + (((long)c * KH + jh) * KW + jw) * c_sk + pos] = v;
    }
}

// dx[b,c,h,w] = sum over (jh,jw) with (h+PH-jh) % SH == 0 and (w+PW-jw) % SW == 0 of dcols[b,(c,jh,jw),(ho,wo)]  (gather form)
__global__ __launch_bounds__(256) void svb_col2im_kernel(const float* dcols, float* dx, int B, int C, int H, int W, int KH,
                                                         int KW, int SH, int SW, int PH, int PW, int Ho, int Wo, long c_sb, long c_sk) {
    const long total = (long)B * C * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int w = (int)(i % W);
        long r = i / W;
        const int h = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        float s = 0.f;
        for (int jh = 0; jh < KH; ++jh) {
            const int th = h + PH - jh;
            if (th < 0 || th % SH) continue;
            const int ho = th / SH;
            if (ho >= Ho) continue;
            for (int jw = 0; jw < KW; ++jw) {
                const int tw = w + PW - jw;
                if (tw < 0 || tw % SW) continue;
                const int wo = tw / SW;
                if (wo >= Wo) continue;
                s += dcols[(long)b * c_sb + (((long)c * KH + jh) * KW + jw) * c_sk + (long)ho * Wo + wo];
            }
        }
        dx[i] = s;
    }
}

static inline int g1d(long total) {
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int svb_im2col(const float* x, float* cols, int B, int C, int H, int W, int KH, int KW, int SH, int SW, int PH,
                          int PW, int Ho, int Wo, long x_sb, long x_sc, long cols_sb, long cols_sk, void* stream) {
    if (!x || !cols || B <= 0 || C <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || SH <= 0 || SW <= 0 || Ho <= 0 || Wo <= 0)
        return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_im2col_kernel, dim3(g1d((long)B * C * KH * KW * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, cols,
                       B, C, H, W, KH, KW, SH, SW, PH, PW, Ho, Wo, x_sb, x_sc, cols_sb, cols_sk);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_col2im(const float* dcols, float* dx, int B, int C, int H, int W, int KH, int KW, int SH, int SW, int PH,
                          int PW, int Ho, int Wo, long cols_sb, long cols_sk, void* stream) {
    if (!dcols || !dx || B <= 0 || C <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || SH <= 0 || SW <= 0 || Ho <= 0 || Wo <= 0)
        return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_col2im_kernel, dim3(g1d((long)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, dcols, dx, B, C, H,
                       W, KH, KW, SH, SW, PH, PW, Ho, Wo, cols_sb, cols_sk);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Space-to-depth + zero border for the stride-2 3x3 convs WITHOUT the 9x im2col expansion.
//   out[(ph*2+pw)*C + c][n][i+1][j+1] = x[n][c][2i+ph][2j+pw],   row 0 and column 0 of every (channel, clip) plane = 0
// (x addressed with element strides; out contiguous [4C][N][Ho+1][Wo+1], Ho = H/2, Wo = W/2, H and W even).
// A 3x3 stride-2 pad-1 conv of x is then a 2x2 stride-1 conv of `out`, i.e. a 1-D conv over the flattened planes of pitch
// P = Wo+1 with taps {-P-1, -P, -1, 0} (svb_conv1d_taps): kernel row kh maps to (plane row offset, row phase) =
// 0 -> (-1, 1), 1 -> (0, 0), 2 -> (0, 1), the same for columns.  Data moved: 1 read + 1 write of the activation.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void svb_s2d_pad_kernel(const float* x, float* out, int N, int C, int Ho, int Wo, long sn,
                                                          long sc, long sh, long sw) {
    const int P = Wo + 1, R = Ho + 1;
    const long total = 4L * C * N * R * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int jp = (int)(i % P);
        long r = i / P;
        const int ip = (int)(r % R); r /= R;
        const int n = (int)(r % N); r /= N;
        const int c = (int)(r % C);
        const int blk = (int)(r / C);
        float v = 0.f;
        if (ip > 0 && jp > 0) {
            const int h = 2 * (ip - 1) + (blk >> 1), w = 2 * (jp - 1) + (blk & 1);
            v = x[(long)n * sn + (long)c * sc + (long)h * sh + (long)w * sw];
        }
        out[i] = v;
    }
}

// inverse gather: dx[n][c][h][w] (contiguous) = dout[((h&1)*2 + (w&1))*C + c][n][h/2 + 1][w/2 + 1]
__global__ __launch_bounds__(256) void svb_s2d_pad_bwd_kernel(const float* dout, float* dx, int N, int C, int Ho, int Wo) {
    const int P = Wo + 1, R = Ho + 1, H = 2 * Ho, W = 2 * Wo;
    const long total = (long)N * C * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int w = (int)(i % W);
        long r = i / W;
        const int h = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        const int blk = (h & 1) * 2 + (w & 1);
        dx[i] = dout[((((long)blk * C + c) * N + n) * R + (h >> 1) + 1) * P + (w >> 1) + 1];
    }
}

extern "C" int svb_s2d_pad(const float* x, float* out, int N, int C, int H, int W, long sn, long sc, long sh, long sw,
                           void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_s2d_pad_kernel, dim3(g1d(4L * C * N * (H / 2 + 1) * (W / 2 + 1))), dim3(256), 0, (hipStream_t)stream,
                       x, out, N, C, H / 2, W / 2, sn, sc, sh, sw);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_s2d_pad_bwd(const float* dout, float* dx, int N, int C, int H, int W, void* stream) {
    if (!dout || !dx || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_s2d_pad_bwd_kernel, dim3(g1d((long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, dout, dx, N, C,
                       H / 2, W / 2);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
