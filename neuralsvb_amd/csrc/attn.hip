// attn.hip -- the glue between the three matrix products of the conformer's relative-position self-attention
// (reference modules/commons/espnet_transformer_attn.py:125-186, RelPositionMultiHeadedAttention, legacy rel_shift):
//
//   scores[i][j] = (ac[i][j] + shift(bd)[i][j]) / sqrt(d_k);  masked keys -> -FLT_MAX;  attn = softmax_j(scores);  masked -> 0
//
// where shift() is the "pad one zero column, view [T+1,T], drop the first row" trick (:125-148), i.e.
//   shift(bd)[i][j] = bd[i][T-1-i+j]        for j <= i
//                   = 0                      for j == i+1
//                   = bd[i+1][j-i-2]         for j >  i+1
// torch runs this as pad / view / slice-copy / add / div / masked_fill / softmax / masked_fill: eight passes over the
// [B,h,T,T] tensors.  Here one wave owns one query row: it reads the ac row and the two bd rows once, keeps the row in
// registers, does the max / sum reductions with wave shuffles and writes attn once.  HBM-bound: 12 B per score element.
#include "svb_common.h"
#include "../../include/svb_hip.h"

#define SVB_ATTN_MAXPL 32      /* row elements per lane kept in registers: T <= 2048 */

__global__ __launch_bounds__(256) void svb_relpos_softmax_kernel(const float* ac, const float* bd, const float* keep,
                                                                 float* attn, int B, int H, int T, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (b*H + h)*T + i
    if (row >= (long)B * H * T) return;
    const int i = (int)(row % T);
    const int b = (int)(row / ((long)H * T));
    const float* acr = ac + row * T;
    const float* bdr = bd + row * T;                                      // bd[i][.]; bd[i+1][.] follows at +T
    const float* kp = keep + (long)b * T;
    float* out = attn + row * T;
    float v[SVB_ATTN_MAXPL];
    float mx = -3.402823466e+38f;
#pragma unroll
    for (int m = 0; m < SVB_ATTN_MAXPL; ++m) {
        const int j = lane + 64 * m;
        float s = -3.402823466e+38f;
        if (j < T) {
            float sh;
            if (j <= i) sh = bdr[T - 1 - i + j];
            else if (j == i + 1) sh = 0.f;
            else sh = bdr[T + j - i - 2];
            s = (acr[j] + sh) * scale;
            if (kp[j] == 0.f) s = -3.402823466e+38f;                      // masked_fill(min)
        }
        v[m] = s;
        mx = fmaxf(mx, s);
    }
    mx = svb_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < SVB_ATTN_MAXPL; ++m) {
        const int j = lane + 64 * m;
        const float e = j < T ? expf(v[m] - mx) : 0.f;
        v[m] = e;
        sum += e;
    }
    sum = svb_wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int m = 0; m < SVB_ATTN_MAXPL; ++m) {
        const int j = lane + 64 * m;
        if (j < T) out[j] = kp[j] == 0.f ? 0.f : v[m] * inv;
    }
}

extern "C" int svb_relpos_softmax(const float* ac, const float* bd, const float* keep, float* attn, int B, int H, int T,
                                  float scale, void* stream) {
    if (!ac || !bd || !keep || !attn || B <= 0 || H <= 0 || T <= 0) return SVB_ERR_ARG;
    if (T > 64 * SVB_ATTN_MAXPL) return SVB_ERR_UNSUPPORTED;
    const long rows = (long)B * H * T;
    hipLaunchKernelGGL(svb_relpos_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ac, bd,
                       keep, attn, B, H, T, scale);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Conformer convolution module, the part between its two pointwise convs (reference
// modules/fastspeech/conformer/layers.py:47-63, eval mode):
//     u = GLU(y, dim=channels) = y[:, :C] * sigmoid(y[:, C:])
//     v = depthwise_conv1d(u; w[C][K], bias, pad (K-1)/2)
//     z = BatchNorm1d(v) with running statistics          (the PPG encoder is frozen and always in eval mode)
//     out = z * sigmoid(z)                                 (Swish)
// torch runs it as glu + MIOpen depthwise conv + batch_norm + sigmoid + mul (five passes, 2C*T reads + 4 x C*T round trips);
// here one workgroup owns one (batch, channel) row tile: the GLU'd row segment + halo goes to LDS once, every thread then
// produces one output position from K LDS reads.  HBM-bound: 12 B per output element (2 reads + 1 write).
// ------------------------------------------------------------------------------------------------------------------
#define SVB_DW_TILE 512
#define SVB_DW_MAXK 63
__global__ __launch_bounds__(256) void svb_glu_dwconv_bn_swish_kernel(const float* y, const float* w, const float* bias,
                                                                      const float* bn_w, const float* bn_b,
                                                                      const float* bn_mean, const float* bn_var, float eps,
                                                                      float* out, int B, int C, int T, int K) {
    __shared__ float u[SVB_DW_TILE + SVB_DW_MAXK];
    __shared__ float wk[SVB_DW_MAXK];
    const int tiles = (T + SVB_DW_TILE - 1) / SVB_DW_TILE;
    const int tile = blockIdx.x % tiles;
    const int bc = blockIdx.x / tiles;
    const int c = bc % C, b = bc / C;
    const int t0 = tile * SVB_DW_TILE, pad = (K - 1) / 2;
    const float* ya = y + ((size_t)b * 2 * C + c) * T;
    const float* yg = ya + (size_t)C * T;
    for (int i = threadIdx.x; i < SVB_DW_TILE + K - 1; i += 256) {
        const int t = t0 - pad + i;
        u[i] = (t >= 0 && t < T) ? ya[t] * svb_sigmoid(yg[t]) : 0.f;
    }
    if (threadIdx.x < K) wk[threadIdx.x] = w[(size_t)c * K + threadIdx.x];
    __syncthreads();
    const float scale = (bn_w ? bn_w[c] : 1.f) / sqrtf(bn_var[c] + eps);
    const float shift = (bn_b ? bn_b[c] : 0.f) - bn_mean[c] * scale + (bias ? bias[c] * scale : 0.f);
    for (int i = threadIdx.x; i < SVB_DW_TILE; i += 256) {
        const int t = t0 + i;
        if (t >= T) break;
        float acc = 0.f;
        for (int j = 0; j < K; ++j) acc = fmaf(wk[j], u[i + j], acc);
        const float z = acc * scale + shift;
        out[((size_t)b * C + c) * T + t] = z * svb_sigmoid(z);
    }
}

extern "C" int svb_glu_dwconv_bn_swish(const float* y, const float* w, const float* bias, const float* bn_w,
                                       const float* bn_b, const float* bn_mean, const float* bn_var, float eps, float* out,
                                       int B, int C, int T, int K, void* stream) {
    if (!y || !w || !bn_mean || !bn_var || !out || B <= 0 || C <= 0 || T <= 0 || K <= 0 || !(K & 1)) return SVB_ERR_ARG;
    if (K > SVB_DW_MAXK) return SVB_ERR_UNSUPPORTED;
    const long blocks = (long)B * C * ((T + SVB_DW_TILE - 1) / SVB_DW_TILE);
    if (blocks > 0x7fffffffL) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_glu_dwconv_bn_swish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, w, bias,
                       bn_w, bn_b, bn_mean, bn_var, eps, out, B, C, T, K);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
