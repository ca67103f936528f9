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
