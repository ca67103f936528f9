// collate.hip -- the data side of a training batch on the GPU: zero-padded collation of ragged clips and the
// pitch-track normalisation, so that a batch costs ONE pinned-memory copy per dtype plus a handful of launches instead of
// ~40 small host tensor ops per clip.
//
//   reference: utils/__init__.py:118-161 (collate_1d / collate_2d), utils/pitch_utils.py:148-177 (norm_f0, norm_interp_f0),
//              tasks/singing/svb_vae_task.py:20-45 + tasks/tts/dataset_utils.py:133-205 (which fields, which pads)
//
// Ragged input: the rows of all clips of a batch concatenated in one staging buffer ([sum_len][W]); off[b] = first row of
// clip b, len[b] = its row count (already truncated to max_frames and the frame multiple by the host).  HBM-bound copies.
#include "svb_common.h"
#include "../../include/svb_hip.h"

static inline int svbc_grid(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 65535L * 16 ? 65535L * 16 : (g < 1 ? 1 : g));
}

__global__ __launch_bounds__(256) void svb_collate_pad_f32_kernel(const float* src, const int* off, const int* len, float* out,
                                                                  int B, int Tmax, int W, float pad) {
    const long total = (long)B * Tmax * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int w = (int)(i % W);
        const long r = i / W;
        const int t = (int)(r % Tmax), b = (int)(r / Tmax);
        out[i] = t < len[b] ? src[((long)off[b] + t) * W + w] : pad;
    }
}

// int64 rows of width 1 (pitch bins, alignments); clip_max (optional, per clip): values are clamped to it -- the reference
// clips `a2p_f0_alignment` to the amateur clip's last frame (svb_vae_task.py:31-33).
__global__ __launch_bounds__(256) void svb_collate_pad_i64_kernel(const int64_t* src, const int* off, const int* len,
                                                                  const int* clip_max, int64_t* out, int B, int Tmax,
                                                                  int64_t pad) {
    const long total = (long)B * Tmax;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int t = (int)(i % Tmax), b = (int)(i / Tmax);
        int64_t v = pad;
        if (t < len[b]) {
            v = src[(long)off[b] + t];
            if (clip_max && v > clip_max[b]) v = clip_max[b];
        }
        out[i] = v;
    }
}

// energy[b][t] = sqrt(sum_f exp(mel[b][t][f])^2) for real frames, 0 for padding (dataset_utils.py:159-160): one wave per frame
__global__ __launch_bounds__(256) void svb_mel_energy_kernel(const float* mels, const int* len, float* energy, int B, int Tmax,
                                                             int W) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * Tmax) return;
    const int t = (int)(row % Tmax), b = (int)(row / Tmax);
    float s = 0.f;
    if (t < len[b])
        for (int f = lane; f < W; f += 64) {
            const float e = expf(mels[row * W + f]);
            s += e * e;
        }
    s = svb_wave_sum(s);
    if (lane == 0) energy[row] = t < len[b] ? sqrtf(s) : 0.f;
}

// ------------------------------------------------------------------------------------------------------------------
// norm_interp_f0 (pitch_utils.py:160-177), fp64 like the reference's numpy path, results rounded to fp32 at the end
// (torch.FloatTensor(f0)):  uv = f0 == 0;  f0 = log2(f0 + 1e-8) | (f0 - mean) / std | f0;  use_uv: f0[uv] = 0;
// all unvoiced -> zeros, else f0[uv] = np.interp(where(uv), where(~uv), f0[~uv]) -- numpy's formula, evaluated without
// fused multiply-add:  slope = (fp[j+1] - fp[j]) / (xp[j+1] - xp[j]);  y = slope * (x - xp[j]) + fp[j];  constant
// extrapolation outside the voiced range.
// One workgroup per clip; the normalised track and the voiced flags sit in LDS; every unvoiced frame walks to its voiced
// neighbours (unvoiced stretches are short).  mode: 0 none, 1 log, 2 standard.
// ------------------------------------------------------------------------------------------------------------------
#define SVBC_F0_MAXT 6144        /* frames per clip held in LDS (54 KB) */
__global__ __launch_bounds__(256) void svb_norm_interp_f0_kernel(const double* src, const int* off, const int* len, float* f0_out,
                                                                 float* uv_out, int Tmax, int mode, double mean, double stdv,
                                                                 int use_uv) {
    __shared__ double val[SVBC_F0_MAXT];
    __shared__ unsigned char voiced[SVBC_F0_MAXT];
    __shared__ int any_voiced;
    const int b = blockIdx.x, n = len[b];
    const double* s = src + off[b];
    if (threadIdx.x == 0) any_voiced = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += 256) {
        const double f = s[t];
        const bool uv = f == 0.0;
        double v = f;
        if (mode == 1) v = log2(f + 1e-8);
        else if (mode == 2) v = (f - mean) / stdv;
        if (use_uv && uv) v = 0.0;
        val[t] = v;
        voiced[t] = uv ? 0 : 1;
        if (!uv) any_voiced = 1;                       // (benign race: every writer stores 1)
    }
    __syncthreads();
    const int have = any_voiced;
    for (int t = threadIdx.x; t < Tmax; t += 256) {
        float o = 0.f, u = 0.f;
        if (t < n) {
            u = voiced[t] ? 0.f : 1.f;
            double v = val[t];
            if (!voiced[t]) {
                if (!have) v = 0.0;
                else {
                    int l = t - 1, r = t + 1;
                    while (l >= 0 && !voiced[l]) --l;
                    while (r < n && !voiced[r]) ++r;
                    if (l < 0) v = val[r];                                  // left of the first voiced frame
                    else if (r >= n) v = val[l];                            // right of the last
                    else {
                        const double slope = __ddiv_rn(__dsub_rn(val[r], val[l]), (double)(r - l));
                        v = __dadd_rn(__dmul_rn(slope, (double)(t - l)), val[l]);
                    }
                }
            }
            o = (float)v;
        }
        f0_out[(long)b * Tmax + t] = o;
        uv_out[(long)b * Tmax + t] = u;
    }
}

extern "C" int svb_collate_pad_f32(const float* src, const int* off, const int* len, float* out, int B, int Tmax, int W, float pad,
                                   void* stream) {
    if (!src || !off || !len || !out || B <= 0 || Tmax <= 0 || W <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_collate_pad_f32_kernel, dim3(svbc_grid((long)B * Tmax * W)), dim3(256), 0, (hipStream_t)stream, src, off,
                       len, out, B, Tmax, W, pad);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_collate_pad_i64(const int64_t* src, const int* off, const int* len, const int* clip_max, int64_t* out, int B,
                                   int Tmax, int64_t pad, void* stream) {
    if (!src || !off || !len || !out || B <= 0 || Tmax <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_collate_pad_i64_kernel, dim3(svbc_grid((long)B * Tmax)), dim3(256), 0, (hipStream_t)stream, src, off, len,
                       clip_max, out, B, Tmax, pad);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_mel_energy(const float* mels, const int* len, float* energy, int B, int Tmax, int W, void* stream) {
    if (!mels || !len || !energy || B <= 0 || Tmax <= 0 || W <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_mel_energy_kernel, dim3((unsigned)(((long)B * Tmax + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mels,
                       len, energy, B, Tmax, W);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_norm_interp_f0(const double* src, const int* off, const int* len, float* f0_out, float* uv_out, int B, int Tmax,
                                  int mode, double mean, double stdv, int use_uv, void* stream) {
    if (!src || !off || !len || !f0_out || !uv_out || B <= 0 || Tmax <= 0 || mode < 0 || mode > 2) return SVB_ERR_ARG;
    if (Tmax > SVBC_F0_MAXT) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_norm_interp_f0_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, src, off, len, f0_out, uv_out, Tmax,
                       mode, mean, stdv, use_uv);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
