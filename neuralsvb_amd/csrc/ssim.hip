// ssim.hip -- SSIM map of two mel "images" [B, T, F] with an 11x11 gaussian window, forward and backward, as
// streaming stencil kernels (gfx950).  Replaces the five depthwise F.conv2d calls + ~15 elementwise passes of the
// reference's mel loss (modules/commons/ssim.py:331-351, called by tasks/tts/fs2.py:166-175 on [B,1,T,80]+6).
//
// One workgroup owns a tile of TT frames x all F bins; x=pred+bias and y=target+bias are staged once into LDS with the
// 5-wide zero halo that F.conv2d(padding=5) implies; the separable window is applied along the bins into LDS planes, then
// along the frames per pixel (see ssim_hpass5 / ssim_vpass5).
// Algorithmic bytes: forward reads 2 and writes 1 value per pixel (12 B/pixel); backward reads 3, writes 1 (+3 maps
// of workspace written and re-read).  Strided inputs are accepted so the [B,80,T] decoder output is read in place.
#include "svb_common.h"
#include "../../include/svb_hip.h"

#define SSIM_W 11
#define SSIM_R 5
#ifndef SSIM_TT
#define SSIM_TT 16
#endif
#define SSIM_FMAX 128
#define SSIM_ROWS_ (SSIM_TT + 2 * SSIM_R)

struct SsimWin { float g[SSIM_W]; };

struct SsimStats { float mu1, mu2, e11, e22, e12; };

typedef SvbDiv256 SsimDiv;        // (svb_common.h: division-free (row, column) walk of a 256-thread workgroup)

// The gaussian window is separable: the five moment images are filtered along the bins first (every staged row, into `hm`:
// 5 planes of rows x F), then along the frames per output pixel -- 11 + 11 taps per moment instead of 121.  (Round 3: the
// full 2-D windows made these kernels VALU/LDS-bound -- ~1.2 k VALU ops and 242 LDS reads per pixel, 70 us per call at
// [32, 1124, 80] for 23 MB of input -- the separable form needs ~150 and ~90.)
// (Round 6: NOT LDS-instruction bound any more.  The same two passes with ds_read_b128 -- four adjacent bins per thread, 5x fewer
// LDS instructions, bit-identical sums -- measured 98.8 us forward / 201 us backward against 96.8 / 151 for these scalar passes
// (profiles/r06_streaming_kernels_3.log; with two output rows per thread the frames pass needed > 256 VGPRs and ran 2x slower,
// r06_streaming_kernels_2.log).  Even the L1-only call, which touches no LDS, takes 34 us for 23 MB: a workgroup is a chain of
// dependent round trips (stage -> barrier -> bins pass -> barrier -> frames pass -> block sums) over 1280 pixels, and the 60 KB of
// LDS hold two workgroups per CU, 4.4 rounds of 2272 workgroups.  The next form is a strip walk -- one workgroup per clip and
// 128-frame strip, the frames window kept in registers -- not a faster tile.)
__device__ __forceinline__ void ssim_hpass5(const float* xs, const float* ys, int ld, int F, int rows, const SsimWin& w, float* hm) {
    const int n = rows * F;
    SsimDiv dv(F);
    for (int i = threadIdx.x; i < n; i += 256, dv.next()) {
        const int r = dv.q, c = dv.m;
        const float* xr = xs + r * ld + c;
        const float* yr = ys + r * ld + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int j = 0; j < SSIM_W; ++j) {
            const float g = w.g[j], x = xr[j], y = yr[j];
            a0 = fmaf(g, x, a0);
            a1 = fmaf(g, y, a1);
            a2 = fmaf(g, x * x, a2);
            a3 = fmaf(g, y * y, a3);
            a4 = fmaf(g, x * y, a4);
        }
        hm[i] = a0; hm[n + i] = a1; hm[2 * n + i] = a2; hm[3 * n + i] = a3; hm[4 * n + i] = a4;
    }
}

__device__ __forceinline__ SsimStats ssim_vpass5(const float* hm, int n, int row, int col, int F, const SsimWin& w) {
    SsimStats s = {0.f, 0.f, 0.f, 0.f, 0.f};
    const float* p = hm + row * F + col;
#pragma unroll
    for (int i = 0; i < SSIM_W; ++i) {
        const float g = w.g[i];
        s.mu1 = fmaf(g, p[i * F], s.mu1);
        s.mu2 = fmaf(g, p[n + i * F], s.mu2);
        s.e11 = fmaf(g, p[2 * n + i * F], s.e11);
        s.e22 = fmaf(g, p[3 * n + i * F], s.e22);
        s.e12 = fmaf(g, p[4 * n + i * F], s.e12);
    }
    return s;
}

// the same for the three cotangent maps of the backward's second stage
__device__ __forceinline__ void ssim_hpass3(const float* ga, const float* gb, const float* gc, int ld, int F, int rows,
                                            const SsimWin& w, float* hm) {
    const int n = rows * F;
    SsimDiv dv(F);
    for (int i = threadIdx.x; i < n; i += 256, dv.next()) {
        const int r = dv.q, c = dv.m;
        const int base = r * ld + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < SSIM_W; ++j) {
            const float g = w.g[j];
            a0 = fmaf(g, ga[base + j], a0);
            a1 = fmaf(g, gb[base + j], a1);
            a2 = fmaf(g, gc[base + j], a2);
        }
        hm[i] = a0; hm[n + i] = a1; hm[2 * n + i] = a2;
    }
}

__device__ __forceinline__ void ssim_vpass3(const float* hm, int n, int row, int col, int F, const SsimWin& w, float& fa, float& fb,
                                            float& fc) {
    const float* p = hm + row * F + col;
    fa = fb = fc = 0.f;
#pragma unroll
    for (int i = 0; i < SSIM_W; ++i) {
        const float g = w.g[i];
        fa = fmaf(g, p[i * F], fa);
        fb = fmaf(g, p[n + i * F], fb);
        fc = fmaf(g, p[2 * n + i * F], fc);
    }
}

// dynamic LDS of the stencil kernels: `tiles` staged images of rows x (F + 10) followed by `planes` filtered planes of rows x F
#define SSIM_ROWS (SSIM_TT + 2 * SSIM_R)
static size_t ssim_lds_bytes(int F, int tiles, int planes) {
    return sizeof(float) * (size_t)SSIM_ROWS * ((size_t)tiles * (F + 2 * SSIM_R) + (size_t)planes * F);
}

// stage x,y rows [t0-halo, t0+TT+halo) x cols [-5, F+5) into LDS (zero outside the image).  The thread -> element map follows the
// source's contiguous axis: the decoder's mel is read in place as [B,80,T] (frames contiguous, st < sf), where a bins-fastest
// map makes every lane of a load touch its own cache line.  All loads of a tile are issued before the first LDS store.
#define SSIM_STAGE_IT ((SSIM_ROWS_ * (SSIM_FMAX + 2 * SSIM_R) + 255) / 256)
__device__ __forceinline__ void ssim_stage(const float* p, long sb, long st, long sf, int b, int T, int F, int t_first, int rows,
                                           float bias, float* dst, int ld) {
    const int n = rows * ld;
    const bool tmaj = st < sf;                    // wave-uniform
    const float* pb = p + (long)b * sb;
    float v[SSIM_STAGE_IT];
    int di[SSIM_STAGE_IT];
    SsimDiv dv(tmaj ? rows : ld);
#pragma unroll
    for (int k = 0; k < SSIM_STAGE_IT; ++k, dv.next()) {
        const int i = threadIdx.x + 256 * k;
        const int r = tmaj ? dv.m : dv.q, c = tmaj ? dv.q : dv.m;
        const int t = t_first + r, f = c - SSIM_R;
        const bool ok = i < n && t >= 0 && t < T && f >= 0 && f < F;
        di[k] = i < n ? r * ld + c : -1;
        v[k] = ok ? pb[(long)t * st + (long)f * sf] + bias : 0.f;
    }
#pragma unroll
    for (int k = 0; k < SSIM_STAGE_IT; ++k)
        if (di[k] >= 0) dst[di[k]] = v[k];
}

__global__ __launch_bounds__(256) void svb_ssim_fwd_kernel(const float* pred, long psb, long pst, long psf, const float* tgt,
                                                           long tsb, long tst, long tsf, float* out, int B, int T, int F,
                                                           float bias, SsimWin w) {
    HIP_DYNAMIC_SHARED(float, ssim_smem)
    float* xs = ssim_smem;
    float* ys = xs + SSIM_ROWS * (F + 2 * SSIM_R);
    float* hm = ys + SSIM_ROWS * (F + 2 * SSIM_R);
    const int hn = SSIM_ROWS * F;
    const int b = blockIdx.y, t0 = blockIdx.x * SSIM_TT;
    const int ld = F + 2 * SSIM_R, rows = SSIM_TT + 2 * SSIM_R;
    ssim_stage(pred, psb, pst, psf, b, T, F, t0 - SSIM_R, rows, bias, xs, ld);
    ssim_stage(tgt, tsb, tst, tsf, b, T, F, t0 - SSIM_R, rows, bias, ys, ld);
    __syncthreads();
    ssim_hpass5(xs, ys, ld, F, rows, w, hm);
    __syncthreads();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    for (SsimDiv dv(F); dv.q < SSIM_TT; dv.next()) {
        const int r = dv.q, c = dv.m;
        const int t = t0 + r;
        if (t >= T) continue;
        const SsimStats s = ssim_vpass5(hm, hn, r, c, F, w);
        const float mu1_sq = s.mu1 * s.mu1, mu2_sq = s.mu2 * s.mu2, mu12 = s.mu1 * s.mu2;
        const float s1 = s.e11 - mu1_sq, s2 = s.e22 - mu2_sq, s12 = s.e12 - mu12;
        out[((long)b * T + t) * F + c] = ((2.f * mu12 + C1) * (2.f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    }
}

// backward stage 1: per-pixel cotangents of (mu1, E[xx], E[xy]) given d(map)
__global__ __launch_bounds__(256) void svb_ssim_bwd1_kernel(const float* pred, long psb, long pst, long psf, const float* tgt,
                                                            long tsb, long tst, long tsf, const float* dmap, float* gws, int B,
                                                            int T, int F, float bias, SsimWin w) {
    HIP_DYNAMIC_SHARED(float, ssim_smem)
    float* xs = ssim_smem;
    float* ys = xs + SSIM_ROWS * (F + 2 * SSIM_R);
    float* hm = ys + SSIM_ROWS * (F + 2 * SSIM_R);
    const int hn = SSIM_ROWS * F;
    const int b = blockIdx.y, t0 = blockIdx.x * SSIM_TT;
    const int ld = F + 2 * SSIM_R, rows = SSIM_TT + 2 * SSIM_R;
    ssim_stage(pred, psb, pst, psf, b, T, F, t0 - SSIM_R, rows, bias, xs, ld);
    ssim_stage(tgt, tsb, tst, tsf, b, T, F, t0 - SSIM_R, rows, bias, ys, ld);
    __syncthreads();
    ssim_hpass5(xs, ys, ld, F, rows, w, hm);
    __syncthreads();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const long plane = (long)B * T * F;
    for (SsimDiv dv(F); dv.q < SSIM_TT; dv.next()) {
        const int r = dv.q, c = dv.m;
        const int t = t0 + r;
        if (t >= T) continue;
        const SsimStats s = ssim_vpass5(hm, hn, r, c, F, w);
        const float mu1_sq = s.mu1 * s.mu1, mu2_sq = s.mu2 * s.mu2, mu12 = s.mu1 * s.mu2;
        const float s1 = s.e11 - mu1_sq, s2 = s.e22 - mu2_sq, s12 = s.e12 - mu12;
        const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
        const float inv = 1.f / (B1 * B2);
        const float m = A1 * A2 * inv;
        const long o = ((long)b * T + t) * F + c;
        const float dm = dmap[o];
        const float dA1 = A2 * inv, dA2 = A1 * inv, dB1 = -m / B1, dB2 = -m / B2;
        gws[o] = dm * (2.f * s.mu2 * dA1 - 2.f * s.mu2 * dA2 + 2.f * s.mu1 * dB1 - 2.f * s.mu1 * dB2);   // d/d mu1
        gws[plane + o] = dm * dB2;                                                                        // d/d E[xx]
        gws[2 * plane + o] = dm * 2.f * dA2;                                                              // d/d E[xy]
    }
}

// backward stage 2: dpred = G*g_mu1 + 2 x (G*g_e11) + y (G*g_e12)   (G symmetric 11x11, zero padded)
__global__ __launch_bounds__(256) void svb_ssim_bwd2_kernel(const float* pred, long psb, long pst, long psf, const float* tgt,
                                                            long tsb, long tst, long tsf, const float* gws, float* dpred, int B,
                                                            int T, int F, float bias, SsimWin w) {
    HIP_DYNAMIC_SHARED(float, ssim_smem)
    float* ga = ssim_smem;
    float* gb = ga + SSIM_ROWS * (F + 2 * SSIM_R);
    float* gc = gb + SSIM_ROWS * (F + 2 * SSIM_R);
    float* hm = gc + SSIM_ROWS * (F + 2 * SSIM_R);
    const int hn = SSIM_ROWS * F;
    const int b = blockIdx.y, t0 = blockIdx.x * SSIM_TT;
    const int ld = F + 2 * SSIM_R, rows = SSIM_TT + 2 * SSIM_R;
    const long plane = (long)B * T * F;
    const long sb = (long)T * F;
    ssim_stage(gws, sb, F, 1, b, T, F, t0 - SSIM_R, rows, 0.f, ga, ld);
    ssim_stage(gws + plane, sb, F, 1, b, T, F, t0 - SSIM_R, rows, 0.f, gb, ld);
    ssim_stage(gws + 2 * plane, sb, F, 1, b, T, F, t0 - SSIM_R, rows, 0.f, gc, ld);
    __syncthreads();
    ssim_hpass3(ga, gb, gc, ld, F, rows, w, hm);
    __syncthreads();
    for (SsimDiv dv(F); dv.q < SSIM_TT; dv.next()) {
        const int r = dv.q, c = dv.m;
        const int t = t0 + r;
        if (t >= T) continue;
        float fa, fb, fc;
        ssim_vpass3(hm, hn, r, c, F, w, fa, fb, fc);
        const float x = pred[(long)b * psb + (long)t * pst + (long)c * psf] + bias;
        const float y = tgt[(long)b * tsb + (long)t * tst + (long)c * tsf] + bias;
        dpred[((long)b * T + t) * F + c] = fa + 2.f * x * fb + y * fc;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused mel loss of one way (reference tasks/tts/fs2.py:143-175: l1_loss, ssim_loss, weights_nonzero_speech): the masked
// L1 term and the masked (1 - SSIM) term of (pred, target) in ONE pass, forward and backward.  Replaces, per way and
// step, ~17 stock elementwise / reduction launches forward (abs, sum, ne, float, sub, abs, mul, sum, div, rsub, ...)
// and ~12 backward, plus the [B,T,F] SSIM map, weight and dmap tensors.
//   w[b,t]   = any_f(target[b,t,f] != 0)                      (== target.abs().sum(-1).ne(0): a sum of magnitudes)
//   out[0]   = sum |pred - target| w / sum w ;  out[1] = sum (1 - ssim) w / sum w ;  out[2] = sum w  (over all pixels)
// Workgroup = the SSIM kernels' tile (16 frames x all bins); per-tile partial sums are written to `part` and summed in
// fixed order by a one-workgroup second launch (deterministic, no atomics).  terms: bit 0 = L1, bit 1 = SSIM.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mel_speech_rows(const float* tgt, long tsb, long tst, long tsf, int b, int t0, int T, int F,
                                                float* wrow) {
    if (threadIdx.x < SSIM_TT) wrow[threadIdx.x] = 0.f;
    __syncthreads();
    for (SsimDiv dv(F); dv.q < SSIM_TT; dv.next()) {
        const int r = dv.q, c = dv.m;
        const int t = t0 + r;
        if (t < T && tgt[(long)b * tsb + (long)t * tst + (long)c * tsf] != 0.f) wrow[r] = 1.f;   // (benign race: one value)
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void svb_mel_loss_fwd_kernel(const float* pred, long psb, long pst, long psf, const float* tgt,
                                                               long tsb, long tst, long tsf, float* part, int B, int T, int F,
                                                               float bias, int terms, SsimWin w) {
    HIP_DYNAMIC_SHARED(float, ssim_smem)
    float* xs = ssim_smem;
    float* ys = xs + SSIM_ROWS * (F + 2 * SSIM_R);
    float* hm = ys + SSIM_ROWS * (F + 2 * SSIM_R);
    const int hn = SSIM_ROWS * F;
    __shared__ float wrow[SSIM_TT];
    __shared__ float red[4];
    const int b = blockIdx.y, t0 = blockIdx.x * SSIM_TT;
    const int ld = F + 2 * SSIM_R, rows = SSIM_TT + 2 * SSIM_R;
    if (terms & 2) {
        ssim_stage(pred, psb, pst, psf, b, T, F, t0 - SSIM_R, rows, bias, xs, ld);
        ssim_stage(tgt, tsb, tst, tsf, b, T, F, t0 - SSIM_R, rows, bias, ys, ld);
    }
    mel_speech_rows(tgt, tsb, tst, tsf, b, t0, T, F, wrow);
    if (terms & 2) {
        ssim_hpass5(xs, ys, ld, F, rows, w, hm);
        __syncthreads();
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float a_l1 = 0.f, a_ss = 0.f, a_w = 0.f;
    for (SsimDiv dv(F); dv.q < SSIM_TT; dv.next()) {
        const int r = dv.q, c = dv.m;
        const int t = t0 + r;
        if (t >= T) continue;
        const float wt = wrow[r];
        a_w += wt;
        if (terms & 1) {
            const float p = pred[(long)b * psb + (long)t * pst + (long)c * psf];
            const float y = tgt[(long)b * tsb + (long)t * tst + (long)c * tsf];
            a_l1 += fabsf(p - y) * wt;
        }
        if (terms & 2) {
            const SsimStats s = ssim_vpass5(hm, hn, r, c, F, w);
            const float mu1_sq = s.mu1 * s.mu1, mu2_sq = s.mu2 * s.mu2, mu12 = s.mu1 * s.mu2;
            const float s1 = s.e11 - mu1_sq, s2 = s.e22 - mu2_sq, s12 = s.e12 - mu12;
            const float m = ((2.f * mu12 + C1) * (2.f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
            a_ss += (1.f - m) * wt;
        }
    }
    a_l1 = svb_block_sum<256>(a_l1, red);
    a_ss = svb_block_sum<256>(a_ss, red);
    a_w = svb_block_sum<256>(a_w, red);
    if (threadIdx.x == 0) {
        float* pp = part + 3 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        pp[0] = a_l1; pp[1] = a_ss; pp[2] = a_w;
    }
}

// out[k] = sum_j part[j][k] (k = 0..2) in a fixed order, then the two means
__global__ __launch_bounds__(256) void svb_mel_loss_final_kernel(const float* part, int n, float* out) {
    __shared__ float red[4];
    float a[3] = {0.f, 0.f, 0.f};
    for (int j = threadIdx.x; j < n; j += 256)
        for (int k = 0; k < 3; ++k) a[k] += part[3 * (size_t)j + k];
    for (int k = 0; k < 3; ++k) a[k] = svb_block_sum<256>(a[k], red);
    if (threadIdx.x == 0) { out[0] = a[0] / a[2]; out[1] = a[1] / a[2]; out[2] = a[2]; }
}

// backward stage 1 with d(map) = -gout[1] * w / sum w computed in place of a dmap tensor
__global__ __launch_bounds__(256) void svb_mel_loss_bwd1_kernel(const float* pred, long psb, long pst, long psf, const float* tgt,
                                                                long tsb, long tst, long tsf, const float* gout,
                                                                const float* sums, float* gws, int B, int T, int F, float bias,
                                                                SsimWin w) {
    HIP_DYNAMIC_SHARED(float, ssim_smem)
    float* xs = ssim_smem;
    float* ys = xs + SSIM_ROWS * (F + 2 * SSIM_R);
    float* hm = ys + SSIM_ROWS * (F + 2 * SSIM_R);
    const int hn = SSIM_ROWS * F;
    __shared__ float wrow[SSIM_TT];
    const int b = blockIdx.y, t0 = blockIdx.x * SSIM_TT;
    const int ld = F + 2 * SSIM_R, rows = SSIM_TT + 2 * SSIM_R;
    ssim_stage(pred, psb, pst, psf, b, T, F, t0 - SSIM_R, rows, bias, xs, ld);
    ssim_stage(tgt, tsb, tst, tsf, b, T, F, t0 - SSIM_R, rows, bias, ys, ld);
    mel_speech_rows(tgt, tsb, tst, tsf, b, t0, T, F, wrow);
    ssim_hpass5(xs, ys, ld, F, rows, w, hm);
    __syncthreads();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const long plane = (long)B * T * F;
    const float gs = -gout[1] / sums[2];
    for (SsimDiv dv(F); dv.q < SSIM_TT; dv.next()) {
        const int r = dv.q, c = dv.m;
        const int t = t0 + r;
        if (t >= T) continue;
        const SsimStats s = ssim_vpass5(hm, hn, r, c, F, w);
        const float mu1_sq = s.mu1 * s.mu1, mu2_sq = s.mu2 * s.mu2, mu12 = s.mu1 * s.mu2;
        const float s1 = s.e11 - mu1_sq, s2 = s.e22 - mu2_sq, s12 = s.e12 - mu12;
        const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
        const float inv = 1.f / (B1 * B2);
        const float m = A1 * A2 * inv;
        const long o = ((long)b * T + t) * F + c;
        const float dm = gs * wrow[r];
        const float dA1 = A2 * inv, dA2 = A1 * inv, dB1 = -m / B1, dB2 = -m / B2;
        gws[o] = dm * (2.f * s.mu2 * dA1 - 2.f * s.mu2 * dA2 + 2.f * s.mu1 * dB1 - 2.f * s.mu1 * dB2);
        gws[plane + o] = dm * dB2;
        gws[2 * plane + o] = dm * 2.f * dA2;
    }
}

// backward stage 2 (terms & 2) plus the L1 term's gout[0] * sgn(pred - target) * w / sum w (terms & 1)
__global__ __launch_bounds__(256) void svb_mel_loss_bwd2_kernel(const float* pred, long psb, long pst, long psf, const float* tgt,
                                                                long tsb, long tst, long tsf, const float* gout,
                                                                const float* sums, const float* gws, float* dpred, int B, int T,
                                                                int F, float bias, int terms, SsimWin w) {
    HIP_DYNAMIC_SHARED(float, ssim_smem)
    float* ga = ssim_smem;
    float* gb = ga + SSIM_ROWS * (F + 2 * SSIM_R);
    float* gc = gb + SSIM_ROWS * (F + 2 * SSIM_R);
    float* hm = gc + SSIM_ROWS * (F + 2 * SSIM_R);
    const int hn = SSIM_ROWS * F;
    __shared__ float wrow[SSIM_TT];
    const int b = blockIdx.y, t0 = blockIdx.x * SSIM_TT;
    const int ld = F + 2 * SSIM_R, rows = SSIM_TT + 2 * SSIM_R;
    const long plane = (long)B * T * F;
    const long sb = (long)T * F;
    if (terms & 2) {
        ssim_stage(gws, sb, F, 1, b, T, F, t0 - SSIM_R, rows, 0.f, ga, ld);
        ssim_stage(gws + plane, sb, F, 1, b, T, F, t0 - SSIM_R, rows, 0.f, gb, ld);
        ssim_stage(gws + 2 * plane, sb, F, 1, b, T, F, t0 - SSIM_R, rows, 0.f, gc, ld);
    }
    mel_speech_rows(tgt, tsb, tst, tsf, b, t0, T, F, wrow);
    if (terms & 2) {
        ssim_hpass3(ga, gb, gc, ld, F, rows, w, hm);
        __syncthreads();
    }
    const float gl = (terms & 1) ? gout[0] / sums[2] : 0.f;
    for (SsimDiv dv(F); dv.q < SSIM_TT; dv.next()) {
        const int r = dv.q, c = dv.m;
        const int t = t0 + r;
        if (t >= T) continue;
        const float p = pred[(long)b * psb + (long)t * pst + (long)c * psf];
        const float y = tgt[(long)b * tsb + (long)t * tst + (long)c * tsf];
        float d = 0.f;
        if (terms & 2) {
            float fa, fb, fc;
            ssim_vpass3(hm, hn, r, c, F, w, fa, fb, fc);
            d = fa + 2.f * (p + bias) * fb + (y + bias) * fc;
        }
        if (terms & 1) {
            const float df = p - y;
            d += (df > 0.f ? gl : (df < 0.f ? -gl : 0.f)) * wrow[r];
        }
        dpred[((long)b * T + t) * F + c] = d;
    }
}

// the stencil kernels' dynamic LDS can exceed the 64 KB default (F = 128: 95 KB)
template <typename K>
static void ssim_allow_lds(K kernel, bool* done) {
    if (!*done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        *done = true;
    }
}
#define SSIM_LAUNCH(kernel, grid, lds, stream, ...)                                         \
    do {                                                                                    \
        static bool attr_ = false;                                                          \
        ssim_allow_lds(kernel, &attr_);                                                     \
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, (hipStream_t)(stream), __VA_ARGS__); \
    } while (0)

static SsimWin make_window() {
    SsimWin w;
    float sum = 0.f;
    for (int i = 0; i < SSIM_W; ++i) {
        w.g[i] = (float)exp(-(double)((i - SSIM_R) * (i - SSIM_R)) / (2.0 * 1.5 * 1.5));
        sum += w.g[i];
    }
    for (int i = 0; i < SSIM_W; ++i) w.g[i] /= sum;
    return w;
}

extern "C" int svb_ssim_fwd(const float* pred, long psb, long pst, long psf, const float* tgt, long tsb, long tst, long tsf,
                            float* out_map, int B, int T, int F, float bias, void* stream) {
    if (!pred || !tgt || !out_map || B <= 0 || T <= 0 || F <= 0 || F > SSIM_FMAX || B > 65535) return SVB_ERR_ARG;
    dim3 grid(svb_cdiv(T, SSIM_TT), B);
    SSIM_LAUNCH(svb_ssim_fwd_kernel, grid, ssim_lds_bytes(F, 2, 5), stream, pred, psb, pst, psf, tgt, tsb, tst, tsf,
                       out_map, B, T, F, bias, make_window());
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_ssim_bwd(const float* pred, long psb, long pst, long psf, const float* tgt, long tsb, long tst, long tsf,
                            const float* dmap, float* dpred, float* workspace, int B, int T, int F, float bias, void* stream) {
    if (!pred || !tgt || !dmap || !dpred || !workspace || B <= 0 || T <= 0 || F <= 0 || F > SSIM_FMAX || B > 65535)
        return SVB_ERR_ARG;
    dim3 grid(svb_cdiv(T, SSIM_TT), B);
    const SsimWin w = make_window();
    SSIM_LAUNCH(svb_ssim_bwd1_kernel, grid, ssim_lds_bytes(F, 2, 5), stream, pred, psb, pst, psf, tgt, tsb, tst, tsf,
                       dmap, workspace, B, T, F, bias, w);
    SSIM_LAUNCH(svb_ssim_bwd2_kernel, grid, ssim_lds_bytes(F, 3, 3), stream, pred, psb, pst, psf, tgt, tsb, tst, tsf,
                       workspace, dpred, B, T, F, bias, w);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_mel_loss_fwd(const float* pred, long psb, long pst, long psf, const float* tgt, long tsb, long tst, long tsf,
                                float* out, float* part, int B, int T, int F, float bias, int terms, void* stream) {
    if (!pred || !tgt || !out || !part || B <= 0 || T <= 0 || F <= 0 || F > SSIM_FMAX || B > 65535 || !(terms & 3))
        return SVB_ERR_ARG;
    dim3 grid(svb_cdiv(T, SSIM_TT), B);
    SSIM_LAUNCH(svb_mel_loss_fwd_kernel, grid, ssim_lds_bytes(F, 2, 5), stream, pred, psb, pst, psf, tgt, tsb, tst, tsf,
                       part, B, T, F, bias, terms, make_window());
    hipLaunchKernelGGL(svb_mel_loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)part,
                       (int)(grid.x * grid.y), out);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_mel_loss_bwd(const float* pred, long psb, long pst, long psf, const float* tgt, long tsb, long tst, long tsf,
                                const float* gout, const float* sums, float* dpred, float* workspace, int B, int T, int F,
                                float bias, int terms, void* stream) {
    if (!pred || !tgt || !gout || !sums || !dpred || B <= 0 || T <= 0 || F <= 0 || F > SSIM_FMAX || B > 65535 || !(terms & 3) ||
        ((terms & 2) && !workspace))
        return SVB_ERR_ARG;
    dim3 grid(svb_cdiv(T, SSIM_TT), B);
    const SsimWin w = make_window();
    if (terms & 2)
        SSIM_LAUNCH(svb_mel_loss_bwd1_kernel, grid, ssim_lds_bytes(F, 2, 5), stream, pred, psb, pst, psf, tgt, tsb, tst,
                           tsf, gout, sums, workspace, B, T, F, bias, w);
    SSIM_LAUNCH(svb_mel_loss_bwd2_kernel, grid, ssim_lds_bytes(F, 3, 3), stream, pred, psb, pst, psf, tgt, tsb, tst, tsf,
                       gout, sums, (const float*)workspace, dpred, B, T, F, bias, terms, w);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
