// dtw.hip -- the offline F0 alignment of the binarizer on the GPU, batched over amateur / professional pairs.
//
//   reference: data_gen/singing/binarize_para.py:168-185 (get_pitch_align, 'EHSADTW')
//              modules/voice_conversion/dtw/enhance_sadtw.py:18-113 (cal_hist_of_f0, cal_hist_dist, EHSADTW)
//              modules/voice_conversion/dtw/align.py:8-37          (time_warp, align_from_distances)
//
// Stages (all pairs of a batch at once; tracks padded to L frames, len[] = real lengths):
//   1. svb_f0_shape_hist   per frame t the normalised histogram of slope classes of the track around t: 8 time windows
//                          (+-64 frames, scaled by the length ratio for the target track) x 6 slope regions; fp64 slopes and
//                          thresholds as in the reference's Python floats, the normalised counts rounded to fp32 (torch.tensor)
//   2. svb_hist_cost       chi-square distance of two histogram sets, fp32: cost[t][s] = sum_m 0.5*(hb-ha)^2 / (hb+ha+1e-8)
//   3. svb_dtw_accumulate  time_warp: D[0][0] = 0, first row / column inf, D[i][j] = cost[i][j] + min(D[i-1][j], D[i][j-1],
//                          D[i-1][j-1]) in fp32 -- bit-exact.  One workgroup per pair sweeps the anti-diagonals (cells of a
//                          diagonal are independent); the last two diagonals live in LDS, one barrier per diagonal; the
//                          arg-min direction of every cell is kept as a byte
//   4. svb_dtw_backtrack   align_from_distances: from (N-1, M-1) follow the first-minimum neighbour in the reference's order
//                          (i-1,j), (i,j-1), (i-1,j-1) while i > 0 and j > 0; result[i] = j (rows never visited stay 0)
#include "svb_common.h"
#include "../../include/svb_hip.h"
#include <math.h>

#define SVBD_BINS 48
#define SVBD_MAXL 4096

__global__ __launch_bounds__(256) void svb_f0_shape_hist_kernel(const double* f0, const int* len, const double* scale, float* hist,
                                                                int L) {
    const int p = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    const int T = len[p];
    if (t >= L) return;
    float* h = hist + ((size_t)p * L + t) * SVBD_BINS;
    if (t >= T) {
        for (int k = 0; k < SVBD_BINS; ++k) h[k] = 0.f;
        return;
    }
    const double* f = f0 + (size_t)p * L;
    const double sc = scale[p], ft = f[t];
    const int wl[8] = {-64, -48, -32, -16, 0, 16, 32, 48}, wr[8] = {-48, -32, -16, 0, 16, 32, 48, 64};
    int cnt[SVBD_BINS];
    for (int k = 0; k < SVBD_BINS; ++k) cnt[k] = 0;
    int total = 0;
    for (int w = 0; w < 8; ++w) {
        int rl = (int)((double)wl[w] * sc), rr = (int)((double)wr[w] * sc);       // int(): truncation toward zero
        if (rl == 0) rl = 1;
        const int lb = min(max(0, rl + t), T), rb = min(max(0, rr + t), T);
        const double wgt = (w == 0 || w == 7) ? 0.5 : ((w == 1 || w == 6) ? 0.75 : ((w == 2 || w == 5) ? 0.9 : 1.0));
        for (int i = lb; i < rb; ++i) {
            const double diff = f[i] - ft;
            double tan_i = diff / (double)(i - t);
            if (wgt != 1.0) tan_i *= wgt;
            const double a = fabs(tan_i);
            const int up = diff >= 0.0;
            int region;
            if (a < 0.57735) region = up ? 2 : 3;
            else if (a < 1.73205) region = up ? 1 : 4;
            else region = up ? 0 : 5;              // (a NaN slope cannot occur: i != t and the tracks are finite)
            ++cnt[w * 6 + region];
            ++total;
        }
    }
    for (int k = 0; k < SVBD_BINS; ++k) h[k] = total > 0 ? (float)((double)cnt[k] / (double)total) : 0.f;
}

__global__ __launch_bounds__(256) void svb_hist_cost_kernel(const float* ha, const int* len_a, const float* hb, const int* len_b,
                                                            float* cost, int La, int Lb) {
    const int p = blockIdx.z;
    const int s = blockIdx.x * 64 + (threadIdx.x & 63), t = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (s >= La || t >= Lb) return;
    float acc = 0.f;
    if (s < len_a[p] && t < len_b[p]) {
        const float* a = ha + ((size_t)p * La + s) * SVBD_BINS;
        const float* b = hb + ((size_t)p * Lb + t) * SVBD_BINS;
        for (int m = 0; m < SVBD_BINS; ++m) {
            const float mi = b[m] - a[m], pl = b[m] + a[m];
            acc += (0.5f * (mi * mi)) / (pl + 0.00000001f);
        }
    }
    cost[((size_t)p * Lb + t) * La + s] = acc;
}

// One workgroup per pair.  N = rows (target frames), M = columns (source frames); cost / dtw / dir have pitch La.
__global__ __launch_bounds__(1024) void svb_dtw_accumulate_kernel(const float* cost, const int* len_b, const int* len_a, float* dtw,
                                                                  unsigned char* dir, int La, int Lb) {
    __shared__ float diag[3][SVBD_MAXL];
    const int p = blockIdx.x;
    const int N = len_b[p], M = len_a[p];
    const float* c = cost + (size_t)p * Lb * La;
    float* D = dtw ? dtw + (size_t)p * Lb * La : nullptr;
    unsigned char* R = dir + (size_t)p * Lb * La;
    for (int d = 0; d <= N + M - 2; ++d) {
        float* cur = diag[d % 3];
        const float* d1 = diag[(d + 2) % 3];     // diagonal d-1
        const float* d2 = diag[(d + 1) % 3];     // diagonal d-2
        const int i_lo = max(0, d - (M - 1)), i_hi = min(N - 1, d);
        for (int i = i_lo + threadIdx.x; i <= i_hi; i += 1024) {
            const int j = d - i;
            float v;
            unsigned char r = 0;
            if (i == 0) v = j == 0 ? 0.f : INFINITY;
            else if (j == 0) v = INFINITY;
            else {
                const float a = d1[i - 1], b = d1[i], cc = d2[i - 1];        // D[i-1][j], D[i][j-1], D[i-1][j-1]
                float mn = a;
                if (b < mn) { mn = b; r = 1; }
                if (cc < mn) { mn = cc; r = 2; }
                v = c[(size_t)i * La + j] + mn;
            }
            cur[i] = v;
            if (D) D[(size_t)i * La + j] = v;
            R[(size_t)i * La + j] = r;
        }
        __syncthreads();
    }
}

__global__ void svb_dtw_backtrack_kernel(const unsigned char* dir, const int* len_b, const int* len_a, int64_t* align, int La, int Lb) {
    const int p = blockIdx.x;
    const int N = len_b[p], M = len_a[p];
    int64_t* out = align + (size_t)p * Lb;
    for (int i = threadIdx.x; i < Lb; i += blockDim.x) out[i] = 0;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned char* R = dir + (size_t)p * Lb * La;
    int i = N - 1, j = M - 1;
    while (i > 0 && j > 0) {
        out[i] = j;
        const unsigned char r = R[(size_t)i * La + j];
        if (r == 0) --i;
        else if (r == 1) --j;
        else { --i; --j; }
    }
}

extern "C" int svb_f0_shape_hist(const double* f0, const int* len, const double* scale, float* hist, int P, int L, void* stream) {
    if (!f0 || !len || !scale || !hist || P <= 0 || L <= 0 || P > 65535) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_f0_shape_hist_kernel, dim3((L + 255) / 256, P), dim3(256), 0, (hipStream_t)stream, f0, len, scale, hist, L);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_hist_cost(const float* ha, const int* len_a, const float* hb, const int* len_b, float* cost, int P, int La,
                             int Lb, void* stream) {
    if (!ha || !hb || !len_a || !len_b || !cost || P <= 0 || La <= 0 || Lb <= 0 || P > 65535 || (Lb + 3) / 4 > 65535)
        return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_hist_cost_kernel, dim3((La + 63) / 64, (Lb + 3) / 4, P), dim3(256), 0, (hipStream_t)stream, ha, len_a, hb,
                       len_b, cost, La, Lb);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_dtw_align(const float* cost, const int* len_b, const int* len_a, float* dtw, unsigned char* dir, int64_t* align,
                             int P, int La, int Lb, void* stream) {
    if (!cost || !len_a || !len_b || !dir || !align || P <= 0 || La <= 0 || Lb <= 0) return SVB_ERR_ARG;
    if (Lb > SVBD_MAXL) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_dtw_accumulate_kernel, dim3(P), dim3(1024), 0, (hipStream_t)stream, cost, len_b, len_a, dtw, dir, La, Lb);
    SVB_CHECK_LAUNCH();
    hipLaunchKernelGGL(svb_dtw_backtrack_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, dir, len_b, len_a, align, La, Lb);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
