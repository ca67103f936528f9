// period_ops.hip -- layout plumbing of the multi-period discriminator (reference modules/hifigan/hifigan.py:171-223,
// DiscriminatorP: x.view(b, c, t // period, period) followed by Conv2d((k,1), (stride,1), padding=(pad,0)) layers).
//
// The product keeps every feature map in the reference's own [B][C][H][p] layout and runs a (k,1) conv as a 1-D conv over the
// flattened [H*p] axis with dilation p (a tap moves by one ROW = p elements): clips are p times longer than in a period-major
// [B*p][C][H] layout, which is what the 64-wide tiles of the conv and weight-gradient kernels need (H is 10..50 in the deep
// layers).  A stride-s layer becomes stride 1 through a space-to-depth of the ROWS: channel (c, r) of the image holds rows
// s*h + r, so that  out[h] = sum_j W_j x[s*h + j - pad]  is a conv with ceil-many taps over s*C channels (tap q, phase r <->
// j = s*q + r + pad; slots without a tap carry zero weights).  This file holds that row space-to-depth and its inverse.
// HBM-bound copies: every element is read once and written once, runs of p contiguous floats on both sides.
#include "svb_common.h"
#include "../../include/svb_hip.h"

// forward  (inverse = 0): img[pl][r][row][w] = x[pl][s*(row - lead) + r][w]   (0 where the source row is outside [0, H))
// inverse  (inverse = 1): x[pl][hh][w]       = img[pl][hh % s][lead + hh / s][w]
// The element index is walked as a mixed-radix counter (w | row | r | plane, resp. w | h | plane) that advances by the grid's
// constant stride with carries: the flat form's three 64-bit divisions per element (~250 VALU instructions for one 4-byte copy)
// made this the one copy kernel of the vocoder step that was bound by instruction issue (34 us per launch, 75 launches).
__global__ __launch_bounds__(256) void svb_period_s2d_kernel(const float* src, float* dst, long planes, int H, int p, int s,
                                                             int lead, int R, int inverse) {
    const long total = inverse ? planes * H * p : planes * s * R * p;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b1 = inverse ? H : R;                 // radix of digit 1 (digit 0: w, radix p)
    const int b2 = inverse ? 1 : s;                 // radix of digit 2 (forward only)
    // digits of the first index and of the stride
    int w = (int)(i % p);
    long t = i / p;
    int d1 = (int)(t % b1);
    t /= b1;
    int d2 = inverse ? 0 : (int)(t % b2);
    long pl = inverse ? t : t / b2;
    const int sw = (int)(stride % p);
    long u = stride / p;
    const int s1 = (int)(u % b1);
    u /= b1;
    const int s2 = inverse ? 0 : (int)(u % b2);
    const long spl = inverse ? u : u / b2;
    for (; i < total; i += stride) {
        if (inverse) {
            const int hh = d1;
            const int q = hh / s, r = hh - q * s, row = lead + q;
            dst[i] = row < R ? src[((pl * s + r) * R + row) * p + w] : 0.f;
        } else {
            const int row = d1, r = d2;
            const int hh = s * (row - lead) + r;
            dst[i] = (row >= lead && hh < H) ? src[(pl * H + hh) * p + w] : 0.f;
        }
        w += sw;
        int c = 0;
        if (w >= p) { w -= p; c = 1; }
        d1 += s1 + c;
        c = 0;
        if (d1 >= b1) { d1 -= b1; c = 1; }
        if (!inverse) {
            d2 += s2 + c;
            c = 0;
            if (d2 >= b2) { d2 -= b2; c = 1; }
        }
        pl += spl + c;
    }
}

extern "C" int svb_period_s2d(const float* src, float* dst, long planes, int H, int p, int s, int lead, int R, int inverse,
                              void* stream) {
    if (!src || !dst || planes <= 0 || H <= 0 || p <= 0 || s <= 0 || lead < 0 || R <= lead) return SVB_ERR_ARG;
    const long total = inverse ? planes * H * p : planes * s * R * p;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(svb_period_s2d_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, planes, H, p, s, lead, R, inverse);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ---- the WEIGHT side of a strided period conv (round 6).  The stride-1 conv over the row space-to-depth image uses the kernel
//   w2[co][ci * s + r][q] = v[co][ci][j],  j = s * q + r + front   (0 where j is outside [0, k): the slots no tap falls on),
// `front` = padding + s * q_min <= 0.  forward (inverse = 0) writes w2 from v; inverse = 1 ADDS (accumulate) or writes the gradient
// of v gathered from the gradient of w2 -- every element of v sits in exactly one slot.  Tiny tensors (<= 3 M elements): one
// thread per element of the output side.
__global__ __launch_bounds__(256) void svb_period_weight_kernel(const float* src, float* dst, int cout, int cin, int k, int s,
                                                                int taps, int front, int inverse, int accumulate) {
    const long total = inverse ? (long)cout * cin * k : (long)cout * cin * s * taps;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (inverse) {
        const int j = (int)(i % k);
        const long t = i / k;
        const int ci = (int)(t % cin);
        const long co = t / cin;
        const int jj = j - front, q = jj / s, r = jj - q * s;
        const float gv = src[(co * cin * s + (long)ci * s + r) * taps + q];
        dst[i] = accumulate ? dst[i] + gv : gv;
    } else {
        const int q = (int)(i % taps);
        const long t = i / taps;
        const int cr = (int)(t % (cin * s));
        const long co = t / (cin * s);
        const int ci = cr / s, r = cr - ci * s;
        const int j = s * q + r + front;
        dst[i] = (j >= 0 && j < k) ? src[(co * cin + ci) * k + j] : 0.f;
    }
}

extern "C" int svb_period_weight(const float* src, float* dst, int cout, int cin, int k, int s, int taps, int front, int inverse,
                                 int accumulate, void* stream) {
    if (!src || !dst || cout <= 0 || cin <= 0 || k <= 0 || s <= 0 || taps <= 0 || front > 0 || s * taps + front < k) return SVB_ERR_ARG;
    const long total = inverse ? (long)cout * cin * k : (long)cout * cin * s * taps;
    hipLaunchKernelGGL(svb_period_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       cout, cin, k, s, taps, front, inverse, accumulate);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
