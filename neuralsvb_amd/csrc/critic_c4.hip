// critic_c4.hip -- the mel critic's FIRST block (reference modules/voice_conversion/multi_window_disc.py:14-22: Conv2d(1, 128, 3x3, s2) on
// the mel window) as streaming kernels (round 6).  In the space-to-depth layout that conv is a 4-tap conv from 4 planes to 128
// channels over L = N (H/2+1) (W/2+1) positions -- 2 x 16 MACs per output value against 4 bytes written: on the MFMA tiles (K padded
// from 4 to 16 channels, 64-row tiles for a 4-row gradient) the forward ran at 15 ... 19 TFLOP/s, the data gradient at 3 ... 9, the
// weight gradient at 2.4 ... 5.7 -- 2.4 / 1 / 1 TB/s of their HBM traffic.  Here a lane owns a position: the forward writes 128
// coalesced rows from 16 input values held in registers, the data gradient reads dy (and its gate) once per tap from L1, the weight
// gradient keeps 8 rows x 4 channels x 2 taps of partial sums per lane and block-reduces them.  Arithmetic: the bf16 hi + lo images
// the MFMA kernels read are summed back to ONE fp32 weight (w to 2^-17); activations stay fp32; fp32 FMA accumulation -- inside the
// bf16x3 kernels' error, not bit-identical to them (every caller of these shapes lands here, so the per-op and the executor paths agree).
#include "svb_common.h"
#include "conv1d_q.h"

__device__ __forceinline__ float c4_w(const unsigned short* hi, const unsigned short* lo, size_t i) {
    const unsigned h = (unsigned)hi[i] << 16, l = (unsigned)lo[i] << 16;
    return __builtin_bit_cast(float, h) + __builtin_bit_cast(float, l);
}

// ---- forward: y[co][p] = act(bias[co] + sum_{t, ci < 4} w[t][co][ci] x[ci][p + off[t]]), x zero outside [0, L)
struct C4FwdArgs {
    const float* x; const unsigned short *w_hi, *w_lo; const float* bias; float* y;
    int L, cout, ntap, act; float slope;
    int off[4];
};

__global__ __launch_bounds__(256) void svb_c4_fwd_kernel(C4FwdArgs a) {
    __shared__ float ws[128 * 16];                 // [co][tap][ci] of this workgroup's row block
    const int co0 = blockIdx.y * 128, nco = min(128, a.cout - co0);
    for (int i = threadIdx.x; i < nco * 16; i += 256) {
        const int co = i >> 4, t = (i >> 2) & 3, ci = i & 3;
        ws[i] = t < a.ntap ? c4_w(a.w_hi, a.w_lo, ((size_t)t * a.cout + co0 + co) * 16 + ci) : 0.f;
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.L) return;
    float xv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = p + (t < a.ntap ? a.off[t] : 0);
        const bool ok = t < a.ntap && q >= 0 && q < a.L;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) xv[4 * t + ci] = ok ? a.x[(size_t)ci * a.L + q] : 0.f;
    }
    const float neg = a.act == SVB_ACT_RELU ? 0.f : (a.act == SVB_ACT_LRELU ? a.slope : 1.f);
    for (int co = 0; co < nco; ++co) {
        const float4* w4 = reinterpret_cast<const float4*>(ws + co * 16);      // (uniform address: an LDS broadcast)
        float acc = a.bias ? a.bias[co0 + co] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 w = w4[t];
            acc = fmaf(w.x, xv[4 * t], acc);
            acc = fmaf(w.y, xv[4 * t + 1], acc);
            acc = fmaf(w.z, xv[4 * t + 2], acc);
            acc = fmaf(w.w, xv[4 * t + 3], acc);
        }
        a.y[(size_t)(co0 + co) * a.L + p] = acc > 0.f ? acc : acc * neg;
    }
}

int svb_c4_fwd_launch(const SvbConvQArgs& q, const SvbConvPlan& p, hipStream_t stream) {
    const int ntap = p.phase_start[1] - p.phase_start[0];
    if (q.B != 1 || q.Cin != 4 || q.G != 1 || p.n_phase != 1 || ntap < 1 || ntap > 4 || q.Tin != q.Tout || q.sx != 1 || q.out_stride != 1 ||
        q.in_gate || q.out_gate || q.residual || q.mask || q.xq || q.out_act == SVB_ACT_TANH || q.w_slab_rows != q.Cout)
        return SVB_ERR_UNSUPPORTED;
    C4FwdArgs a;
    a.x = q.x; a.w_hi = q.wq_hi; a.w_lo = q.wq_lo; a.bias = q.bias; a.y = q.y;
    a.L = q.Tin; a.cout = q.Cout; a.ntap = ntap; a.act = q.out_act; a.slope = q.out_slope;
    for (int t = 0; t < 4; ++t) a.off[t] = t < ntap ? p.tap_off[p.phase_start[0] + t] : 0;
    for (int t = 0; t < ntap; ++t)
        if (p.tap_w[p.phase_start[0] + t] != t) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_c4_fwd_kernel, dim3(svb_cdiv(a.L, 256), svb_cdiv(a.cout, 128)), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? SVB_OK : SVB_ERR_LAUNCH;
}

// ---- data gradient: dx[ci][p] = sum_{t, co} wT[t][ci][co] g(dy[co][p + off[t]]),  g(v) = v * lrelu'(gate) (gate optional), 4 outputs
struct C4BwdArgs {
    const float* dy; const float* gate; const unsigned short *w_hi, *w_lo; float* dx;
    int L, cin, ntap; float slope;
    int off[4];
};

__global__ __launch_bounds__(256) void svb_c4_bwd_kernel(C4BwdArgs a) {
    HIP_DYNAMIC_SHARED(float, wsb)                 // [co][tap][ci out]: cin * 16 floats
    // packed image: [tap][chunk = co / 16][row = ci out (4 rows)][16 co]
    for (int i = threadIdx.x; i < a.cin * 16; i += 256) {
        const int co = i >> 4, t = (i >> 2) & 3, ci = i & 3;
        wsb[i] = t < a.ntap ? c4_w(a.w_hi, a.w_lo, (((size_t)t * (a.cin >> 4) + (co >> 4)) * 4 + ci) * 16 + (co & 15)) : 0.f;
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.L) return;
    int q[4];
    bool ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        q[t] = p + (t < a.ntap ? a.off[t] : 0);
        ok[t] = t < a.ntap && q[t] >= 0 && q[t] < a.L;
        q[t] = ok[t] ? q[t] : p;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int co = 0; co < a.cin; ++co) {
        const float* row = a.dy + (size_t)co * a.L;
        const float* grow = a.gate ? a.gate + (size_t)co * a.L : nullptr;
        const float4* w4 = reinterpret_cast<const float4*>(wsb + co * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = ok[t] ? row[q[t]] : 0.f;
            if (grow) v *= svb_gate(grow[q[t]], a.slope);
            const float4 w = w4[t];
            acc[0] = fmaf(w.x, v, acc[0]);
            acc[1] = fmaf(w.y, v, acc[1]);
            acc[2] = fmaf(w.z, v, acc[2]);
            acc[3] = fmaf(w.w, v, acc[3]);
        }
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) a.dx[(size_t)ci * a.L + p] = acc[ci];
}

int svb_c4_bwd_launch(const SvbConvQArgs& q, const SvbConvPlan& p, hipStream_t stream) {
    const int ntap = p.phase_start[1] - p.phase_start[0];
    if (q.B != 1 || q.Cout != 4 || q.Cin % 16 || q.Cin > 1024 || q.G != 1 || p.n_phase != 1 || ntap < 1 || ntap > 4 || q.Tin != q.Tout ||
        q.sx != 1 || q.out_stride != 1 || q.bias || q.out_gate || q.residual || q.mask || q.xq || q.out_act != SVB_ACT_NONE ||
        q.w_slab_rows != 4)
        return SVB_ERR_UNSUPPORTED;
    C4BwdArgs a;
    a.dy = q.x; a.gate = q.in_gate; a.w_hi = q.wq_hi; a.w_lo = q.wq_lo; a.dx = q.y;
    a.L = q.Tin; a.cin = q.Cin; a.ntap = ntap; a.slope = q.in_slope;
    for (int t = 0; t < 4; ++t) a.off[t] = t < ntap ? p.tap_off[p.phase_start[0] + t] : 0;
    for (int t = 0; t < ntap; ++t)
        if (p.tap_w[p.phase_start[0] + t] != t) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_c4_bwd_kernel, dim3(svb_cdiv(a.L, 256)), dim3(256), (size_t)a.cin * 16 * sizeof(float), stream, a);
    return hipGetLastError() == hipSuccess ? SVB_OK : SVB_ERR_LAUNCH;
}

// ---- weight gradient: part[s][co][ci][j] = sum_{p in split s} g(a[co][p]) b[ci][p + j - pad],  j < 2, ci < 4; bias_part[s][co] = sum g(a)
struct C4WgArgs {
    const float* a; const float* a_gate; const float* b; float* part; float* bias_part;
    int L, CA, pad, nsplit; float slope;
};

#define C4_ROWS 8
__global__ __launch_bounds__(256) void svb_c4_wgrad_kernel(C4WgArgs a) {
    __shared__ float red[4][C4_ROWS * 9];
    const int co0 = blockIdx.x * C4_ROWS, s = blockIdx.y;
    const int per = (a.L + a.nsplit - 1) / a.nsplit;
    const int p0 = s * per, p1 = min(a.L, p0 + per);
    float acc[C4_ROWS][8], bs[C4_ROWS];
#pragma unroll
    for (int r = 0; r < C4_ROWS; ++r) {
        bs[r] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
    }
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        float xv[8];                                  // [ci][j]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = p + j - a.pad;
            const bool ok = q >= 0 && q < a.L;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) xv[2 * ci + j] = ok ? a.b[(size_t)ci * a.L + q] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < C4_ROWS; ++r) {
            const int co = min(co0 + r, a.CA - 1);
            float v = a.a[(size_t)co * a.L + p];
            if (a.a_gate) v *= svb_gate(a.a_gate[(size_t)co * a.L + p], a.slope);
            bs[r] += v;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] = fmaf(v, xv[e], acc[r][e]);
        }
    }
    // block sums: lanes of a wave by shuffles, the four waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < C4_ROWS; ++r) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            float v = e < 8 ? acc[r][e < 8 ? e : 0] : bs[r];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            if (lane == 0) red[wave][r * 9 + e] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < C4_ROWS * 9) {
        const int r = threadIdx.x / 9, e = threadIdx.x - 9 * r;
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const int co = co0 + r;
        if (co < a.CA) {
            if (e < 8) a.part[((size_t)s * a.CA + co) * 8 + e] = v;
            else if (a.bias_part) a.bias_part[(size_t)s * a.CA + co] = v;
        }
    }
}

// split count of this kernel for L positions: workgroups of ~2.5 k positions, at most 64 splits
int svb_c4_wgrad_nsplit(int L) {
    int ns = L / 2560;
    if (ns < 1) ns = 1;
    if (ns > 64) ns = 64;
    return ns;
}

bool svb_c4_wgrad_applies(int B, int CA, int CB, int groups, int k, int sx, int dil) {
    return B == 1 && CB == 4 && groups == 1 && k == 2 && sx == 1 && dil == 1 && CA >= 1;
}

int svb_c4_wgrad_launch(const float* a_t, const float* b_t, float* part, int CA, int L, int pad, const float* a_gate, float a_slope,
                        int nsplit, float* bias_part, hipStream_t stream) {
    C4WgArgs a;
    a.a = a_t; a.a_gate = a_gate; a.b = b_t; a.part = part; a.bias_part = bias_part;
    a.L = L; a.CA = CA; a.pad = pad; a.nsplit = nsplit; a.slope = a_slope;
    hipLaunchKernelGGL(svb_c4_wgrad_kernel, dim3(svb_cdiv(CA, C4_ROWS), nsplit), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? SVB_OK : SVB_ERR_LAUNCH;
}
