// optim.hip -- gradient clipping + AdamW of one optimizer as TWO launches over flat buffers.
//
// Replaces, per optimizer step, torch.nn.utils.clip_grad_norm_ (reference tasks/singing/svb_vae_task.py:390-404 ->
// _foreach_norm + stack + vector_norm + clamp + _foreach_mul_ over ~200 tensors) and torch.optim.AdamW(fused) (reference
// :84-118; its Python side walks every parameter of the group each step): 1.9 ms of HOST time per train step on a step the host
// bounds (tools/host_split.py, round 4), against two kernel launches here.  The Trainer keeps the parameters, gradients and both
// moments of an optimizer in four flat fp32 buffers with one offset table (utils/flat_optim.py), so the update is elementwise.
//
//   coef  = min(1, max_norm / (||g|| + 1e-6))                      (clip_grad_norm_, error_if_nonfinite=False; max_norm <= 0: 1)
//   g'    = coef * g
//   p    *= 1 - lr * wd
//   m     = beta1 * m + (1 - beta1) * g'          (torch: lerp(m, g', 1 - beta1))
//   v     = beta2 * v + (1 - beta2) * g'^2
//   p    -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)           bc1 = 1 - beta1^t, bc2 = 1 - beta2^t  (computed by the caller)
// HBM-bound: 16 B read + 12 B written per element.
#include "svb_common.h"

#define SVB_OPT_THREADS 256
#define SVB_OPT_MAX_PARTS 1024

__global__ __launch_bounds__(SVB_OPT_THREADS) void svb_sumsq_parts_kernel(const float* g, size_t n4, float* part) {
    __shared__ float red[SVB_OPT_THREADS / 64];
    float acc = 0.f;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * SVB_OPT_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * SVB_OPT_THREADS) {
        const float4 t = g4[i];
        acc += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    acc = svb_block_sum<SVB_OPT_THREADS>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

struct SvbAdamWArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    size_t n4;                 // float4 elements (the flat buffers are padded to 4)
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, max_norm;
    const float* part;         // squared-norm partials of g (svb_sumsq_parts_kernel), or NULL: no clipping
    int npart;
    float* norm_out;           // optional: ||g|| (what clip_grad_norm_ returns)
};

__global__ __launch_bounds__(SVB_OPT_THREADS) void svb_adamw_flat_kernel(SvbAdamWArgs a) {
    __shared__ float red[SVB_OPT_THREADS / 64];
    float coef = 1.f;
    if (a.part) {
        // every workgroup sums the (<= 1024) partials itself, in the same fixed order: no third launch, deterministic
        float s = 0.f;
        for (int i = threadIdx.x; i < a.npart; i += SVB_OPT_THREADS) s += a.part[i];
        s = svb_block_sum<SVB_OPT_THREADS>(s, red);
        const float norm = sqrtf(s);
        if (a.max_norm > 0.f) coef = fminf(1.f, a.max_norm / (norm + 1e-6f));
        if (a.norm_out && blockIdx.x == 0 && threadIdx.x == 0) a.norm_out[0] = norm;
    }
    const float decay = 1.f - a.lr * a.wd, step = a.lr / a.bc1;
    const float b1 = a.beta1, b2 = a.beta2, ob1 = 1.f - a.beta1, ob2 = 1.f - a.beta2;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    for (size_t i = (size_t)blockIdx.x * SVB_OPT_THREADS + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * SVB_OPT_THREADS) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        float pe[4] = {p.x, p.y, p.z, p.w}, me[4] = {m.x, m.y, m.z, m.w}, ve[4] = {v.x, v.y, v.z, v.w};
        const float ge[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gc = ge[e] * coef;
            float q = pe[e] * decay;
            me[e] = me[e] + ob1 * (gc - me[e]);                 // torch.lerp(m, g, 1 - beta1), weight < 0.5 form
            ve[e] = b2 * ve[e] + ob2 * gc * gc;
            const float denom = sqrtf(ve[e]) / a.bc2_sqrt + a.eps;
            pe[e] = q - step * (me[e] / denom);
        }
        (void)b1;
        p4[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
        m4[i] = make_float4(me[0], me[1], me[2], me[3]);
        v4[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
    }
}

extern "C" int svb_adamw_flat_workspace_floats(void) { return SVB_OPT_MAX_PARTS; }

extern "C" int svb_adamw_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float bias_correction1, float bias_correction2_sqrt, float max_norm,
                              float* workspace, float* norm_out, void* stream) {
    if (!p || !g || !m || !v || n == 0 || (n & 3) || bias_correction1 <= 0.f || bias_correction2_sqrt <= 0.f) return SVB_ERR_ARG;
    if ((max_norm > 0.f || norm_out) && !workspace) return SVB_ERR_ARG;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + SVB_OPT_THREADS - 1) / SVB_OPT_THREADS;
    if (blocks > SVB_OPT_MAX_PARTS) blocks = SVB_OPT_MAX_PARTS;
    SvbAdamWArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n4 = n4;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = bias_correction1; a.bc2_sqrt = bias_correction2_sqrt; a.max_norm = max_norm;
    a.part = nullptr; a.npart = 0; a.norm_out = norm_out;
    if (max_norm > 0.f || norm_out) {
        hipLaunchKernelGGL(svb_sumsq_parts_kernel, dim3((unsigned)blocks), dim3(SVB_OPT_THREADS), 0, (hipStream_t)stream, g, n4, workspace);
        SVB_CHECK_LAUNCH();
        a.part = workspace; a.npart = (int)blocks;
    }
    size_t ublocks = (n4 + SVB_OPT_THREADS - 1) / SVB_OPT_THREADS;
    if (ublocks > 4096) ublocks = 4096;
    hipLaunchKernelGGL(svb_adamw_flat_kernel, dim3((unsigned)ublocks), dim3(SVB_OPT_THREADS), 0, (hipStream_t)stream, a);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ---- gradients autograd handed over as tensors of their own -> their slices of the flat gradient buffer --------------------------
// A parameter whose gradient only ever arrives through torch's autograd (norm affines, biases of torch ops, embeddings ...) has
// `.grad = None` between steps, so AccumulateGrad adopts the incoming tensor instead of launching `grad += new` into a zeroed
// view (one small launch per parameter, ~30 per train step).  Before the flat AdamW step those tensors are copied into the flat
// buffer by ONE launch: up to SVB_GATHER_BATCH segments per launch, pointers by value in the kernel arguments.
#define SVB_GATHER_BATCH 48
struct SvbGatherBatch {
    const float* src[SVB_GATHER_BATCH];
    size_t off[SVB_GATHER_BATCH];
    size_t n[SVB_GATHER_BATCH];
};

__global__ __launch_bounds__(SVB_OPT_THREADS) void svb_gather_segments_kernel(SvbGatherBatch bt, float* dst) {
    const int s = blockIdx.y;
    const float* src = bt.src[s];
    float* d = dst + bt.off[s];
    const size_t n = bt.n[s];
    for (size_t i = (size_t)blockIdx.x * SVB_OPT_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * SVB_OPT_THREADS) d[i] = src[i];
}

extern "C" int svb_gather_segments(const void* const* src, const size_t* dst_off, const size_t* n, int count, float* dst, void* stream) {
    if (count < 0 || (count && (!src || !dst_off || !n || !dst))) return SVB_ERR_ARG;
    for (int s0 = 0; s0 < count; s0 += SVB_GATHER_BATCH) {
        const int c = count - s0 < SVB_GATHER_BATCH ? count - s0 : SVB_GATHER_BATCH;
        SvbGatherBatch bt;
        size_t nmax = 0;
        for (int s = 0; s < c; ++s) {
            if (!src[s0 + s] && n[s0 + s]) return SVB_ERR_ARG;
            bt.src[s] = static_cast<const float*>(src[s0 + s]);
            bt.off[s] = dst_off[s0 + s];
            bt.n[s] = n[s0 + s];
            if (bt.n[s] > nmax) nmax = bt.n[s];
        }
        for (int s = c; s < SVB_GATHER_BATCH; ++s) { bt.src[s] = nullptr; bt.off[s] = 0; bt.n[s] = 0; }
        size_t bx = (nmax + 4 * SVB_OPT_THREADS - 1) / (4 * SVB_OPT_THREADS);
        if (bx < 1) bx = 1;
        if (bx > 256) bx = 256;
        hipLaunchKernelGGL(svb_gather_segments_kernel, dim3((unsigned)bx, (unsigned)c), dim3(SVB_OPT_THREADS), 0, (hipStream_t)stream, bt, dst);
        SVB_CHECK_LAUNCH();
    }
    return SVB_OK;
}
