// critic_tower.hip -- host-side executor of one window's tower of the mel critic (reference modules/voice_conversion/
// multi_window_disc.py:14-64: three [Conv2d 3x3 s2 + LeakyReLU -> Dropout2d -> InstanceNorm2d] blocks and the score layer) and of its
// backward: ONE C-ABI call per direction issues every launch of the tower through the library's own entry points, in the order
// neuralsvb_amd/functional.py:_CriticTowerFn issues them from Python (round 6; the gated stacks have had theirs since round 3,
// wn_stack.hip).  No new device code: the point is the host -- the step runs six towers per direction, 7 launches forward and up
// to 23 backward each, at 15 ... 25 us of Python / autograd / ctypes per launch against ~3 us from a C loop.  Forward results are
// those of the Python sequence bit for bit (same kernels, same arguments, same order); the backward's weight gradients are reduced
// in-stream right after their partials (the Python pass may defer its reduces to the end of the pass: same sums, same order).
#include <string.h>
#include "svb_common.h"
#include "../../include/svb_hip.h"

#define CT_TRY(call)                   \
    do {                               \
        const int rc_ = (call);        \
        if (rc_ != SVB_OK) return rc_; \
    } while (0)

// one event per block and host thread / device (see wn_stack.hip for the reasoning)
#define SVB_CT_MAX_DEVICES 16
static thread_local hipEvent_t t_ct_events_all[SVB_CT_MAX_DEVICES][SVB_CT_MAX_BLOCKS];
static thread_local bool t_ct_events_ready[SVB_CT_MAX_DEVICES];

static int ct_events(hipEvent_t** set) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SVB_CT_MAX_DEVICES) return SVB_ERR_LAUNCH;
    if (!t_ct_events_ready[dev]) {
        for (int i = 0; i < SVB_CT_MAX_BLOCKS; ++i)
            if (hipEventCreateWithFlags(&t_ct_events_all[dev][i], hipEventDisableTiming) != hipSuccess) return SVB_ERR_LAUNCH;
        t_ct_events_ready[dev] = true;
    }
    *set = t_ct_events_all[dev];
    return SVB_OK;
}

static int ct_check(const SvbCriticTower* t) {
    if (!t || t->nb <= 0 || t->nb > SVB_CT_MAX_BLOCKS || t->N <= 0 || !t->x4 || !t->score_w || !t->score) return SVB_ERR_ARG;
    int H = t->H, W = t->W, C = t->C;
    for (int b = 0; b < t->nb; ++b) {
        const SvbCtBlock& k = t->blk[b];
        if (H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || k.cout <= 0 || !k.a_hi || !k.a_lo || !k.y4 || !k.out ||
            (k.gamma && !k.stats))
            return SVB_ERR_ARG;
        H /= 2; W /= 2; C = k.cout;
    }
    return SVB_OK;
}

extern "C" int svb_critic_tower_forward(const SvbCriticTower* t, void* stream) {
    CT_TRY(ct_check(t));
    int H = t->H, W = t->W, C = t->C;
    const float* x = t->x4;
    for (int b = 0; b < t->nb; ++b) {
        const SvbCtBlock& k = t->blk[b];
        const bool last = b + 1 == t->nb;
        const int Ho = H / 2, Wo = W / 2, P = Wo + 1;
        const int L = t->N * (Ho + 1) * P;
        const int offs[4] = {-P - 1, -P, -1, 0};
        SvbConvEpilogue e;
        memset(&e, 0, sizeof(e));
        e.bias = k.bias;
        e.out_act = t->has_slope ? SVB_ACT_LRELU : SVB_ACT_NONE;
        e.out_slope = t->has_slope ? t->slope : 0.f;
        e.force_cfg = k.cfg_fwd;
        CT_TRY(svb_conv1d_taps_bf16x3(x, k.a_hi, k.a_lo, k.y4, 1, 4 * C, k.cout, L, L, 4, offs, &e, stream));
        CT_TRY(svb_crop_drop_inorm_fwd(k.y4, k.keep, k.gamma, k.beta, k.eps, k.out, k.stats, t->N, k.cout, Ho, Wo, last ? 0 : 1, stream));
        x = k.out;
        H = Ho; W = Wo; C = k.cout;
    }
    // the last block's output is [C][N][H][W] memory: plane n of channel c at (c * N + n) * H * W
    return svb_plane_score_fwd(x, (long)H * W, (long)t->N * H * W, t->score_w, t->score_b, t->score, t->N, C, H * W, stream);
}

// weight gradient of one block: the derived 2x2 kernel's tap pairs {0,1} and {2,3} as two 2-tap gradients (+ the bias gradient out of
// the first), reduced in-stream, then gathered back into the 3x3 kernel's gradient buffer
static int ct_wgrad(const SvbCriticTower* t, const SvbCtBackward* g, const SvbCtBlock& k, const float* x, int C, int L, int P,
                    void* st) {
    const float a_slope = t->has_slope ? t->slope : 0.f;
    const float* yact = t->has_slope ? k.y4 : nullptr;
    const size_t rb_floats = (size_t)k.cout * 4 * C * 2;
    float* ra = g->ws;
    float* rb = ra + ((rb_floats + 15) & ~(size_t)15);
    float* part = rb + ((rb_floats + 15) & ~(size_t)15);
    const size_t part_cap = g->ws_floats - 2 * ((rb_floats + 15) & ~(size_t)15);
    if (g->ws_floats < 2 * ((rb_floats + 15) & ~(size_t)15)) return SVB_ERR_UNSUPPORTED;
    for (int half = 0; half < 2; ++half) {
        const int pad = half == 0 ? P + 1 : 1;
        int ns = 0;
        const size_t nfl = svb_conv1d_wgrad_bf16x3_workspace_floats(1, k.cout, 4 * C, 1, L, 2, 1, pad, 1, &ns);
        if (!nfl || ns <= 0) return SVB_ERR_UNSUPPORTED;
        const bool bias = half == 0 && k.d_bias != nullptr;
        const size_t need = ((nfl + 15) & ~(size_t)15) + (bias ? (((size_t)ns * k.cout + 15) & ~(size_t)15) : 0);
        if (need > part_cap) return SVB_ERR_UNSUPPORTED;
        float* bias_part = bias ? part + ((nfl + 15) & ~(size_t)15) : nullptr;
        CT_TRY(svb_conv1d_wgrad_bf16x3(k.dy4, x, part, 1, k.cout, 4 * C, 1, L, L, 2, 1, pad, 1, yact, a_slope, nullptr, 0.f, ns, bias_part,
                                       st));
        // (accumulate = 2: the weight rows overwrite ra / rb, the bias gradient adds into its `.grad` buffer)
        CT_TRY(svb_wgrad_reduce(part, ns, nullptr, nullptr, half == 0 ? ra : rb, nullptr, k.cout, 4 * C * 2, 0, bias ? 2 : 0, bias_part,
                                bias ? k.d_bias : nullptr, st));
    }
    return svb_s2_weight_bwd(ra, rb, k.d_weight, k.cout, C, 1, st);
}

extern "C" int svb_critic_tower_backward(const SvbCriticTower* t, const SvbCtBackward* g, void* stream, void* side_stream) {
    CT_TRY(ct_check(t));
    if (!g || !g->ds || !g->dh) return SVB_ERR_ARG;
    hipEvent_t* events = nullptr;
    CT_TRY(ct_events(&events));
    hipStream_t main = (hipStream_t)stream;
    hipStream_t side = side_stream ? (hipStream_t)side_stream : main;
    // planes of every block's input
    int Hs[SVB_CT_MAX_BLOCKS + 1], Ws[SVB_CT_MAX_BLOCKS + 1], Cs[SVB_CT_MAX_BLOCKS + 1];
    Hs[0] = t->H; Ws[0] = t->W; Cs[0] = t->C;
    for (int b = 0; b < t->nb; ++b) { Hs[b + 1] = Hs[b] / 2; Ws[b + 1] = Ws[b] / 2; Cs[b + 1] = t->blk[b].cout; }
    const int nb = t->nb;
    const int Hl = Hs[nb], Wl = Ws[nb], Cl = Cs[nb];
    const float* h = t->blk[nb - 1].out;
    CT_TRY(svb_plane_score_bwd(g->ds, g->ds_stride, h, (long)Hl * Wl, (long)t->N * Hl * Wl, t->score_w, g->dh, g->d_score_w, g->d_score_b,
                               t->N, Cl, Hl * Wl, stream));
    const float* dout = g->dh;
    for (int b = nb - 1; b >= 0; --b) {
        const SvbCtBlock& k = t->blk[b];
        const bool last = b + 1 == nb;
        const int C = Cs[b], Ho = Hs[b + 1], Wo = Ws[b + 1], P = Wo + 1;
        const int L = t->N * (Ho + 1) * P;
        if (!k.dy4 || (k.gamma && !k.dgb)) return SVB_ERR_ARG;
        // (last block: dh is the [N][C][Ho][Wo] view of [C][N][Ho][Wo] memory; earlier blocks: the next block's input gradient in its
        //  space-to-depth layout, strides ignored)
        CT_TRY(svb_crop_drop_inorm_bwd(dout, last ? (long)Ho * Wo : 0, last ? (long)t->N * Ho * Wo : 0, last ? (long)Wo : 0, last ? 1 : 0,
                                       k.y4, k.keep, k.gamma, k.stats, k.dy4, k.dgb, t->N, k.cout, Ho, Wo, last ? 0 : 1, stream));
        const float* x = b == 0 ? t->x4 : t->blk[b - 1].out;
        if (k.d_weight) {
            if (side != main) {
                if (hipEventRecord(events[b], main) != hipSuccess) return SVB_ERR_LAUNCH;
                if (hipStreamWaitEvent(side, events[b], 0) != hipSuccess) return SVB_ERR_LAUNCH;
            }
            CT_TRY(ct_wgrad(t, g, k, x, C, L, P, side));
        } else if (k.d_bias) {
            return SVB_ERR_ARG;                      // (a bias gradient without its weight gradient: the caller's per-op path)
        }
        if (k.dx4) {
            if (!k.b_hi || !k.b_lo) return SVB_ERR_ARG;
            const int noffs[4] = {P + 1, P, 1, 0};
            SvbConvEpilogue e;
            memset(&e, 0, sizeof(e));
            e.in_gate = t->has_slope ? k.y4 : nullptr;
            e.in_slope = t->has_slope ? t->slope : 0.f;
            e.force_cfg = k.cfg_bwd;
            CT_TRY(svb_conv1d_taps_bf16x3(k.dy4, k.b_hi, k.b_lo, k.dx4, 1, k.cout, 4 * C, L, L, 4, noffs, &e, stream));
            dout = k.dx4;
        } else {
            if (b > 0) return SVB_ERR_ARG;           // (only the tower's input may go without a gradient)
        }
    }
    return SVB_OK;
}
