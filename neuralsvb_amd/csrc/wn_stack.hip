// wn_stack.hip -- host-side executor of the gated conv stack (`WN.forward`, reference modules/fastspeech/fs2_vae.py:61-91) and of
// its hand-scheduled backward: ONE C-ABI call issues every launch of the stack (4 per layer forward, 7 + the weight-gradient
// reduces backward) through the library's own entry points, in the order neuralsvb_amd/functional.py:_WNStackFn issues them from
// Python.  No new device code: the point is the host.  The step is bounded by the rate at which Python can issue launches on
// hosts slower than the GPU (~15 us per launch through autograd + ctypes; the two stacks of the VAE are ~130 launches per
// generator pass); a C loop issues the same launches at the runtime's own cost.  Results are bit-identical to the Python
// sequence by construction (same kernels, same arguments, same order per stream).
#include <string.h>
#include "svb_common.h"
#include "../../include/svb_hip.h"

// Weight gradients of a backward pass run on `side` (if given) beside the data-gradient chain on `stream`: before each of them
// the side stream catches up with the main one (an event per layer and direction), and the caller joins the side stream after
// the call.  Their split-K partials collect in `arena`; svb_wgrad_reduce_multi finishes the pending ones whenever the next
// does not fit and at the end of the stack (the capped deferral of kernels.py, inside the executor).
// (one event set per HOST THREAD and device: an event belongs to the device it was created on, and a record / wait pair must not
//  be interleaved with another caller's -- two host threads driving two streams, or two devices, each use their own set, so the
//  executor is re-entrant per stream as the rest of the C ABI is.  The set is handed down as a local pointer, never through a
//  process-wide variable.)
#define SVB_WN_MAX_DEVICES 16
static thread_local hipEvent_t t_wn_events_all[SVB_WN_MAX_DEVICES][2 * SVB_WN_MAX_LAYERS];
static thread_local bool t_wn_events_ready[SVB_WN_MAX_DEVICES];

static int wn_events(hipEvent_t** set) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SVB_WN_MAX_DEVICES) return SVB_ERR_LAUNCH;
    if (!t_wn_events_ready[dev]) {
        for (int i = 0; i < 2 * SVB_WN_MAX_LAYERS; ++i)
            if (hipEventCreateWithFlags(&t_wn_events_all[dev][i], hipEventDisableTiming) != hipSuccess) return SVB_ERR_LAUNCH;
        t_wn_events_ready[dev] = true;
    }
    *set = t_wn_events_all[dev];
    return SVB_OK;
}

#define WN_TRY(call)                 \
    do {                             \
        const int rc_ = (call);      \
        if (rc_ != SVB_OK) return rc_; \
    } while (0)

static int wn_check(const SvbWnStack* s) {
    if (!s || s->n_layers <= 0 || s->n_layers > SVB_WN_MAX_LAYERS || s->B <= 0 || s->C <= 0 || s->T <= 0 || s->k <= 0 ||
        s->dil_rate <= 0 || !s->x0 || !s->xin || !s->acts || (s->n_layers > 1 && !s->xbuf))
        return SVB_ERR_ARG;
    for (int i = 0; i < s->n_layers; ++i) {
        const int want = i + 1 < s->n_layers ? 2 * s->C : s->C;
        if (s->layer[i].rs_cout != want) return SVB_ERR_ARG;
    }
    return SVB_OK;
}

static inline const float* wn_x(const SvbWnStack* s, int i) {
    return i == 0 ? s->x0 : s->xbuf + (size_t)(i - 1) * s->B * s->C * s->T;
}

extern "C" int svb_wn_stack_forward(const SvbWnStack* s, float* rs_scratch, float* out, void* stream) {
    WN_TRY(wn_check(s));
    if (!rs_scratch || !out) return SVB_ERR_ARG;
    const size_t nC = (size_t)s->B * s->C * s->T;
    int dil = 1;
    for (int i = 0; i < s->n_layers; ++i) {
        const SvbWnLayer& L = s->layer[i];
        const bool last = i + 1 == s->n_layers;
        const int pad = (s->k * dil - dil) / 2;
        const float* x = wn_x(s, i);
        float* xin = s->xin + (size_t)i * 2 * nC;
        float* acts = s->acts + (size_t)i * nC;
        SvbConvEpilogue e;
        memset(&e, 0, sizeof(e));
        e.bias = L.in_bias;
        e.force_cfg = L.cfg_in_fwd;
        WN_TRY(svb_conv1d_forward_bf16x3(x, L.in_a_hi, L.in_a_lo, xin, s->B, s->C, 2 * s->C, 1, s->T, s->T, s->k, 1, pad, dil, &e,
                                         stream));
        WN_TRY(svb_wn_gate_fwd(xin, s->G, acts, nullptr, s->B, s->C, s->T, s->G ? s->g_channels : 0, i * 2 * s->C, stream));
        memset(&e, 0, sizeof(e));
        e.bias = L.rs_bias;
        e.force_cfg = L.cfg_rs_fwd;
        WN_TRY(svb_conv1d_forward_bf16x3(acts, L.rs_a_hi, L.rs_a_lo, rs_scratch, s->B, s->C, L.rs_cout, 1, s->T, s->T, 1, 1, 0, 1,
                                         &e, stream));
        float* x_new = last ? nullptr : s->xbuf + (size_t)i * nC;
        WN_TRY(svb_wn_res_skip(x, rs_scratch, s->mask, i == 0 ? nullptr : out, x_new, out, nullptr, s->B, s->C, s->T, last ? 1 : 0,
                               stream));
        dil *= s->dil_rate;
    }
    return SVB_OK;
}

// ---- backward ----------------------------------------------------------------------------------------------------------------
struct WnPending {
    SvbReduceDesc d[2 * SVB_WN_MAX_LAYERS];
    int n;
    size_t used;
};

static int wn_flush(WnPending& p, void* st) {
    if (p.n) WN_TRY(svb_wgrad_reduce_multi(p.d, p.n, st));
    p.n = 0;
    p.used = 0;
    return SVB_OK;
}

// one weight gradient with all of its results accumulated into gradient buffers: partials into the arena, reduce recorded
static int wn_wgrad(const SvbWnStack* s, const SvbWnBackward* b, WnPending& p, const float* a, const float* bt, int CA, int CB,
                    int k, int pad, int dil, const float* v, const float* g, float* dv, float* dg, float* db, void* st) {
    int ns = 0;
    const size_t nfl = svb_conv1d_wgrad_bf16x3_workspace_floats(s->B, CA, CB, 1, s->T, k, 1, pad, dil, &ns);
    if (!nfl || ns <= 0) return SVB_ERR_UNSUPPORTED;
    const size_t need = ((nfl + 15) & ~(size_t)15) + (((size_t)ns * CA + 15) & ~(size_t)15);
    if (need > b->arena_floats) return SVB_ERR_UNSUPPORTED;
    if (p.used + need > b->arena_floats || p.n == 2 * SVB_WN_MAX_LAYERS) WN_TRY(wn_flush(p, st));
    float* part = b->arena + p.used;
    float* bias_part = part + ((nfl + 15) & ~(size_t)15);
    p.used += need;
    WN_TRY(svb_conv1d_wgrad_bf16x3(a, bt, part, s->B, CA, CB, 1, s->T, s->T, k, 1, pad, dil, nullptr, 0.f, nullptr, 0.f, ns, bias_part,
                                   st));
    SvbReduceDesc& d = p.d[p.n++];
    d.part = part; d.v = v; d.g = g; d.dv = dv; d.dg = g ? dg : nullptr; d.bias_part = bias_part; d.db = db;
    d.nsplit = ns; d.rows = CA; d.rowlen = CB * k; d.weight_norm = g ? 1 : 0; d.accumulate = 1; d.row_start = 0;
    return SVB_OK;
}

extern "C" int svb_wn_stack_backward(const SvbWnStack* s, const SvbWnBackward* b, void* stream, void* side_stream) {
    WN_TRY(wn_check(s));
    if (!b || !b->dout || !b->drs || !b->dxin || !b->dacts || !b->dxm || !b->dx || !b->arena) return SVB_ERR_ARG;
    hipEvent_t* events = nullptr;
    WN_TRY(wn_events(&events));
    hipStream_t main = (hipStream_t)stream;
    hipStream_t side = side_stream ? (hipStream_t)side_stream : main;
    const size_t nC = (size_t)s->B * s->C * s->T;
    WnPending pend;
    pend.n = 0;
    pend.used = 0;
    int dils[SVB_WN_MAX_LAYERS];
    dils[0] = 1;
    for (int i = 1; i < s->n_layers; ++i) dils[i] = dils[i - 1] * s->dil_rate;
    bool have_dx = false;          // b->dx holds the gradient flowing into x_{i+1}
    for (int i = s->n_layers - 1; i >= 0; --i) {
        const SvbWnLayer& L = s->layer[i];
        const bool last = i + 1 == s->n_layers;
        const int dil = dils[i], pad = (s->k * dil - dil) / 2;
        const float* x = wn_x(s, i);
        const float* xin = s->xin + (size_t)i * 2 * nC;
        const float* acts = s->acts + (size_t)i * nC;
        const bool need_in_w = L.d_in_v != nullptr, need_rs_w = L.d_rs_v != nullptr;
        const bool need_dx = i > 0 || b->need_dx0;
        const float* drs;
        const float* dxm = nullptr;
        if (last) {
            drs = b->dout;
        } else {
            float* drs_w = b->drs + (size_t)i * 2 * nC;
            WN_TRY(svb_wn_res_skip_bwd(have_dx ? b->dx : nullptr, b->dout, s->mask, drs_w, b->dxm, nullptr, s->B, s->C, s->T, stream));
            drs = drs_w;
            dxm = b->dxm;
        }
        if (need_rs_w) {
            if (side != main) {
                if (hipEventRecord(events[2 * i], main) != hipSuccess) return SVB_ERR_LAUNCH;
                if (hipStreamWaitEvent(side, events[2 * i], 0) != hipSuccess) return SVB_ERR_LAUNCH;
            }
            WN_TRY(wn_wgrad(s, b, pend, drs, acts, L.rs_cout, s->C, 1, 0, 1, L.rs_g ? L.rs_v : nullptr, L.rs_g, L.d_rs_v, L.d_rs_g,
                            L.d_rs_b, side));
        }
        if (!(need_in_w || b->dG || need_dx)) {
            have_dx = false;
            continue;
        }
        SvbConvEpilogue e;
        memset(&e, 0, sizeof(e));
        e.force_cfg = L.cfg_rs_bwd;
        WN_TRY(svb_conv1d_transposed_bf16x3(drs, L.rs_b_hi, L.rs_b_lo, b->dacts, s->B, L.rs_cout, s->C, 1, s->T, s->T, 1, 1, 0, 1, &e,
                                            stream));
        float* dxin = b->dxin + (size_t)i * 2 * nC;
        WN_TRY(svb_wn_gate_bwd(xin, s->G, b->dacts, dxin, b->dG, nullptr, s->B, s->C, s->T, (s->G || b->dG) ? s->g_channels : 0,
                               i * 2 * s->C, stream));
        if (need_in_w) {
            if (side != main) {
                if (hipEventRecord(events[2 * i + 1], main) != hipSuccess) return SVB_ERR_LAUNCH;
                if (hipStreamWaitEvent(side, events[2 * i + 1], 0) != hipSuccess) return SVB_ERR_LAUNCH;
            }
            WN_TRY(wn_wgrad(s, b, pend, dxin, x, 2 * s->C, s->C, s->k, pad, dil, L.in_g ? L.in_v : nullptr, L.in_g, L.d_in_v, L.d_in_g,
                            L.d_in_b, side));
        }
        if (need_dx) {
            memset(&e, 0, sizeof(e));
            e.residual = dxm;
            e.force_cfg = L.cfg_in_bwd;
            WN_TRY(svb_conv1d_transposed_bf16x3(dxin, L.in_b_hi, L.in_b_lo, b->dx, s->B, 2 * s->C, s->C, 1, s->T, s->T, s->k, 1, pad, dil,
                                                &e, stream));
            have_dx = true;
        } else {
            have_dx = false;
        }
    }
    WN_TRY(wn_flush(pend, side));
    return SVB_OK;
}
