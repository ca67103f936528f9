// Internal kernel-argument structs for conv1d.hip (passed by value to the kernels).
#pragma once
#include "../../include/svb_hip.h"

#define SVB_MAX_TAPS 64
#define SVB_MAX_PHASE 16
#define SVB_WGRAD_BS_TOTAL 10240

// Tap tables: a transposed conv / data-gradient is `n_phase` = stride independent stride-1 convs, one per
// output phase; an ordinary conv is a single phase.  Tap t of phase p reads input position
// q_local*sx + tap_off[t] and uses packed weight slab tap_w[t]; the phase writes output position
// q_local*out_stride + phase_out_base[p] for q_local in [0, phase_nq[p]).
struct SvbConvPlan {
    int n_phase;
    int phase_start[SVB_MAX_PHASE + 1];
    int phase_nq[SVB_MAX_PHASE];
    int phase_out_base[SVB_MAX_PHASE];
    int phase_min_off[SVB_MAX_PHASE];
    int phase_span_off[SVB_MAX_PHASE];
    int tap_off[SVB_MAX_TAPS];
    int tap_w[SVB_MAX_TAPS];
};

struct SvbConvArgs {
    const float* x;
    const float* wp;
    const float* bias;
    float* y;
    const float* in_gate;
    const float* out_gate;
    const float* mask;
    const float* residual;
    float in_slope, out_slope, out_gate_slope;
    int out_act;
    int B, Cin, Cout, G, Cin_g, Cout_g, Tin, Tout;
    int sx, out_stride;
    int w_tap_stride, w_ld, w_goff_k, w_goff_m;
    int xrow, kc, ph_len;
    int ws_floats, xs_floats;   // dynamic LDS carve (weight tile, x tile)
    int tg, fast_x, w_vec;   // taps per weight stage; register-staged x path; 16-byte weight loads allowed
    int force_cfg;
};

struct SvbWgradArgs {
    const float* a;
    const float* b;
    float* part;
    const float* a_gate;
    const float* b_gate;
    float a_slope, b_slope;
    int B, CA, CB, G, CA_g, CB_g, TA, TB;
    int k, sx, off0, dil;
    int qc, n_tg, a_tiles, b_tiles, chunks_per_b, total_chunks, nsplit, brow;
};
