// Kernel-argument struct of the bf16x3 implicit-GEMM conv family (conv1d_bf16.hip, conv1d_tw.hip).
#pragma once
#include "conv1d.h"

struct SvbConvQArgs {
    const float* x;
    const unsigned short* xq;     // optional Q image of x (svb_q.h): staged with plain 16-byte copies
    const unsigned short* wq_hi;
    const unsigned short* wq_lo;
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
    int w_tap_slabs, w_g_slabs, w_slab_rows, w_goff_m, kchunks;
    int tg, kch, xrows, fast_x, xit;   // xit: 128-position groups per chunk on the register-staged path (1..3)
    int fast;                          // direct-A tiles, every K phase has exactly SLB slabs: straight-line pipelined loop
    int w_floats16, x_floats16;   // LDS carve sizes in 16-byte units (one of hi/lo each)
    int force_cfg;
#ifdef SVB_INSTRUMENT
    int prio;                          // experiment switch (env SVB_CONV_PRIO): wave priority of the MFMA stage (1) / of the staging stage (2)
    unsigned long long* dbg;      // optional per-phase cycle stamps (svb_debug_set_timing_buffer; tools/stage_timing.py)
    int dbg_block0;               // first launch-order workgroup id that is stamped (env SVB_DBG_BLOCK0)
#endif
};

// conv1d_tw.hip: the 8-wave tile-walking kernel.  variant = 0 .. SVB_TW_NVARIANTS-1; returns SVB_ERR_UNSUPPORTED when the
// conv is outside its domain (strided / grouped / ragged channel chunks / too many taps) -- the caller then takes a
// tile of conv1d_bf16.hip.
#define SVB_TW_NVARIANTS 3
int svb_tw_launch(const SvbConvQArgs& a, const SvbConvPlan& p, int variant, hipStream_t stream);

// conv1d_pw.hip: the GEMM form (1 .. 16 taps, stride 1, equal length).  variant = 0 .. SVB_PW_NVARIANTS-1 (128x128, 128x256, 64x256,
// 64x128, 96x128, 96x256 output tiles); returns SVB_ERR_UNSUPPORTED outside its domain (strides, groups, Tin != Tout, Cin % 16, Cout % 8).
#define SVB_PW_NVARIANTS 6
int svb_pw_launch(const SvbConvQArgs& a, const SvbConvPlan& p, int variant, hipStream_t stream);

// ---- weight gradients (conv1d_bf16.hip: the tap-group kernel; conv1d_wgrad_pw.hip: the direct-operand 1-tap kernel)
struct SvbWgradQArgs {
    const float* a;
    const float* b;
    float* part;
    float* bias_part;   // optional [nsplit][CA]: per-split sums of the (gated) A rows = bias-gradient partials when A = dy
    const float* a_gate;
    const float* b_gate;
    float a_slope, b_slope;
    int B, CA, CB, G, CA_g, CB_g, TA, TB;
    int k, off0, dil, sx;
    int n_tg, a_tiles, b_tiles, chunks_per_b, total_chunks, nsplit;
    int at, bt;   // 32x32 accumulator tiles per wave along A / B rows (workgroup tile 64*at x 64*bt)
    int gp_ca, gp_cb;   // group packing: G / CA_g / CB_g above describe `gp` real groups merged into one (their channels are
                        // contiguous), so that a 64x64 tile holds gp diagonal blocks of gp_ca x gp_cb REAL per-group channels instead
                        // of one (MSD's grouped k41 convs: 16 x 8 channels per group); only those blocks are stored.  0 = off.
    int pa, pb;   // LDS row pitches in dwords (2 * odd)
    int xcd_map;  // 1: XCD-aware work ids (the product's constant; the instrumentation build can switch it off for an A/B)
    // tap groups.  Stride 1: group i = taps [i*TGW, ...), Bt position of tile index t: q0 + off0 + j0*dil + t.
    // Stride s > 1 (dil 1): taps are grouped by phase r = (j - pad) mod s; within a phase the strided gather
    // q*s + j - pad = (q + o_j)*s + r is a stride-1 walk over the phase-r subsequence of Bt, so a group is a stride-1
    // problem on positions (q0 + o0 + t)*s + r with weight taps j0, j0 + s, j0 + 2s, ...
    short tg_j0[SVB_MAX_TAPS], tg_ntap[SVB_MAX_TAPS], tg_r[SVB_MAX_TAPS], tg_o0[SVB_MAX_TAPS];
};

// conv1d_wgrad_pw.hip: dW = dy . x^T of 1-tap, stride-1, ungrouped convs with both operands loaded straight into MFMA fragments
// (no LDS).  Same partial layout / split assignment as the tap-group kernel; returns SVB_ERR_UNSUPPORTED outside its domain.
int svb_wgrad_pw_launch(const SvbWgradQArgs& a, hipStream_t stream);
// its domain test and wave tile (at x bt accumulator tiles per wave, workgroup tile 64 at x 64 bt) -- also sizes the split count
bool svb_wgrad_pw_plan(int CA, int CB, int groups, int k, int sx, int pad, int dil, int TA, int* at, int* bt);
