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
