// conv1d_tw.hip -- the "tile-walking" member of the bf16x3 implicit-GEMM conv family (round 4).
//
// Same arithmetic as conv1d_bf16.hip (operands split v = hi + lo into bf16, products hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16, fp32 accumulate), a different machine mapping for the stride-1, ungrouped convs and data
// gradients that carry most of the train step's FLOPs (reference: F.conv1d / F.conv_transpose1d of
// modules/fastspeech/fs2_vae.py:73,83, modules/commons/common_layers.py:739-773, modules/voice_conversion/vae_models.py:86-127):
//
//   * ONE 512-thread workgroup (8 waves, 2 per SIMD) per CU walks several output tiles (persistent, static round-robin over
//     an XCD-aware id): the x / weight tiles of the NEXT tile's first K phase are requested before the epilogue of the
//     current tile, so only the first tile of a workgroup pays a prologue and the epilogue's stores drain behind the next
//     tile's MFMAs.
//   * weights go through LDS: a K phase's (tap, 16-channel chunk) slabs are copied global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, no VGPRs, no ds_write) ONCE per workgroup and shared by the four waves that own the same
//     rows -- the direct-A tiles of conv1d_bf16.hip re-read them per wave through the texture path (80 KB per phase and CU
//     against 40 here, and 4.5 MFMAs per weight VMEM instruction against 12..18).
//   * a wave owns (32*AF) x (32*BF) outputs (AF x BF accumulators): per slab 2AF + 2BF 16-byte LDS reads feed 3*AF*BF MFMAs.
//   * the column space is FLATTENED over the batch: column n <-> (clip n / Tp, position n % Tp) with Tp = Tout + (tap span),
//     so a tile's columns run across clip boundaries (the gap columns between two clips see exactly the conv's zero
//     padding and are not stored) -- tiles do not have to divide T, whatever the sequence length.
//   * LDS images are planar: plane = (hi|lo, 8-channel half kb), one 16-byte element per (row | position); every
//     ds_read_b128 / ds_write_b128 of a 32-lane half touches 32 consecutive elements: bank-conflict free for any tap shift.
//
// Domain: sx == 1, out_stride == 1, one output phase, groups == 1, Cin % 16 == 0, <= TW_MAX_TAPS taps.  Everything else
// stays on conv1d_bf16.hip (svb_tw_launch returns SVB_ERR_UNSUPPORTED).
#include "svb_common.h"
#include "svb_q.h"
#include "conv1d_q.h"
#include <svb_glds.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define TW_MAX_TAPS 8
#define TW_SMAX 8          /* slabs (tap x chunk) per K phase */
#define TW_XU 5            /* x staging units (8 channels of one position) per producer thread and phase */
#define TW_LDS_BUDGET (158 * 1024)

struct SvbTwArgs {
    const float* x;
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
    int B, Cin, Cout, Tin, Tout;
    int Tp, ncols;          // clip pitch of the flattened column space, B * Tp
    int min_off, span;      // smallest tap offset; x-tile rows per chunk (BN + tap span)
    int ntap, kch, nphase, S;   // taps, chunks per K phase, K phases per tile, slabs per phase (kch * ntap)
    int m_tiles, ntiles;
    int w_rows, w_chunk16;  // rows per weight slab; 16-byte units between the slabs of consecutive chunks
    int xunits;             // ceil(kch * span / 128): x staging units per producer thread and phase
    int drain_per_pair;     // deferred-store groups (4 stores each, 16 per tile) a consumer issues behind each pair of slabs
    unsigned slx[TW_SMAX / 2];  // slab s -> row offset inside an x plane (chunk * span + tap_off - min_off), 16 bits each
    int w_tap16;                // 16-byte units between the slabs of consecutive taps (weight slab of tap t = t)
#ifdef SVB_INSTRUMENT
    unsigned long long* dbg;    // stage stamps [workgroup < 4][wave][phase < 32][8] (tools/tw_stage_timing.py)
    int ablate;                 // (unused: the timing-only ablations are compile-time, SVB_TW_ABLATE: 1 no x loads, 2 no weight DMA,
                                // 4 no MFMAs, 8 no stores, 16 no x split / LDS store, 32 no fragment reads)
#endif
};

#ifdef SVB_INSTRUMENT
static unsigned long long* g_tw_dbg = nullptr;
static int g_tw_ablate = 0, g_tw_drain = 0;
extern "C" void svb_debug_set_tw(void* buf, int ablate, int drain_per_pair) {
    g_tw_dbg = (unsigned long long*)buf; g_tw_ablate = ablate; g_tw_drain = drain_per_pair;
}
#ifndef SVB_TW_ABLATE
#define SVB_TW_ABLATE 0
#endif
#define TW_ABL(bit) (((SVB_TW_ABLATE) & (bit)) != 0)      /* compile-time: `make abl` builds one library per mask */
#define TW_STAMP(slot)                                                                                          \
    if (arg.dbg && vid < 4 && lane == 0 && dbg_ph < 32)                                                          \
        arg.dbg[(((size_t)vid * 8 + wave) * 32 + dbg_ph) * 8 + (slot)] = __builtin_readcyclecounter();
#define TW_NEXT_PHASE ++dbg_ph;
#else
#define TW_ABL(bit) false
#define TW_STAMP(slot)
#define TW_NEXT_PHASE
#endif

// Roles: waves 0..3 (one per SIMD) are CONSUMERS -- each owns a 64 x 64 block of the workgroup's (64 CM) x (64 CN) tile as four
// 32x32 accumulators and does nothing but LDS fragment reads and MFMAs; waves 4..7 are PRODUCERS -- they stage the next K
// phase (x: buffer loads -> split -> ds_write, weights: LDS-DMA) into the other LDS buffer.  One barrier per K phase.
// (Round 4 measured the symmetric form first -- every wave staging AND multiplying: a wave's serial non-MFMA work per phase
//  was ~4k cycles against 1.9k cycles of MFMA issue, phases took 8.6k cycles where the matrix pipe needs 3.8k --
//  profiles/r04_tw_symmetric_stages.log.)
// The epilogue of the plain form (bias + ReLU / LeakyReLU / none) is DEFERRED: at the end of a tile the consumer applies bias +
// activation into a second register set and drains it with a few stores per slab of the NEXT tile, behind its MFMAs -- the
// 25-32 MB write burst of 256 workgroups reaching their epilogue together (20k cycles per tile, measured) becomes a steady
// ~3 B/clk/CU stream.  Epilogues with gate / residual / mask / tanh run at the end of their tile.
template <int CM, int CN, int GATE>
__global__ __launch_bounds__(512, 2) void svb_conv1d_tw_kernel(SvbTwArgs arg) {
    constexpr int AF = 2, BF = 2;
    constexpr int BM = 64 * CM, BN = 64 * CN;
    constexpr int NR = 2 * AF + 2 * BF, NM = 3 * AF * BF, NAB = AF * BF;
    constexpr int NGRP = NAB * 4;                     // deferred-store groups: 4 consecutive accumulator rows each
    static_assert(CM * CN == 4, "four consumer waves");
    HIP_DYNAMIC_SHARED(uint4, smem)
    // kernel arguments as plain locals (a by-value struct that lambdas capture by reference can end up in scratch)
    const int p_B = arg.B, p_Cin = arg.Cin, p_Cout = arg.Cout, p_Tin = arg.Tin, p_Tout = arg.Tout;
    const int p_Tp = arg.Tp, p_ncols = arg.ncols, p_min_off = arg.min_off, p_span = arg.span;
    const int p_ntap = arg.ntap, p_kch = arg.kch, p_nphase = arg.nphase, p_S = arg.S;
    const int p_m_tiles = arg.m_tiles, p_ntiles = arg.ntiles, p_w_rows = arg.w_rows;
    const unsigned p_w_chunk16 = (unsigned)arg.w_chunk16, p_w_tap16 = (unsigned)arg.w_tap16;
    const int p_xunits = arg.xunits;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = lane >> 5, l31 = lane & 31;
    const int XR = p_kch * p_span;                    // rows per x plane
    const int WU = p_S * 4 * BM;                      // 16-byte units per weight buffer
    const int XU16 = 4 * XR;                          // ... per x buffer
    const int xbuf0 = 2 * WU;                         // LDS (16-byte units): W0 | W1 | X0 | X1

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int qd = nwg >> 3, rd = nwg & 7, xcd = orig & 7;
    const int vid = (xcd < rd ? xcd * (qd + 1) : rd * (qd + 1) + (xcd - rd) * qd) + (orig >> 3);
    if (vid >= p_ntiles) return;
#ifdef SVB_INSTRUMENT
    int dbg_ph = 0;
#endif

    if (wave >= 4) {
        // =============================================== PRODUCERS ===============================================================
        const int pw = wave - 4;
        const int ptid = tid - 256;
        const int xh = pw >> 1;                           // waves 4,5: channel half 0 of a chunk; waves 6,7: half 1
        const int x_row0 = ptid & 127;
        // weight DMA roles: unit v = pw + 4 j of a slab's 4*CM (plane, 64-row block) units
        int w_pl[CM], w_rb[CM];
        const uint4* w_arr[CM];            // hi or lo array, advanced to the plane's 8-channel half
        {
            const uint4* const w_hi16 = reinterpret_cast<const uint4*>(arg.wq_hi);
            const uint4* const w_lo16 = reinterpret_cast<const uint4*>(arg.wq_lo);
#pragma unroll
            for (int j = 0; j < CM; ++j) {
                const int v = pw + 4 * j;
                w_pl[j] = v / CM;
                w_rb[j] = v - w_pl[j] * CM;
                w_arr[j] = ((w_pl[j] & 2) ? w_lo16 : w_hi16) + (w_pl[j] & 1);
            }
        }
        // x staging: unit u of a thread = 8 channels (half xh of a 16-channel chunk) of plane row u * 128 + x_row0; loads go
        // through a buffer descriptor: one per-lane 32-bit offset per unit and tile, the channel part is a scalar offset
        const __amdgpu_buffer_rsrc_t x_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(arg.x), 0, 4u * (unsigned)(p_B * p_Cin * p_Tin), 0x00020000);
        const __amdgpu_buffer_rsrc_t g_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE ? arg.in_gate : arg.x), 0, 4u * (unsigned)(p_B * p_Cin * p_Tin), 0x00020000);
        unsigned x_off[TW_XU];
        unsigned x_okmask = 0;                  // bit u: the unit's position is inside a clip (else the row is zero)
        unsigned w_row16[CM];
        auto setup_tile = [&](int tile) {
            const int mt = tile % p_m_tiles, nt = tile / p_m_tiles;
            const int m_base = mt * BM, n0 = nt * BN;
            x_okmask = 0;
#pragma unroll
            for (int u = 0; u < TW_XU; ++u) {
                const int idx = u * 128 + x_row0;
                const int c = idx / p_span, i = idx - c * p_span;
                const int f = n0 + i;
                const int bf = f / p_Tp, uf = f - bf * p_Tp;
                const int pos = uf + p_min_off;
                const bool ok = idx < XR && bf < p_B && pos >= 0 && pos < p_Tin;
                x_okmask |= (ok ? 1u : 0u) << u;
                x_off[u] = ok ? 4u * (unsigned)((bf * p_Cin + c * 16 + xh * 8) * p_Tin + pos) : 0u;
            }
#pragma unroll
            for (int j = 0; j < CM; ++j)
                w_row16[j] = 2u * (unsigned)min(m_base + w_rb[j] * 64 + lane, p_w_rows - 1);
        };
        float xr[TW_XU][8];
        float gr[GATE == 1 ? TW_XU : 1][8];      // (GATE 2: the gate tensor is x itself -- conv(leaky_relu(x)) -- no second load)
        // all loads of a phase are requested at once; a unit outside the tile (or its clip) reads element 0 of the channel row
        auto issue_x = [&](int ph) {
            if (TW_ABL(1)) return;
            const unsigned ch0 = (unsigned)(ph * p_kch * 16);
#pragma unroll
            for (int u = 0; u < TW_XU; ++u)
                if (u < p_xunits) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        xr[u][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, x_off[u], 4u * (ch0 + e) * (unsigned)p_Tin, 0));
                    if (GATE == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            gr[u][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, x_off[u], 4u * (ch0 + e) * (unsigned)p_Tin, 0));
                    }
                }
        };
        auto finish_x = [&](int xb16) {           // split + store into the x buffer that starts at 16-byte unit xb16
            if (TW_ABL(16)) return;
            const float in_slope = arg.in_slope;
#pragma unroll
            for (int u = 0; u < TW_XU; ++u)
                if (u < p_xunits && u * 128 + x_row0 < XR) {
                    if (GATE == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) xr[u][e] *= svb_gate(gr[u][e], in_slope);
                    } else if (GATE == 2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) xr[u][e] *= svb_gate(xr[u][e], in_slope);
                    }
                    unsigned h[4], l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) svbq_split2(xr[u][2 * e], xr[u][2 * e + 1], h[e], l[e]);
                    uint4 hi = make_uint4(h[0], h[1], h[2], h[3]), lo = make_uint4(l[0], l[1], l[2], l[3]);
                    if (!((x_okmask >> u) & 1u)) { hi = make_uint4(0u, 0u, 0u, 0u); lo = hi; }
                    smem[xb16 + xh * XR + u * 128 + x_row0] = hi;
                    smem[xb16 + (2 + xh) * XR + u * 128 + x_row0] = lo;
                }
        };
        // weight slabs of K phase ph -> the weight buffer that starts at 16-byte unit wb16 (LDS-DMA: lane L of an instruction
        // fetches row (64-row block) + L of one plane, the 64 x 16 bytes land consecutively)
        auto issue_w = [&](int ph, int wb16) {
            if (TW_ABL(2)) return;
            unsigned cb16 = (unsigned)(ph * p_kch) * p_w_chunk16;
            int dst = wb16;
            for (int c = 0; c < p_kch; ++c, cb16 += p_w_chunk16) {
                unsigned sb16 = cb16;
                for (int t = 0; t < p_ntap; ++t, sb16 += p_w_tap16, dst += 4 * BM) {
#pragma unroll
                    for (int j = 0; j < CM; ++j)
                        svb_glds16(w_arr[j] + (size_t)sb16 + w_row16[j], smem, 16u * (unsigned)(dst + w_pl[j] * BM + w_rb[j] * 64));
                }
            }
        };

        int tile = vid, ph = 0, buf = 0;
        setup_tile(tile);
        issue_w(0, 0);
        issue_x(0);
        finish_x(xbuf0);
        __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) lgkmcnt(0): DMA landed, LDS stores done
        __builtin_amdgcn_s_barrier();
        while (true) {
            const bool last_ph = ph + 1 == p_nphase;
            const int ntile = last_ph ? tile + nwg : tile;
            const bool has_next = ntile < p_ntiles;
            const int nph = last_ph ? 0 : ph + 1;
            TW_STAMP(0)
            if (has_next) {
                if (last_ph) setup_tile(ntile);
                issue_w(nph, (buf ^ 1) * WU);
                issue_x(nph);
                TW_STAMP(1)
                finish_x(xbuf0 + (buf ^ 1) * XU16);
                TW_STAMP(2)
            }
            __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) lgkmcnt(0)
            TW_STAMP(3)
            __builtin_amdgcn_s_barrier();
            TW_STAMP(4)
            TW_NEXT_PHASE
            if (!has_next) break;
            buf ^= 1;
            ph = nph;
            tile = ntile;
        }
        return;
    }

    // ================================================= CONSUMERS =================================================================
    const int cm = wave % CM, cn = wave / CM;
    const int a_lane = kb * BM + cm * 32 * AF + l31;           // A fragment: plane kb, this wave's rows
    const int b_lane = kb * XR + cn * 32 * BF + l31;           // B fragment: plane kb, this wave's columns
    f32x16 acc[AF][BF], pend[AF][BF];
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int n = 0; n < BF; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][n][r] = 0.f; pend[i][n][r] = 0.f; }
    // (opaque to the optimiser: with `pend` a known constant the whole tile loop is peeled once -- twice the code -- for a first
    // tile that never stores it)
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int n = 0; n < BF; ++n) svb_opaque16(pend[i][n]);

    const unsigned y_bytes = 4u * (unsigned)(p_B * p_Cout * p_Tout);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(arg.y, 0, y_bytes, 0x00020000);
    const float* const e_bias = arg.bias;
    const int e_act = arg.out_act;
    const float e_slope = arg.out_slope;
    const bool e_plain = !arg.out_gate && !arg.residual && !arg.mask && e_act != SVB_ACT_TANH;
    const float e_neg = e_act == SVB_ACT_RELU ? 0.f : (e_act == SVB_ACT_LRELU ? e_slope : 1.f);      // v > 0 ? v : v * neg

    // per-lane store offsets of a tile: voff[i][n] = 4 ((first row of accumulator i) * Tout + clip * Cout * Tout + position), or
    // an offset beyond the buffer for columns outside the tensor (a buffer store there is dropped by the bounds check)
    auto tile_offsets = [&](int m_base, int n0, unsigned (&voff)[AF][BF], int& ml0) {
        ml0 = m_base + cm * 32 * AF + 4 * kb;
#pragma unroll
        for (int n = 0; n < BF; ++n) {
            const int f = n0 + cn * 32 * BF + 32 * n + l31;
            const int bf = f / p_Tp, uf = f - bf * p_Tp;
            const bool ok = f < p_ncols && uf < p_Tout;
#pragma unroll
            for (int i = 0; i < AF; ++i)
                voff[i][n] = ok ? 4u * (unsigned)((ml0 + 32 * i) * p_Tout + bf * p_Cout * p_Tout + uf) : 0x80000000u;
        }
    };
    // deferred stores: straight-line code, validity folded into the offset (a buffer store at an offset >= 2^31 is dropped by the
    // bounds check: columns outside the tensor, rows beyond Cout in a ragged last row tile)
    unsigned p_voff[AF][BF];
    int p_lim = 0;                          // rows of the pending tile inside the tensor, counted from this lane's first row
    int p_next = NGRP;                      // next group to store (NGRP: nothing pending)
    const unsigned t4 = 4u * (unsigned)p_Tout;
    auto store_group = [&](int g) {         // rows 8 (g & 3) + 4 kb + {0..3} of accumulator (g >> 2)
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                if (g == ab * 4 + r4) {
                    const int i = ab / BF, n = ab - i * BF;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int dr = 8 * r4 + e;
                        const unsigned off = (32 * i + dr < p_lim) ? p_voff[i][n] + (unsigned)dr * t4 : 0x80000000u;
                        const float v = pend[i][n][4 * r4 + e];      // (a scalar first: bit_cast of a vector element reads element 0)
                        if (!TW_ABL(8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, off, 0, 0);
                    }
                }
    };
    // end of a tile, plain form: pend = act(acc + bias), offsets kept, acc = 0.  (The previous tile's stores have all been
    // issued: drain_per_pair covers the 16 groups within one tile's slabs.)
    auto retire_tile = [&](int m_base, int n0) {
        int ml0;
        tile_offsets(m_base, n0, p_voff, ml0);
        p_lim = p_Cout - ml0;
#pragma unroll
        for (int i = 0; i < AF; ++i) {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = e_bias ? e_bias[min(ml0 + 32 * i + (r & 3) + 8 * (r >> 2), p_Cout - 1)] : 0.f;
#pragma unroll
            for (int n = 0; n < BF; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][n][r] + bv[r];
                    pend[i][n][r] = v > 0.f ? v : v * e_neg;
                    acc[i][n][r] = 0.f;
                }
        }
        p_next = 0;
    };
    // end of a tile, general form (gate / residual / mask / tanh): stored at once
    auto epilogue_now = [&](int m_base, int n0) {
        const float* const e_mask = arg.mask;
        const bool has_gate = arg.out_gate != nullptr, has_res = arg.residual != nullptr;
        const __amdgpu_buffer_rsrc_t og_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_gate ? arg.out_gate : arg.y), 0, y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_res ? arg.residual : arg.y), 0, y_bytes, 0x00020000);
        const float gslope = arg.out_gate_slope;
        unsigned voff[AF][BF];
        int ml0;
        tile_offsets(m_base, n0, voff, ml0);
        const bool full_m = m_base + BM <= p_Cout;
        float mk[BF];
#pragma unroll
        for (int n = 0; n < BF; ++n) {
            const int f = n0 + cn * 32 * BF + 32 * n + l31;
            const int bf = f / p_Tp, uf = f - bf * p_Tp;
            mk[n] = (e_mask && f < p_ncols && uf < p_Tout) ? e_mask[bf * p_Tout + uf] : 1.f;
        }
#pragma unroll
        for (int i = 0; i < AF; ++i) {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = e_bias ? e_bias[min(ml0 + 32 * i + (r & 3) + 8 * (r >> 2), p_Cout - 1)] : 0.f;
#pragma unroll
            for (int n = 0; n < BF; ++n)
                if (voff[i][n] < 0x80000000u) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        const unsigned soff = 4u * (unsigned)(dr * p_Tout);
                        if (full_m || ml0 + 32 * i + dr < p_Cout) {
                            float v = svb_apply_act(acc[i][n][r] + bv[r], e_act, e_slope);
                            if (has_gate) v *= svb_gate(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(og_rsrc, voff[i][n], soff, 0)), gslope);
                            if (has_res) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rsrc, voff[i][n], soff, 0));
                            v *= mk[n];
                            if (!TW_ABL(8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, voff[i][n], soff, 0);
                        }
                        acc[i][n][r] = 0.f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
                }
        }
    };

    // ---- one K phase of MFMAs.  Fragments are double-buffered in registers by slab parity; every LDS read of slab s+1 sits,
    // placed by hand and pinned, behind one MFMA of slab s.  LDS addresses: one per-lane byte address per operand plus
    // wave-uniform offsets.
    uint4 fa[2][2 * AF], fb[2][2 * BF];
    const unsigned x_lo16 = 32u * (unsigned)XR;                  // byte distance hi plane -> lo plane of the x image
    const char* const lds = reinterpret_cast<const char*>(smem);
    const unsigned a_lane16 = 16u * (unsigned)a_lane, b_lane16 = 16u * (unsigned)b_lane;
    auto read_frag = [&](int par, int r, unsigned wofs, unsigned xofs) {   // r: 0 .. NR-1 = A hi[AF], A lo[AF], B hi[BF], B lo[BF]
        if (r < 2 * AF) {
            const int arr = r / AF, i = r - arr * AF;
            fa[par][r] = *reinterpret_cast<const uint4*>(lds + (a_lane16 + wofs) + 16u * (unsigned)(arr * 2 * BM + i * 32));
        } else {
            const int q = r - 2 * AF, arr = q / BF, n = q - arr * BF;
            fb[par][q] = *reinterpret_cast<const uint4*>(lds + (b_lane16 + xofs + (arr ? x_lo16 : 0u)) + 16u * (unsigned)(n * 32));
        }
    };
    auto mfma_step = [&](int par, int j) {
        const int prod = j / NAB, idx = j - prod * NAB;
        const int i = idx / BF, n = idx - i * BF;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&fa[par][i]), al = *reinterpret_cast<const bf16x8*>(&fa[par][AF + i]);
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&fb[par][n]), bl = *reinterpret_cast<const bf16x8*>(&fb[par][BF + n]);
        if (prod == 0) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][n], 0, 0, 0);
        else if (prod == 1) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][n], 0, 0, 0);
        else acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][n], 0, 0, 0);
    };
    // slab -> row offset inside an x plane (chunk * span + tap row), four 16-bit fields per word (host-packed)
    const unsigned long long slx_lo = (unsigned long long)arg.slx[0] | ((unsigned long long)arg.slx[1] << 32);
    const unsigned long long slx_hi = (unsigned long long)arg.slx[2] | ((unsigned long long)arg.slx[3] << 32);
    auto slab_row = [&](int s) -> unsigned {       // (shifts, not a select chain: that becomes a lookup table in scratch)
        const unsigned long long w = (s & 4) ? slx_hi : slx_lo;
        return (unsigned)(w >> (16 * (s & 3))) & 0xFFFFu;
    };
    // the MFMAs of the slab held in fragment set `par`, the fragments of slab `sn` read into the other set meanwhile (past
    // the phase's last slab `sn` is clamped: a harmless re-read instead of a second copy of the body)
    auto slab_body = [&](int par, unsigned wb, unsigned xb, int sn) {
        const unsigned wofs = wb + (unsigned)sn * (64u * BM), xofs = xb + 16u * slab_row(sn);
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            if (!TW_ABL(4)) mfma_step(par, j);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (r >= j * NR / NM && r < (j + 1) * NR / NM && !TW_ABL(32)) read_frag(par ^ 1, r, wofs, xofs);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int s_last = p_S - 1;
    const int drain = arg.drain_per_pair;           // deferred-store groups issued behind each pair of slabs
    int buf = 0;
    __builtin_amdgcn_s_barrier();                   // the first K phase is staged
    for (int tile = vid; tile < p_ntiles; tile += nwg) {
        const int mt = tile % p_m_tiles, nt = tile / p_m_tiles;
        for (int ph = 0; ph < p_nphase; ++ph) {
            const unsigned wb = 16u * (unsigned)(buf * WU), xb = 16u * (unsigned)(xbuf0 + buf * XU16);
            TW_STAMP(0)
#pragma unroll
            for (int r = 0; r < NR; ++r) read_frag(0, r, wb, xb + 16u * slab_row(0));
            for (int s = 0; s < p_S; s += 2) {
                slab_body(0, wb, xb, min(s + 1, s_last));
                if (s + 1 < p_S) slab_body(1, wb, xb, min(s + 2, s_last));
                for (int d = 0; d < drain && p_next < NGRP; ++d, ++p_next) store_group(p_next);
            }
            TW_STAMP(1)
            __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0): the clamped re-reads of the last slab are done with this buffer
            __builtin_amdgcn_s_barrier();
            TW_STAMP(2)
            TW_NEXT_PHASE
            buf ^= 1;
        }
        if (e_plain) retire_tile(mt * BM, nt * BN);
        else epilogue_now(mt * BM, nt * BN);
    }
    for (; p_next < NGRP; ++p_next) store_group(p_next);
}

// ==================================================================================================================
struct TwCfg { int CM, CN; };
static const TwCfg kTwCfgs[SVB_TW_NVARIANTS] = {{2, 2}, {1, 4}, {4, 1}};      // 128x128, 64x256, 256x64

// (the dynamic-LDS attribute belongs to the (function, device) pair and the CU count to the device: both are kept per device, so a
//  process that drives several GPUs gets them on each.  Racing first calls of two host threads set the same value twice.)
#define SVB_TW_MAX_DEVICES 16
static int tw_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SVB_TW_MAX_DEVICES) return -1;
    return dev;
}

template <int CM, int CN, int GATE>
static void tw_launch_kernel(const SvbTwArgs& a, int grid, size_t lds, hipStream_t stream, int dev) {
    static bool attr_set[SVB_TW_MAX_DEVICES] = {};
    if (dev < 0 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&svb_conv1d_tw_kernel<CM, CN, GATE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((svb_conv1d_tw_kernel<CM, CN, GATE>), dim3(grid), dim3(512), lds, stream, a);
}

template <int CM, int CN>
static void tw_launch_gate(const SvbTwArgs& a, int grid, size_t lds, hipStream_t stream, int dev) {
    if (a.in_gate && a.in_gate == a.x) tw_launch_kernel<CM, CN, 2>(a, grid, lds, stream, dev);
    else if (a.in_gate) tw_launch_kernel<CM, CN, 1>(a, grid, lds, stream, dev);
    else tw_launch_kernel<CM, CN, 0>(a, grid, lds, stream, dev);
}

static int g_tw_cus[SVB_TW_MAX_DEVICES] = {};

int svb_tw_launch(const SvbConvQArgs& q, const SvbConvPlan& p, int variant, hipStream_t stream) {
    if (variant < 0 || variant >= SVB_TW_NVARIANTS) return SVB_ERR_UNSUPPORTED;
    if (p.n_phase != 1 || q.sx != 1 || q.out_stride != 1 || q.G != 1 || q.Cin % 16 || q.xq) return SVB_ERR_UNSUPPORTED;
    const int ntap = p.phase_start[1] - p.phase_start[0];
    if (ntap < 1 || ntap > TW_MAX_TAPS || p.phase_out_base[0] != 0 || p.phase_nq[0] != q.Tout) return SVB_ERR_UNSUPPORTED;
    const int BM = 64 * kTwCfgs[variant].CM, BN = 64 * kTwCfgs[variant].CN;
    if (q.Cout < 32) return SVB_ERR_UNSUPPORTED;
    // 32-bit byte offsets inside the kernel; an offset >= 2^31 marks a column outside the tensor
    if ((long)q.B * q.Cin * q.Tin >= (1L << 30) || (long)q.B * q.Cout * q.Tout >= (1L << 29)) return SVB_ERR_UNSUPPORTED;

    SvbTwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = q.x; a.wq_hi = q.wq_hi; a.wq_lo = q.wq_lo; a.bias = q.bias; a.y = q.y;
    a.in_gate = q.in_gate; a.out_gate = q.out_gate; a.mask = q.mask; a.residual = q.residual;
    a.in_slope = q.in_slope; a.out_slope = q.out_slope; a.out_gate_slope = q.out_gate_slope; a.out_act = q.out_act;
    a.B = q.B; a.Cin = q.Cin; a.Cout = q.Cout; a.Tin = q.Tin; a.Tout = q.Tout;
    const int span_off = p.phase_span_off[0];
    a.min_off = p.phase_min_off[0];
    a.Tp = q.Tout + span_off;
    if ((long)q.B * a.Tp >= (1L << 30)) return SVB_ERR_UNSUPPORTED;
    a.ncols = q.B * a.Tp;
    a.span = BN + span_off;
    a.ntap = ntap;
    const int kchunks = q.Cin / 16;
    a.w_rows = q.w_slab_rows;
    a.w_chunk16 = q.w_slab_rows * 2;
    if ((long)q.w_tap_slabs * q.w_slab_rows * 2 * (SVB_MAX_TAPS + 1) >= (1L << 31)) return SVB_ERR_UNSUPPORTED;
    for (int t = 0; t < ntap; ++t)
        if (p.tap_w[p.phase_start[0] + t] != t) return SVB_ERR_UNSUPPORTED;
    a.w_tap16 = q.w_tap_slabs * q.w_slab_rows * 2;
    // chunks per phase: as many as divide the K extent, fit the slab / staging-unit caps and the LDS budget
    int kch = 0;
    for (int k = TW_SMAX / ntap; k >= 1; --k) {
        if (kchunks % k) continue;
        if ((k * a.span + 127) / 128 > TW_XU) continue;
        const size_t lds = (size_t)2 * ((size_t)k * ntap * 4 * BM + (size_t)4 * k * a.span) * 16;
        if (lds > TW_LDS_BUDGET) continue;
        kch = k;
        break;
    }
    if (!kch) return SVB_ERR_UNSUPPORTED;
    a.kch = kch;
    a.S = kch * ntap;
    a.nphase = kchunks / kch;
    a.xunits = (kch * a.span + 127) / 128;
    // the 16 deferred-store groups of a tile drain behind the slab pairs of the next one: all of them within one tile
    const int pairs = a.nphase * ((a.S + 1) / 2);
    a.drain_per_pair = (16 + pairs - 1) / pairs;
#ifdef SVB_INSTRUMENT
    a.dbg = g_tw_dbg;
    a.ablate = g_tw_ablate;
    if (g_tw_drain > a.drain_per_pair) a.drain_per_pair = g_tw_drain;
#endif
    if (kch * a.span + span_off >= 65536) return SVB_ERR_UNSUPPORTED;
    for (int s = 0; s < a.S; ++s) {
        const unsigned row = (unsigned)((s / ntap) * a.span + p.tap_off[p.phase_start[0] + s % ntap] - a.min_off);
        a.slx[s >> 1] |= row << (16 * (s & 1));
    }
    a.m_tiles = svb_cdiv(q.Cout, BM);
    const long n_tiles = ((long)a.ncols + BN - 1) / BN;
    if (a.m_tiles * n_tiles >= (1L << 30)) return SVB_ERR_UNSUPPORTED;
    a.ntiles = (int)(a.m_tiles * n_tiles);
    size_t lds = (size_t)2 * ((size_t)a.S * 4 * BM + (size_t)4 * kch * a.span) * 16;
    if (lds < 82 * 1024) lds = 82 * 1024;          // one workgroup per CU whatever the tile needs (grid = CUs)
    const int dev = tw_device();
    int cus = dev >= 0 ? g_tw_cus[dev] : 0;
    if (!cus) {
        if (dev < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (dev >= 0) g_tw_cus[dev] = cus;
    }
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    switch (variant) {
        case 0: tw_launch_gate<2, 2>(a, grid, lds, stream, dev); break;
        case 1: tw_launch_gate<1, 4>(a, grid, lds, stream, dev); break;
        default: tw_launch_gate<4, 1>(a, grid, lds, stream, dev); break;
    }
    // a launch the runtime refuses (e.g. the LDS attribute could not be set on this device) is reported as "outside this kernel's
    // domain": the dispatcher then runs the heuristic tile of conv1d_bf16.hip instead of failing the conv
    if (hipGetLastError() != hipSuccess) return SVB_ERR_UNSUPPORTED;
    return SVB_OK;
}
