// conv1d_pw.hip -- the GEMM member of the bf16x3 conv family (round 6): stride-1, equal-length, ungrouped convs of 1 .. 16 taps as
//
//        Y[Cout x N] = sum_tap W_tap[Cout x Cin] . shift_tap(X)[Cin x N],      N = B * T  (the batch's positions, flattened),
//
// first built for the 1-tap case -- every nn.Linear / Conv1d(k=1) of the path and their data gradients (reference: modules/fastspeech/conformer/layers.py:182-258
// FFN / pointwise convs, modules/commons/espnet_transformer_attn.py:125-186 q/k/v/out projections, modules/fastspeech/fs2_vae.py:66-91
// res_skip and cond layers, modules/voice_conversion/svb_vae.py:152-162 condition projection).  Same arithmetic as conv1d_bf16.hip --
// operands split v = hi + lo into bf16, products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate, K walked in
// ascending 16-channel chunks -- so results are BIT-IDENTICAL to the other tiles of the family; only the machine mapping differs:
//
//   * these convs are streaming problems (K = 12 ... 64 chunks, 64 ... 300 FLOP per HBM byte): what the tap-table kernels pay for on
//     them is machinery -- an x tile staged through VGPRs -> split -> LDS with two barriers per K phase, weight fragments re-read
//     per wave through the texture path, per-clip tiles that waste up to 27 % of a T = 281 row.
//   * here the column space is flattened over the batch (a tile's columns run across clip boundaries: no ragged last tile per clip);
//   * the four waves of a workgroup own DISJOINT column blocks and ALL of the tile's rows: a lane's MFMA B operand (8 channels of
//     one position) is loaded straight global -> VGPR (8 dword loads, coalesced over the 32 positions of the lane group) and split in
//     registers -- the x operand never touches LDS and is fetched exactly once per workgroup;
//   * the weights, which all four waves share, come in by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), one fully
//     coalesced 1-KiB piece per (32-row block, hi|lo); the LDS image is [block][hi|lo][channel half][row] x 16 B, so every A-operand
//     ds_read_b128 touches 64 consecutive 16-byte slots: bank-conflict free, one base VGPR + immediate offsets;
//   * K phases of 2 or 4 chunks, double-buffered weights, ONE barrier per phase; x is prefetched 2 .. 4 chunks ahead in registers
//     (loads pinned in front of each chunk's arithmetic); the next phase's weight pieces are requested right behind the barrier
//     and have a whole phase to land.
//
//   * taps: the K walk runs over (chunk, tap) slabs, tap fastest; a tap only moves the position a lane loads (its column + the
//     tap's offset; outside the clip the buffer bounds check answers 0 = the conv's zero padding).  x is then fetched once per tap
//     (L1 / L2 hits after the first) instead of staged once into LDS -- measured on the MI355X that is still 20-40 % faster than
//     the tap-table tiles on the vocoder's k = 3 / 7 / 11 residual convs and their data gradients (profiles/r06_tile_tuner_taps16.log);
//   * an input gate (x * lrelu'(gate): the data gradients of the critic towers; gate == x: `conv(leaky_relu(x))` of the HifiGAN
//     generator) is applied in registers before the split.
//
// Domain: 1 .. 16 taps in one phase with weight slabs in arithmetic progression, Tin == Tout, groups == 1, sx == out_stride == 1,
// (Cin / 16) * taps a multiple of the tile's phase length (4; 2 for the 128x128 tile), Cin % 16 == 0, Cout % 8 == 0.
// Everything else stays on conv1d_bf16.hip (svb_pw_launch returns SVB_ERR_UNSUPPORTED).
#include "svb_common.h"
#include "svb_q.h"
#include "conv1d_q.h"
#include <svb_glds.h>

typedef __bf16 pw_bf16x8 __attribute__((ext_vector_type(8)));


struct SvbPwArgs {
    const float* x;
    const unsigned short* wq_hi;
    const unsigned short* wq_lo;
    const float* bias;
    float* y;
    const float* out_gate;
    const float* mask;
    const float* residual;
    float out_slope, out_gate_slope;
    int out_act;
    int B, Cin, Cout, T;
    int ncols;                  // B * T
    int nph;                    // K phases: (Cin / 16) * ntap slabs in groups of P
    int w_rows;                 // rows per packed weight slab
    int m_tiles, ntiles;
    // taps (round 6b): slab g = chunk * ntap + tap reads x at position + tap_off(tap) (zero outside the clip) and the weight slab
    // (tw0 + tap * tstep) * w_tap16 + chunk * 2 w_rows (16-byte units)
    int ntap, tw0, tstep, rcp;  // rcp = ceil(2^16 / ntap): g / ntap == (g * rcp) >> 16 for g < 4096
    unsigned w_tap16;
    unsigned long long offs[4];               // sixteen signed 16-bit tap offsets
    const float* in_gate;       // optional: x is multiplied by gate'(in_gate) (1 where in_gate > 0, else in_slope); == x: self-gated
    float in_slope;
};

__device__ __forceinline__ int pw_tap_off(const SvbPwArgs& a, int tap) {
    // (selects, not an indexed read: an indexed kernel argument becomes a scratch table)
    const unsigned long long lo = (tap & 4) ? a.offs[1] : a.offs[0], hi = (tap & 4) ? a.offs[3] : a.offs[2];
    return (int)(short)(unsigned short)(((tap & 8) ? hi : lo) >> (16 * (tap & 3)));
}

// s_waitcnt immediate: vmcnt(n) only (expcnt / lgkmcnt untouched)
#define PW_VMCNT(n) ((((n) & 15) | 0x0F70 | (((n) >> 4) << 14)))

// ---- epilogue of one wave: BF column blocks of 32 (first column n_first, consecutive blocks 32 apart) x AF row blocks from m_base:
// v = act(acc + bias) [* gate'(out_gate)] [+ residual] [* mask]; a store is `uniform row offset + per-lane column offset` through
// a buffer descriptor (columns outside the tensor carry an offset the bounds check drops)
template <int AF, int BF>
__device__ __forceinline__ void pw_epilogue(const SvbPwArgs& a, f32x16 (&acc)[AF][BF], int m_base, int n_first, int lane) {
    const int kb = lane >> 5, l31 = lane & 31;
    const int p_T = a.T, p_Cout = a.Cout;
    const unsigned t4 = 4u * (unsigned)p_T;
    const unsigned y_bytes = 4u * (unsigned)(a.B * p_Cout * p_T);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, y_bytes, 0x00020000);
    const bool has_gate = a.out_gate != nullptr, has_res = a.residual != nullptr;
    const __amdgpu_buffer_rsrc_t g_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_gate ? a.out_gate : a.y), 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_res ? a.residual : a.y), 0, y_bytes, 0x00020000);
    const int act = a.out_act;
    const float slope = a.out_slope, gslope = a.out_gate_slope;
    const bool plain = !has_gate && !has_res && !a.mask && act != SVB_ACT_TANH;
    const float neg = act == SVB_ACT_RELU ? 0.f : (act == SVB_ACT_LRELU ? slope : 1.f);      // v > 0 ? v : v * neg
    unsigned yv[BF];
    float mk[BF];
#pragma unroll
    for (int j = 0; j < BF; ++j) {
        const int n = n_first + 32 * j + l31;
        const bool ok = n < a.ncols;
        const int b = n / p_T, t = n - b * p_T;
        yv[j] = ok ? 4u * (unsigned)((b * p_Cout + 4 * kb) * p_T + t) : 0x80000000u;
        mk[j] = (a.mask && ok) ? a.mask[n] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < AF; ++i) {
        const int row0 = m_base + 32 * i;
        if (row0 >= p_Cout) break;                                       // (uniform) a ragged last row tile
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row0 + (r & 3) + 8 * (r >> 2) + 4 * kb;
            bv[r] = a.bias ? a.bias[min(m, p_Cout - 1)] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < BF; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = row0 + (r & 3) + 8 * (r >> 2);             // (+ 4 kb: Cout % 8 == 0, so the pair is in or out together)
                if (mr < p_Cout) {
                    const unsigned soff = (unsigned)mr * t4;
                    float v = acc[i][n][r] + bv[r];
                    if (plain) {
                        v = v > 0.f ? v : v * neg;
                    } else {
                        v = svb_apply_act(v, act, slope);
                        if (has_gate) v *= svb_gate(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, yv[n], soff, 0)), gslope);
                        if (has_res) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, yv[n], soff, 0));
                        v *= mk[n];
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rsrc, yv[n], soff, 0);
                }
            }
    }
}

// ---- the x operand of one wave: BF column blocks of 32, chunk by chunk, PF chunks in flight: 8 dword loads per block and chunk
// (lane = position, channels 8 kb .. 8 kb + 7), split in registers one chunk ahead of its MFMAs.
// (Round 6 also measured this kernel on a PRE-SPLIT planar image of x -- [chunk][hi h0, hi h1, lo h0, lo h1][column][8 bf16], a B
//  fragment = ONE 16-byte load, no VALU work -- with timing-only data: 15-20 % per launch on the small tiles, 0.35 ms of a 12.5 ms
//  step if every producer of a pointwise conv's input emitted that image for free; profiles/r06_pwbench_qin.log.  Not built: the
//  producers are a dozen kernels on both sides of autograd, and the step does not see 0.35 ms of PPG-stream kernel time.)
template <int BF, int PF, int GATE>
struct PwX {
    float xr[PF][BF][8];
    float gr[GATE == 1 ? PF : 1][BF][8];
    uint4 bh[2][BF], bl[2][BF];
    unsigned xv[BF];             // byte offset of (clip, channel 8 kb, position) of this lane's column; >= 2^31: no such column
    int col_t[BF];               // its position inside the clip
    unsigned t4;
    int T, last_chunk;
    float slope;
    __amdgpu_buffer_rsrc_t rsrc, grsrc;

    __device__ __forceinline__ void init(const SvbPwArgs& a, int nchunk) {
        t4 = 4u * (unsigned)a.T;
        T = a.T;
        last_chunk = nchunk - 1;
        slope = a.in_slope;
        const unsigned bytes = 4u * (unsigned)(a.B * a.Cin * a.T);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, bytes, 0x00020000);
        grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATE == 1 ? a.in_gate : a.x), 0, bytes, 0x00020000);
    }
    __device__ __forceinline__ void set_columns(const SvbPwArgs& a, int n_first, int lane) {
        const int kb = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int j = 0; j < BF; ++j) {
            const int n = n_first + 32 * j + l31;
            const int b = n / a.T, t = n - b * a.T;
            col_t[j] = t;
            xv[j] = n < a.ncols ? 4u * (unsigned)((b * a.Cin + 8 * kb) * a.T + t) : 0x80000000u;
        }
    }
    // chunk c (clamped: a harmless re-read past the end) at tap offset `off` -> register set; positions outside the clip (the
    // conv's zero padding) and columns outside the tensor get an offset the buffer's bounds check answers with 0
    __device__ __forceinline__ void load(int set, int c, int off) {
        const unsigned s0 = 16u * (unsigned)min(c, last_chunk) * t4;
#pragma unroll
        for (int j = 0; j < BF; ++j) {
            const bool ok = xv[j] < 0x80000000u && (unsigned)(col_t[j] + off) < (unsigned)T;
            const unsigned v = ok ? xv[j] + 4u * (unsigned)off : 0x80000000u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xr[set][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, v, s0 + (unsigned)e * t4, 0));
                if (GATE == 1) gr[set][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grsrc, v, s0 + (unsigned)e * t4, 0));
            }
        }
    }
    __device__ __forceinline__ void split(int bp, int set) {
#pragma unroll
        for (int j = 0; j < BF; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = xr[set][j][e];
                if (GATE == 1) v[e] *= svb_gate(gr[set][j][e], slope);
                if (GATE == 2) v[e] *= svb_gate(v[e], slope);
            }
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) svbq_split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
            bh[bp][j] = make_uint4(h[0], h[1], h[2], h[3]);
            bl[bp][j] = make_uint4(l[0], l[1], l[2], l[3]);
        }
    }
};

// the 3 AF BF MFMAs of one chunk: products lo*hi, hi*lo, hi*hi per accumulator, in the family's order
template <int AF, int BF, int PF, int GATE>
__device__ __forceinline__ void pw_mfma_slab(f32x16 (&acc)[AF][BF], const uint4 (&fa)[2 * AF], const PwX<BF, PF, GATE>& X, int bp) {
#pragma unroll
    for (int prod = 0; prod < 3; ++prod)
#pragma unroll
        for (int i = 0; i < AF; ++i)
#pragma unroll
            for (int n = 0; n < BF; ++n) {
                const pw_bf16x8 ah = *reinterpret_cast<const pw_bf16x8*>(&fa[2 * i]);
                const pw_bf16x8 al = *reinterpret_cast<const pw_bf16x8*>(&fa[2 * i + 1]);
                const pw_bf16x8 xh = *reinterpret_cast<const pw_bf16x8*>(&X.bh[bp][n]);
                const pw_bf16x8 xl = *reinterpret_cast<const pw_bf16x8*>(&X.bl[bp][n]);
                if (prod == 0) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, acc[i][n], 0, 0, 0);
                else if (prod == 1) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, acc[i][n], 0, 0, 0);
                else acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, acc[i][n], 0, 0, 0);
            }
}

// ==================================================================================================================
// Phased form.  AF x BF: 32x32 accumulators per wave (workgroup tile 32 AF x 128 BF); PW_P: chunks per K phase; PF: chunks of x in
// flight per wave (PF | PW_P; 8 BF registers per chunk).
template <int AF, int BF, int PW_P, int PF, int GATE>
__global__ __launch_bounds__(256, 2) void svb_conv1d_pw_kernel(SvbPwArgs a) {
    static_assert(PW_P % PF == 0 && PW_P % 2 == 0 && PF >= 2 && (GATE == 1 ? 16 : 8) * BF * PF <= 63,
                  "register set of a chunk = chunk % PF; vmcnt counts to 63");
    constexpr int BM = 32 * AF, BN = 128 * BF;
    constexpr int SLAB16 = AF * 2 * 64;               // 16-byte units per weight slab image (AF blocks x hi|lo x 64 lanes)
    constexpr int PHASE16 = PW_P * SLAB16;
    constexpr int NPIECE = PW_P * AF * 2;             // 1-KiB DMA pieces per phase
    constexpr int PPW = (NPIECE + 3) / 4;             // ... per wave
    HIP_DYNAMIC_SHARED(uint4, smem)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = lane >> 5, l31 = lane & 31;
    const int p_nph = a.nph;

    // XCD-aware bijective remap: consecutive tile ids run on one XCD; the M tiles of a column block are consecutive ids, so the
    // column block's x rows are fetched from HBM once and re-read through that XCD's L2
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int qd = nwg >> 3, rd = nwg & 7, xcd = orig & 7;
    const int tile = (xcd < rd ? xcd * (qd + 1) : rd * (qd + 1) + (xcd - rd) * qd) + (orig >> 3);
    const int mt = tile % a.m_tiles, nt = tile / a.m_tiles;
    const int m_base = mt * BM, n_first = nt * BN + wave * BF * 32;

    const int p_ntap = a.ntap;
    PwX<BF, PF, GATE> X;
    X.init(a, a.Cin / 16);
    X.set_columns(a, n_first, lane);

    // ---- weight pieces of a phase: piece q = slab * 2 AF + block * 2 + (hi|lo); lane L copies 16 bytes of row 32 block + (L & 31),
    // channel half L >> 5 -- the 64 lanes together one contiguous KiB of the packed weight -- to slot q * 64 + L ------------
    const uint4* const w_hi16 = reinterpret_cast<const uint4*>(a.wq_hi);
    const uint4* const w_lo16 = reinterpret_cast<const uint4*>(a.wq_lo);
    unsigned w_src[PPW];         // 16-byte unit of this lane's source inside a slab, per piece of this wave
    const unsigned w_chunk16 = 2u * (unsigned)a.w_rows, w_tap16 = a.w_tap16;
    const int p_rcp = a.rcp, p_tw0 = a.tw0, p_tstep = a.tstep;
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int q = wave + 4 * u;
        const int sl = q / (2 * AF), rem = q - sl * 2 * AF, blk = rem >> 1;
        const int row = min(m_base + 32 * blk + l31, a.w_rows - 1);
        w_src[u] = 2u * (unsigned)row + (unsigned)kb;
    }
    auto issue_w = [&](int ph, int buf) {
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int q = wave + 4 * u;
            if (NPIECE % 4 == 0 || q < NPIECE) {
                const int g = ph * PW_P + q / (2 * AF);                       // slab -> (chunk, tap), tap fastest
                const int c = (g * p_rcp) >> 16, t = g - c * p_ntap;
                const unsigned slab16 = (unsigned)(p_tw0 + t * p_tstep) * w_tap16 + (unsigned)c * w_chunk16;
                const uint4* src = ((q & 1) ? w_lo16 : w_hi16) + (size_t)(slab16 + w_src[u]);
                svb_glds16(src, smem, 16u * (unsigned)(buf * PHASE16 + q * 64));
            }
        }
    };

    f32x16 acc[AF][BF];
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int n = 0; n < BF; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    const char* const lds = reinterpret_cast<const char*>(smem) + 16 * lane;
    uint4 fa[2][2 * AF];
    auto read_a = [&](int par, int buf, int s) {      // A fragments of slab s of the phase in buffer buf
#pragma unroll
        for (int r = 0; r < 2 * AF; ++r)
            fa[par][r] = *reinterpret_cast<const uint4*>(lds + 16 * (buf * PHASE16 + s * SLAB16 + r * 64));
    };

    // Software pipeline over the slabs g = 0, 1, ... (slab = (chunk, tap), tap fastest): while the MFMAs of slab g run, the x
    // registers of slab g + 1 are split (they were requested PF - 1 slabs ago), the A fragments of slab g + 1 are read, and -- FIRST,
    // pinned in front of the slab's arithmetic -- the x loads of slab g + PF are issued into the register set that has just been
    // vacated.  (Left to itself the scheduler sinks every load of a phase to its end and waits vmcnt(0) at the top of the next
    // one; and the prologue's loads are pinned in chunk order, because the wait-count pass merges the loop header's state with
    // that block's: a set requested LAST there turns the first wait of every phase into a near-complete drain of the prefetch.)
    int lc = 0, lt = 0;                               // (chunk, tap) of the next slab to request
    auto load_next = [&](int set) {
        X.load(set, lc, pw_tap_off(a, lt));
        if (++lt == p_ntap) { lt = 0; ++lc; }
    };
    issue_w(0, 0);
#pragma unroll
    for (int s = 0; s < PF; ++s) { load_next(s); __builtin_amdgcn_sched_barrier(0); }
    X.split(0, 0);
    for (int ph = 0; ph < p_nph; ++ph) {
        const int buf = ph & 1;
        // this wave's pieces of phase ph were requested a phase ago, BEFORE the x loads that may still be in flight (vmcnt retires
        // in order); after the barrier every wave's pieces have landed and nobody reads the other buffer any more
        __builtin_amdgcn_s_waitcnt(PW_VMCNT((GATE == 1 ? 16 : 8) * BF * PF));
        __builtin_amdgcn_s_barrier();
        if (ph + 1 < p_nph) issue_w(ph + 1, buf ^ 1);
        read_a(0, buf, 0);
#pragma unroll
        for (int s = 0; s < PW_P; ++s) {
            const int par = s & 1;
            load_next(s % PF);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < PW_P) read_a(par ^ 1, buf, s + 1);
            X.split(par ^ 1, (s + 1) % PF);
            pw_mfma_slab<AF, BF, PF, GATE>(acc, fa[par], X, par);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    pw_epilogue<AF, BF>(a, acc, m_base, n_first, lane);
}

// ==================================================================================================================
struct PwCfg { int AF, BF, P, PF; };
static const PwCfg kPwCfgs[SVB_PW_NVARIANTS] = {{4, 1, 2, 2}, {4, 2, 4, 2}, {2, 2, 4, 2}, {2, 1, 4, 4}, {3, 1, 4, 4}, {3, 2, 4, 2}};
//                                               128x128       128x256       64x256        64x128        96x128        96x256
// (Round 6, measured on the MI355X, profiles/r06_pwbench_*.log: the small tiles win -- these launches are one to four workgroups per
//  CU, all resident at once, and a workgroup's life is a serial chain of latencies, so what pays is MORE workgroups per CU, not more
//  reuse per workgroup: 128-row tiles with 4-chunk phases (64 KiB of LDS, two workgroups per CU) lose 20-50 % to the same tile with
//  2-chunk phases.  A weight-stationary persistent form -- all of a row tile's weights in LDS once, waves walking column blocks with
//  no barrier -- was 10-15 % SLOWER than the phased small tiles on every shape and is not in the tree.)

template <int AF, int BF, int P, int PF, int GATE>
static void pw_launch_kernel(const SvbPwArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)2 * P * AF * 2 * 64 * 16;
    hipLaunchKernelGGL((svb_conv1d_pw_kernel<AF, BF, P, PF, GATE>), dim3(a.ntiles), dim3(256), lds, stream, a);
}

template <int AF, int BF, int P, int PF>
static int pw_launch_gate(const SvbPwArgs& a, hipStream_t stream) {
    if (a.in_gate && a.in_gate == a.x) {
        if constexpr (AF * BF < 8) pw_launch_kernel<AF, BF, P, PF, 2>(a, stream);
        else return SVB_ERR_UNSUPPORTED;             // (128x256: the gate's temporaries spill)
    } else if (a.in_gate) {
        // a second operand stream: two register sets at most, 128-column tiles only
        if constexpr (BF == 1) pw_launch_kernel<AF, BF, P, 2, 1>(a, stream);
        else return SVB_ERR_UNSUPPORTED;
    } else pw_launch_kernel<AF, BF, P, PF, 0>(a, stream);
    return SVB_OK;
}

int svb_pw_launch(const SvbConvQArgs& q, const SvbConvPlan& p, int variant, hipStream_t stream) {
    if (variant < 0 || variant >= SVB_PW_NVARIANTS) return SVB_ERR_UNSUPPORTED;
    if (p.n_phase != 1 || q.sx != 1 || q.out_stride != 1 || q.G != 1 || q.xq) return SVB_ERR_UNSUPPORTED;
    const int t0 = p.phase_start[0], ntap = p.phase_start[1] - t0;
    if (ntap < 1 || ntap > 16) return SVB_ERR_UNSUPPORTED;
    if (p.phase_out_base[0] != 0 || p.phase_nq[0] != q.Tout || q.Tin != q.Tout) return SVB_ERR_UNSUPPORTED;
    const PwCfg c = kPwCfgs[variant];
    const int nslab = (q.Cin / 16) * ntap;
    if (q.Cin % 16 || nslab % c.P || q.Cout % 8 || q.Cout < 32 || nslab >= 4096) return SVB_ERR_UNSUPPORTED;
    // 32-bit byte offsets inside the kernel; an offset >= 2^31 marks a column outside the tensor
    if ((long)q.B * q.Cin * q.Tin >= (1L << 28) || (long)q.B * q.Cout * q.Tout >= (1L << 28)) return SVB_ERR_UNSUPPORTED;
    if ((long)q.w_tap_slabs * q.w_slab_rows * 2 * (SVB_MAX_TAPS + 1) >= (1L << 31)) return SVB_ERR_UNSUPPORTED;
    SvbPwArgs a;
    memset(&a, 0, sizeof(a));
    a.ntap = ntap;
    a.tw0 = p.tap_w[t0];
    a.tstep = ntap > 1 ? p.tap_w[t0 + 1] - p.tap_w[t0] : 0;
    for (int t = 0; t < ntap; ++t) {
        const int off = p.tap_off[t0 + t];
        if (p.tap_w[t0 + t] != a.tw0 + t * a.tstep || off < -32768 || off > 32767) return SVB_ERR_UNSUPPORTED;
        const unsigned long long v = (unsigned long long)(unsigned short)(short)off << (16 * (t & 3));
        a.offs[t >> 2] |= v;
    }
    a.rcp = (65536 + ntap - 1) / ntap;
    a.w_tap16 = (unsigned)q.w_tap_slabs * (unsigned)q.w_slab_rows * 2u;
    a.x = q.x; a.wq_hi = q.wq_hi; a.wq_lo = q.wq_lo; a.bias = q.bias; a.y = q.y;
    a.in_gate = q.in_gate; a.in_slope = q.in_slope;
    a.out_gate = q.out_gate; a.mask = q.mask; a.residual = q.residual;
    a.out_slope = q.out_slope; a.out_gate_slope = q.out_gate_slope; a.out_act = q.out_act;
    a.B = q.B; a.Cin = q.Cin; a.Cout = q.Cout; a.T = q.Tin;
    a.ncols = q.B * q.Tin;
    a.nph = nslab / c.P;
    a.w_rows = q.w_slab_rows;
    a.m_tiles = svb_cdiv(q.Cout, 32 * c.AF);
    const long n_tiles = ((long)a.ncols + 128 * c.BF - 1) / (128 * c.BF);
    if (a.m_tiles * n_tiles >= (1L << 30)) return SVB_ERR_UNSUPPORTED;
    a.ntiles = (int)(a.m_tiles * n_tiles);
    int rc;
    switch (variant) {
        case 0: rc = pw_launch_gate<4, 1, 2, 2>(a, stream); break;
        case 1: rc = pw_launch_gate<4, 2, 4, 2>(a, stream); break;
        case 2: rc = pw_launch_gate<2, 2, 4, 2>(a, stream); break;
        case 3: rc = pw_launch_gate<2, 1, 4, 4>(a, stream); break;
        case 4: rc = pw_launch_gate<3, 1, 4, 4>(a, stream); break;
        default: rc = pw_launch_gate<3, 2, 4, 2>(a, stream); break;
    }
    if (rc != SVB_OK) return rc;
    if (hipGetLastError() != hipSuccess) return SVB_ERR_UNSUPPORTED;
    return SVB_OK;
}
