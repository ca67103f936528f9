// conv1d_wgrad_pw.hip -- weight gradient of the 1-tap (pointwise) convs as a GEMM with DIRECT operands (round 6)
//
//        dW[Cout x Cin] = sum over (clip, position)  dy[Cout x n] . x[Cin x n]^T
//
// (every nn.Linear / Conv1d(k=1) of the path: reference modules/fastspeech/conformer/layers.py:182-258, espnet_transformer_attn.py:125-186,
// fs2_vae.py:66-91 -- their .weight.grad).  Same arithmetic as the tap-group kernel of conv1d_bf16.hip: operands split v = hi + lo into
// bf16, products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate, per-split partial gradients + bias partials that
// svb_wgrad_reduce* sums in a fixed order.  What differs is the machine mapping:
//
//   * the contraction runs over positions, and positions are the CONTIGUOUS axis of both tensors: the MFMA operand of a lane -- eight
//     consecutive positions of one channel row -- is 32 contiguous bytes in memory.  Both operands are therefore loaded straight
//     global -> VGPR (two 16-byte buffer loads per 32x16 fragment), gated and split in registers, and fed to the MFMA: no LDS, no
//     barrier, no staging transpose.  The tap-group kernel stages 64-position chunks through VGPRs -> split -> LDS with two barriers per
//     chunk; for one tap it has 12 .. 24 MFMAs per wave between barriers and ran at 85 .. 105 TFLOP/s (10 % of the MFMA peak at three
//     products per MAC);
//   * a wave owns AT x BT accumulator tiles of 32 x 32 (wave tile 64 x 64: 12 MFMAs per 16-position step against 32 loaded values
//     per lane; 64 x 32 / 32 x 64 for narrow or gated gradients), a workgroup 2 x 2 waves; rows two waves share are fetched twice, the second time from L1;
//   * steps of 16 positions are prefetched PF (2 .. 4) deep in registers (loads pinned in front of each step's arithmetic); a clip's ragged last
//     step loads element-wise with positions past the clip masked by the buffer bounds check; steps wholly past the clip are skipped.
//
// Domain: k == 1, stride 1, groups == 1, pad == 0, TA == TB, CA >= 64, CB >= 64, tensors below 2 GiB, and (T <= 512 or both channel
// counts multiples of 128: see svb_wgrad_pw_plan).  Everything else: conv1d_bf16.hip.
#include "svb_common.h"
#include "svb_q.h"
#include "conv1d_q.h"

typedef __bf16 wp_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wp_u4 __attribute__((ext_vector_type(4)));

#define WP_OOB 0x80000000u

// position walk of one split: chunks vsplit, vsplit + nsplit, ... of B x chunks_per_b 64-position chunks (the tap-group kernel's
// assignment: same partial sums per split), each in steps of 16 positions that start inside the clip
struct WpWalk {
    int chunk, total, nsplit, cpb, T;
    int b, t;                 // clip and first position of the current step
    int left;                 // steps left in the current chunk after this one
    bool valid;
    __device__ __forceinline__ void enter() {
        valid = chunk < total;
        if (!valid) return;
        b = chunk / cpb;
        t = (chunk - b * cpb) * 64;
        const int n = (T - t + 15) >> 4;
        left = (n < 4 ? n : 4) - 1;
    }
    __device__ __forceinline__ void init(int vsplit, const SvbWgradQArgs& a) {
        chunk = vsplit; total = a.total_chunks; nsplit = a.nsplit; cpb = a.chunks_per_b; T = a.TA;
        enter();
    }
    __device__ __forceinline__ void advance() {
        if (left > 0) { --left; t += 16; return; }
        chunk += nsplit;
        enter();
    }
};

// AT x BT: accumulator tiles per wave.  GA: dy carries a gate tensor (dy * lrelu'(a_gate)).  GB: 0 = x as it is, 1 = gate tensor, 2 = x is
// its own gate (the gradient of conv(leaky_relu(x))).  PF: steps in flight.
template <int AT, int BT, int GA, int GB, int PF>
__global__ __launch_bounds__(256, 2) void svb_wgrad_pw_kernel(SvbWgradQArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int kb = lane >> 5, l31 = lane & 31;
    // XCD-aware work ids (see the tap-group kernel): the tiles of one split -- which read the same positions -- share an XCD's L2
    const int orig = blockIdx.x + gridDim.x * blockIdx.y, nwg = gridDim.x * gridDim.y;
    const int qd = nwg >> 3, rd = nwg & 7, xcd = orig & 7;
    const int wgid = a.xcd_map ? (xcd < rd ? xcd * (qd + 1) : rd * (qd + 1) + (xcd - rd) * qd) + (orig >> 3) : orig;
    const int vsplit = wgid / (int)gridDim.x;
    const int idx = wgid - vsplit * (int)gridDim.x;
    const int bt = idx % a.b_tiles, at = idx / a.b_tiles;
    const int a0 = (at * 2 + wm) * 32 * AT, b0 = (bt * 2 + wn) * 32 * BT;       // this wave's first dy row / x row
    const int T = a.TA, CA = a.CA, CB = a.CB;

    const unsigned a_bytes = 4u * (unsigned)(a.B * CA * T), b_bytes = 4u * (unsigned)(a.B * CB * T);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.a), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b), 0, b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rga = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GA ? a.a_gate : a.a), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rgb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GB == 1 ? a.b_gate : a.b), 0, b_bytes, 0x00020000);
    const float a_slope = a.a_slope, b_slope = a.b_slope;

    unsigned a_row[AT], b_row[BT];        // byte offset of (row, position 8 kb) inside a clip; rows past the tensor repeat the last one
#pragma unroll
    for (int i = 0; i < AT; ++i) a_row[i] = 4u * (unsigned)(min(a0 + 32 * i + l31, CA - 1) * T + 8 * kb);
#pragma unroll
    for (int i = 0; i < BT; ++i) b_row[i] = 4u * (unsigned)(min(b0 + 32 * i + l31, CB - 1) * T + 8 * kb);

    f32x16 acc[AT][BT];
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int j = 0; j < BT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[AT];
#pragma unroll
    for (int i = 0; i < AT; ++i) bsum[i] = 0.f;

    float xa[PF][AT][8], xb[PF][BT][8];
    float ga[GA ? PF : 1][AT][8], gb[GB == 1 ? PF : 1][BT][8];

    auto load_rows = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned row, unsigned soff, bool full, int t, float (&v)[8]) {
        if (full) {
            const wp_u4 p = __builtin_bit_cast(wp_u4, __builtin_amdgcn_raw_buffer_load_b128(rs, row, soff, 0));
            const wp_u4 q = __builtin_bit_cast(wp_u4, __builtin_amdgcn_raw_buffer_load_b128(rs, row + 16u, soff, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned pe = p[e], qe = q[e];      // (a bit_cast straight from a vector element reads element 0 -- clang)
                v[e] = __builtin_bit_cast(float, pe);
                v[4 + e] = __builtin_bit_cast(float, qe);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = t + 8 * kb + e < T;
                v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? row + 4u * (unsigned)e : WP_OOB, soff, 0));
            }
        }
    };
    auto load = [&](int set, const WpWalk& w) {
        const bool full = w.t + 16 <= T;                                   // (uniform)
        const unsigned sa = 4u * (unsigned)(w.b * CA * T + w.t), sb = 4u * (unsigned)(w.b * CB * T + w.t);
#pragma unroll
        for (int i = 0; i < AT; ++i) {
            load_rows(ra, a_row[i], sa, full, w.t, xa[set][i]);
            if (GA) load_rows(rga, a_row[i], sa, full, w.t, ga[GA ? set : 0][i]);
        }
#pragma unroll
        for (int i = 0; i < BT; ++i) {
            load_rows(rb, b_row[i], sb, full, w.t, xb[set][i]);
            if (GB == 1) load_rows(rgb, b_row[i], sb, full, w.t, gb[GB == 1 ? set : 0][i]);
        }
    };

    uint4 fah[AT], fal[AT], fbh[BT], fbl[BT];
    auto split = [&](int set) {
#pragma unroll
        for (int i = 0; i < AT; ++i) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = xa[set][i][e];
                if (GA) v[e] *= svb_gate(ga[GA ? set : 0][i][e], a_slope);
                bsum[i] += v[e];
            }
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) svbq_split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
            fah[i] = make_uint4(h[0], h[1], h[2], h[3]);
            fal[i] = make_uint4(l[0], l[1], l[2], l[3]);
        }
#pragma unroll
        for (int i = 0; i < BT; ++i) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = xb[set][i][e];
                if (GB == 1) v[e] *= svb_gate(gb[GB == 1 ? set : 0][i][e], b_slope);
                if (GB == 2) v[e] *= svb_gate(v[e], b_slope);
            }
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) svbq_split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
            fbh[i] = make_uint4(h[0], h[1], h[2], h[3]);
            fbl[i] = make_uint4(l[0], l[1], l[2], l[3]);
        }
    };
    auto mfma = [&]() {
#pragma unroll
        for (int prod = 0; prod < 3; ++prod)
#pragma unroll
            for (int i = 0; i < AT; ++i)
#pragma unroll
                for (int j = 0; j < BT; ++j) {
                    const wp_bf16x8 ah = *reinterpret_cast<const wp_bf16x8*>(&fah[i]), al = *reinterpret_cast<const wp_bf16x8*>(&fal[i]);
                    const wp_bf16x8 bh = *reinterpret_cast<const wp_bf16x8*>(&fbh[j]), bl = *reinterpret_cast<const wp_bf16x8*>(&fbl[j]);
                    if (prod == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
                    else if (prod == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
                }
    };

    // ---- pipeline: set u holds the step consumed next; right after its registers are split the step PF ahead is requested into it
    WpWalk w;
    w.init(vsplit, a);
    int pending = 0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        if (w.valid) {
            load(u, w);
            ++pending;
            w.advance();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    while (pending > 0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (pending > 0) {                                              // (uniform)
                split(u);
                --pending;
                if (w.valid) {
                    load(u, w);
                    ++pending;
                    w.advance();
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- this split's partial gradient [CA][CB] and bias partial [CA]
    float* part = a.part + (size_t)vsplit * CA * CB;
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            const int ci = b0 + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = a0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kb;
                if (co < CA && ci < CB) part[(size_t)co * CB + ci] = acc[i][j][r];
            }
        }
    if (a.bias_part != nullptr && bt == 0 && wn == 0) {
#pragma unroll
        for (int i = 0; i < AT; ++i) {
            const float v = bsum[i] + __shfl_xor(bsum[i], 32);
            const int co = a0 + 32 * i + l31;
            if (kb == 0 && co < CA) a.bias_part[(size_t)vsplit * CA + co] = v;
        }
    }
}

template <int AT, int BT, int GA, int GB, int PF>
static void wp_launch_kernel(const SvbWgradQArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((svb_wgrad_pw_kernel<AT, BT, GA, GB, PF>), grid, dim3(256), 0, st, a);
}

template <int AT, int BT>
static void wp_launch_gates(SvbWgradQArgs& a, dim3 grid, hipStream_t st) {
    const bool self_b = a.b_gate && a.b_gate == a.b;
    constexpr int PF0 = AT * BT == 4 ? 3 : 4;          // steps in flight, ungated: what 256 VGPRs hold beside the accumulators
    if (!a.a_gate && !a.b_gate) wp_launch_kernel<AT, BT, 0, 0, PF0>(a, grid, st);
    else if (!a.a_gate && self_b) wp_launch_kernel<AT, BT, 0, 2, PF0>(a, grid, st);
    else if constexpr (AT * BT < 4) {                  // (gate tensors never reach the 64 x 64 wave tile: svb_wgrad_pw_launch)
        if (a.a_gate && !a.b_gate) wp_launch_kernel<AT, BT, 1, 0, 2>(a, grid, st);
        else {
            // the general form carries both gate streams; a missing one reads its own operand with slope 1 (a factor of exactly 1)
            if (!a.a_gate) { a.a_gate = a.a; a.a_slope = 1.f; }
            if (!a.b_gate) { a.b_gate = a.b; a.b_slope = 1.f; }
            wp_launch_kernel<AT, BT, 1, 1, 2>(a, grid, st);
        }
    }
}

// wave tile for a CA x CB gradient: the padded MFMA work of each candidate over its relative efficiency (measured ordering: wider
// tiles do more arithmetic per loaded value)
//
// Where it is used (measured on the MI355X, B = 32, profiles/r06_conv_per_shape_wgpw*.log, us per launch, tap-group kernel -> this one):
//   1536 -> 256, T 1124: 288 -> 236     3072 -> 256, T 281: 165 -> 133     256 -> 768, T 1124: 134 -> 117     256 -> 256, T 1124: 55 -> 54
//   192 -> 384, T 281: 32.4 -> 26.4     192 -> 256, T 281: 25.5 -> 20.6    192 -> 192, T 281: 22.5 -> 19.0
//   192 -> 384, T 1124: 62.6 -> 69.7 (lost)     192 -> 192, T 1124: 44.7 -> 51 (lost)
// i.e. it wins on short clips (the tap-group kernel's 64-position chunks are per clip: T = 281 is 4.4 of them) and on gradients
// that fill 128 x 128 tiles, and loses where a quarter of a tile is padding.  It is NOT the 2x a direct-operand GEMM promises on paper:
// a fragment load touches 64 different cache lines per instruction (one row per lane), eight times the tag lookups of a coalesced
// load, and the kernel sits on the L1 tag rate at ~120 TFLOP/s whatever the prefetch depth (2 / 3 / 4 steps: same time).  The form that
// would beat both kernels feeds LDS by DMA with row-contiguous reads and a multi-stage ring; it is not built.
bool svb_wgrad_pw_plan(int CA, int CB, int groups, int k, int sx, int pad, int dil, int TA, int* at, int* bt) {
    (void)dil;
    if (groups != 1 || k != 1 || sx != 1 || pad != 0 || CA < 64 || CB < 64) return false;
    if (TA > 512 && (CA % 128 || CB % 128)) return false;
    // (a 64 x 96 wave tile spills at two steps in flight: 96 accumulator + 80 operand + 40 fragment registers)
    static const int cand[3][2] = {{2, 2}, {2, 1}, {1, 2}};
    static const float eff[3] = {1.f, 0.8f, 0.8f};
    float best = 0.f;
    for (int c = 0; c < 3; ++c) {
        const int ta = svb_cdiv(CA, 64 * cand[c][0]), tb = svb_cdiv(CB, 64 * cand[c][1]);
        const float cost = (float)ta * tb * cand[c][0] * cand[c][1] / eff[c];
        if (c == 0 || cost < best) { best = cost; *at = cand[c][0]; *bt = cand[c][1]; }
    }
    return true;
}

int svb_wgrad_pw_launch(const SvbWgradQArgs& q, hipStream_t stream) {
    SvbWgradQArgs a = q;
    int at = 0, bt = 0;
    if (!svb_wgrad_pw_plan(a.CA, a.CB, a.G, a.k, a.sx, -a.off0, a.dil, a.TA, &at, &bt) || a.TA != a.TB || a.gp_ca) return SVB_ERR_UNSUPPORTED;
    if ((long)a.B * a.CA * a.TA >= (1L << 29) || (long)a.B * a.CB * a.TB >= (1L << 29)) return SVB_ERR_UNSUPPORTED;
    // gate tensors are a second operand stream in registers: the 64 x 64 wave tile does not hold them (a self-gated x needs none)
    if ((a.a_gate || (a.b_gate && a.b_gate != a.b)) && at == 2 && bt == 2) {
        if (a.a_gate && !a.b_gate) at = 1;
        else bt = 1;
    }
    a.at = at; a.bt = bt;
    a.a_tiles = svb_cdiv(a.CA, 64 * at); a.b_tiles = svb_cdiv(a.CB, 64 * bt);
    dim3 grid(a.a_tiles * a.b_tiles, a.nsplit);
    if (at == 2 && bt == 2) wp_launch_gates<2, 2>(a, grid, stream);
    else if (at == 2 && bt == 1) wp_launch_gates<2, 1>(a, grid, stream);
    else wp_launch_gates<1, 2>(a, grid, stream);
    if (hipGetLastError() != hipSuccess) return SVB_ERR_UNSUPPORTED;
    return SVB_OK;
}
