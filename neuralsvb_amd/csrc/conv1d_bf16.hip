// conv1d_bf16.hip -- the implicit-GEMM conv family on the *bf16* matrix cores with an fp32-class split ("bf16x3").
//
// Same tap-offset formulation, tile shapes, plan and fused prologue/epilogue as conv1d.hip, but every fp32 operand is
// split on the fly into two bf16 values  v = hi + lo  (hi = rne_bf16(v), lo = rne_bf16(v - hi)) and each product is
// evaluated as  hi*hi + hi*lo + lo*hi  on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 3 MFMAs of 32 cycles per 16
// K-values instead of 8 fp32 MFMAs of 64 cycles (5.3x the fp32-MFMA rate).  The dropped lo*lo term is 2^-16 relative;
// measured on the whole MleSVBVAE forward the mel-L1 against the fp32 result is 2.7e-5 (gate: 1e-4).
//
// Layouts (chosen so that every MFMA operand is ONE aligned 16-byte LDS read, bank-conflict free):
//   weights  (global, packed by svb_weight_pack_bf16x3):  [tap][k-chunk of 16 channels][m][16 bf16]  (hi and lo arrays)
//   W tile   (LDS):  [slab = tap*kch + chunk][m][16 bf16], row pitch 48 B   (lane: row m = l&31, bytes 16*(l>>5))
//   x tile   (LDS):  [chunk][position][16 bf16],           row pitch 48 B   (transposed + converted while staging:
//                    a thread loads 8 channels of one position with 8 coalesced dword loads and issues two 16-byte
//                    LDS stores, hi and lo)
// 48-byte rows put the 16 lanes of every ds_read_b128 service group on 16 disjoint 4-bank windows.
#include "svb_common.h"
#include "svb_q.h"
#include "conv1d.h"
#include "conv1d_q.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SVBQ_PITCH 24          /* bf16 elements per LDS row (16 used + 8 pad) = 48 bytes */
#define SVBQ_XUNITS 4          /* register-staged x units per thread: (16-channel chunks per phase) x (128-position groups) */
#define SVBQ_QUNITS 8          /* Q-input staging: 16-byte units per thread and phase */


static const bool g_svbq_nofast = SVB_ENV_FLAG("SVB_NO_FASTLOOP");          // A/B switches of the instrumentation build
static const int g_svbq_lds_pad = SVB_ENV_INT("SVB_LDS_PAD_KB", 0) * 1024;    // extra LDS -> 1 workgroup per CU
static const bool g_svbq_wg_narrow = SVB_ENV_FLAG("SVB_WGRAD_NARROW");        // 64x64 weight-gradient tiles only
#ifdef SVB_INSTRUMENT
static unsigned long long* g_svbq_dbg = nullptr;
static const int g_svbq_prio = SVB_ENV_INT("SVB_CONV_PRIO", 0);
extern "C" void svb_debug_set_timing_buffer(void* p) { g_svbq_dbg = (unsigned long long*)p; }
static const int g_svbq_dbg_block0 = SVB_ENV_INT("SVB_DBG_BLOCK0", 0);        // first sampled workgroup
#define SVBQ_DBG_BLOCKS 64
#define SVBQ_DBG_STAGES 32
#define SVBQ_STAMP(slot)                                                                                            \
    if (a.dbg && tid == 0 && dbg_id >= 0 && dbg_id < SVBQ_DBG_BLOCKS && dbg_stage < SVBQ_DBG_STAGES)                 \
        a.dbg[((size_t)dbg_id * SVBQ_DBG_STAGES + dbg_stage) * 8 + (slot)] = __builtin_readcyclecounter();
#define SVBQ_PRIO(cond, level) if (cond) __builtin_amdgcn_s_setprio(level);
#define SVBQ_NEXT_STAGE ++dbg_stage;
#define SVBQ_LAST_STAGE dbg_stage = SVBQ_DBG_STAGES - 1;
#else
#define SVBQ_NEXT_STAGE
#define SVBQ_LAST_STAGE
#define SVBQ_STAMP(slot)
#define SVBQ_PRIO(cond, level)
#endif

// load base[byte_off]: `base` wave-uniform, byte_off a 32-bit per-lane offset (scalar-base + vector-offset addressing)
__device__ __forceinline__ float svbq_ld(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ void svbq_split(float v, unsigned& hi, unsigned& lo) {
    unsigned h2, l2;
    svbq_split2(v, 0.f, h2, l2);
    hi = h2 & 0xFFFFu;
    lo = l2 & 0xFFFFu;
}
__device__ __forceinline__ void svbq_split8(const float* v, uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) svbq_split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// MODE 0: fp32 x, split while staging;  1: the same with the activation-derivative gate on the load;  2: x comes pre-split
// (Q image): the tile is staged with 16-byte copies -- four consecutive lanes fetch the 64-byte row of one position.
// (Round 3 also measured MODE 0 / 1 tiles read with 16-byte loads ALONG time -- one thread = 4 positions of a channel pair, a
//  quarter of the VMEM instructions, the tile written to LDS as 8 ds_write_b32 per unit instead of 2 ds_write_b128: every conv
//  shape of the step got 15-30 % SLOWER (profiles/r03_conv_quad_staging_ab.log), so the VMEM instruction count of the x tile
//  is not what bounds the kernel -- removed.)
// (Round 3 measured the gate / res-skip / gate-backward epilogue variants of this kernel on the MI355X: fewer launches, but
//  0.2-0.35 ms MORE kernel time per step than the streaming kernels they replaced -- profiles/r03_fused_epilogue_ab.log -- removed.)
template <int WM, int WN, int NT, int SLB, int MODE>
__global__ __launch_bounds__(256, 2) void svb_conv1d_bf16x3_kernel(SvbConvQArgs a, SvbConvPlan p) {
    // MODE 3 / 4 = MODE 0 / 1 with ONE bf16 product per operand pair (hi * hi only: plain bf16 arithmetic with fp32 accumulation,
    // `conv_precision: bf16`, a secondary, reduced-precision line -- svb_conv_set_single_product); the cross products are compiled out
    // MODE 5 = MODE 1 whose gate tensor IS the input (`conv(leaky_relu(x))`, in_gate == x -- every conv of the HifiGAN generator):
    // the activation derivative is taken from the value just loaded instead of from a second load of the same element, which
    // halves the x tile's global loads (the staging of that tile is what bounds these kernels).  Same arithmetic: bit-identical.
    constexpr bool SINGLE = MODE == 3 || MODE == 4;
    constexpr bool SELFG = MODE == 5;
    constexpr bool GATE = MODE == 1 || MODE == 4 || MODE == 5, QIN = MODE == 2;
    constexpr int BM = 32 * WM, BN = 32 * WN * NT;
    constexpr int WTASKS = SLB * BM * 2;                 // 16-byte units per (hi|lo) weight tile
    constexpr int WU = (2 * WTASKS + 255) / 256;         // per-thread units, hi and lo together
    // "direct-A" instantiations (SLB <= 5): every wave reads the MFMA A operands of its 32 weight rows straight from global
    // memory (waves that share rows -- WN > 1 -- repeat the read: 1 KiB per slab, an L1/L2 hit); SLB = 8 stages weights in LDS
    constexpr bool DIRECT_A = SLB <= 5;
    static_assert(WM * WN == 4, "256 threads = 4 waves");
    static_assert(DIRECT_A || WTASKS % 256 == 0, "a staging unit index u addresses either the hi or the lo weight array");
    HIP_DYNAMIC_SHARED(uint4, dyn_smem)
    uint4* w_hi = dyn_smem;
    uint4* w_lo = w_hi + a.w_floats16;
    uint4* x_hi = w_lo + a.w_floats16;
    uint4* x_lo = x_hi + a.x_floats16;
    int* tap_lds = reinterpret_cast<int*>(x_lo + a.x_floats16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int kb = lane >> 5, l31 = lane & 31;

    const int nwg = gridDim.x * gridDim.y;
    const int orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int qd = nwg >> 3, rd = nwg & 7, xcd = orig & 7;
    const int wgid = (xcd < rd ? xcd * (qd + 1) : rd * (qd + 1) + (xcd - rd) * qd) + (orig >> 3);
    const int mt = wgid % gridDim.x, qt = wgid / gridDim.x;
#ifdef SVB_INSTRUMENT
    const int dbg_id = a.dbg ? (int)blockIdx.z * nwg + orig - a.dbg_block0 : -1;      // launch order, for the stage stamps
    int dbg_stage = 0;
#endif
    const int m_tiles_g = gridDim.x / a.G;
    const int g = mt / m_tiles_g, mtile = mt % m_tiles_g;
    const int b = blockIdx.z / p.n_phase, ph = blockIdx.z % p.n_phase;
    const int nq = p.phase_nq[ph];
    const int q0 = qt * BN;
    if (q0 >= nq) return;

    const int t0 = p.phase_start[ph], ntap = p.phase_start[ph + 1] - t0;
    const int min_off = p.phase_min_off[ph];
    const int lo_pos = q0 * a.sx + min_off;
    const int span = (BN - 1) * a.sx + p.phase_span_off[ph] + 1;
    if (tid < ntap) tap_lds[tid] = p.tap_off[t0 + tid] - min_off;

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    const int m_base = mtile * BM;
    const int m_valid = min(BM, a.Cout_g - m_base);
    const float* xb = a.x + ((size_t)b * a.Cin + (size_t)g * a.Cin_g) * a.Tin;
    const float* gb = GATE ? a.in_gate + ((size_t)b * a.Cin + (size_t)g * a.Cin_g) * a.Tin : nullptr;
    const int tap_step = ntap > 1 ? (p.tap_w[t0 + 1] - p.tap_w[t0]) : 0;
    const size_t slab_elems = (size_t)a.w_slab_rows * 16;                      // bf16 per slab
    const size_t w_off0 = ((size_t)p.tap_w[t0 < SVB_MAX_TAPS ? t0 : 0] * a.w_tap_slabs + (size_t)g * a.w_g_slabs) * slab_elems +
                          ((size_t)g * a.w_goff_m + m_base) * 16;

    // ---- hoisted per-thread decode of the weight-tile units: unit -> (hi|lo, tap t, chunk c, row m, half h) ---------
    int w_t[WU], w_c[WU], w_src[WU], w_dst[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int unit = u * 256 + tid;
        const int arr = unit / WTASKS, r = unit - arr * WTASKS;
        const int slab = r / (BM * 2), rem = r - slab * (BM * 2);
        const int mm = rem >> 1, h = rem & 1;
        const int t = slab / a.kch, c = slab - t * a.kch;
        const bool ok = arr < 2 && t < a.tg && mm < m_valid;
        w_t[u] = ok ? t : (1 << 20);
        w_c[u] = c;
        w_src[u] = (int)((size_t)t * tap_step * a.w_tap_slabs * slab_elems / 8 + (size_t)c * slab_elems / 8 + mm * 2 + h) ;   // in 16-byte units
        w_dst[u] = (arr == 1 ? a.w_floats16 : 0) + (slab * BM + mm) * 3 + h;   // 48-byte rows = 3 x 16 bytes
    }
    // x staging roles: threads 0..127 own channel half 0, 128..255 half 1; position = (tid & 127) + 128 * it
    const int xh = tid >> 7, xp0 = tid & 127;

    int xw = 0;                                            // x-tile buffer the staging lambdas write (16-byte units)
    uint4 wr[WU];
    float xr[SVBQ_XUNITS][8];
    int u_c[SVBQ_XUNITS], u_it[SVBQ_XUNITS];               // unit u -> (chunk u / xit, 128-position group u % xit)
#pragma unroll
    for (int u = 0; u < SVBQ_XUNITS; ++u) { u_c[u] = u / a.xit; u_it[u] = u - u_c[u] * a.xit; }

    auto load_w = [&](int kc0, int tg0) {
        const int nt_here = min(a.tg, ntap - tg0);
        const int kch_here = min(a.kch, a.kchunks - kc0);
        const size_t base16 = (w_off0 + ((size_t)tg0 * tap_step * a.w_tap_slabs + (size_t)kc0) * slab_elems) / 8;
        const uint4* src_hi = reinterpret_cast<const uint4*>(a.wq_hi) + base16;
        const uint4* src_lo = reinterpret_cast<const uint4*>(a.wq_lo) + base16;
        // branch-free: units that are not part of this phase read slot 0 (always in bounds) and are never consumed, so
        // every load of the phase is in flight before the first wait
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const bool ok = w_t[u] < nt_here && w_c[u] < kch_here;
            const uint4* sp = (u * 256) < WTASKS ? src_hi : src_lo;
            const uint4 t = sp[ok ? w_src[u] : 0];
            wr[u] = t;
        }
    };
    auto store_w = [&](int tg0) {
        const int nt_here = min(a.tg, ntap - tg0);
#pragma unroll
        for (int u = 0; u < WU; ++u)
            if (w_t[u] < nt_here) w_hi[w_dst[u]] = wr[u];       // w_dst already carries the hi/lo array offset
    };
    // hoisted per-thread staging invariants: the position part of each unit's offset and its validity do not depend
    // on the channel chunk, so a load is one scalar-base + vector-offset instruction with no per-element address math
    unsigned x_off[SVBQ_XUNITS];      // BYTE offset (32-bit, zero-extended onto a wave-uniform base pointer)
    bool x_ok[SVBQ_XUNITS];
#pragma unroll
    for (int u = 0; u < SVBQ_XUNITS; ++u) {
        const int i = xp0 + 128 * u_it[u];
        const int pos = lo_pos + i;
        x_ok[u] = i < span && pos >= 0 && pos < a.Tin;
        x_off[u] = x_ok[u] ? 4u * (unsigned)(xh * 8 * a.Tin + pos) : 0u;
    }
    // register-staged path: phases whose chunks are all full 16-channel chunks (a ragged last chunk takes stage_x_slow)
    auto phase_fast = [&](int kc0) { return a.fast_x && (kc0 + min(a.kch, a.kchunks - kc0)) * 16 <= a.Cin_g; };
    auto load_x = [&](int kc0) {
        const int kch_here = min(a.kch, a.kchunks - kc0);
#pragma unroll
        for (int u = 0; u < SVBQ_XUNITS; ++u) {
            const int c = u_c[u];
            if (c < kch_here) {
                const int chb = (kc0 + c) * 16;                           // wave-uniform
                const float* pc = xb + (size_t)chb * a.Tin;
                // branch-free: out-of-range lanes read element 0 of the channel row and are zeroed when staged
#pragma unroll
                for (int e = 0; e < 8; ++e) xr[u][e] = svbq_ld(pc + (size_t)e * a.Tin, x_off[u]);
                if constexpr (SELFG) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xr[u][e] *= svb_gate(xr[u][e], a.in_slope);
                } else if (GATE) {
                    const float* gc = gb + (size_t)chb * a.Tin;
#pragma unroll
                    for (int e = 0; e < 8; ++e) xr[u][e] *= svb_gate(svbq_ld(gc + (size_t)e * a.Tin, x_off[u]), a.in_slope);
                }
            }
        }
    };
    auto store_x = [&](int kc0) {
        const int kch_here = min(a.kch, a.kchunks - kc0);
#pragma unroll
        for (int u = 0; u < SVBQ_XUNITS; ++u) {
            const int c = u_c[u];
            const int i = xp0 + 128 * u_it[u];
            if (c < kch_here && i < span) {
                uint4 hi, lo;
                svbq_split8(xr[u], hi, lo);
                if (!x_ok[u]) { hi = make_uint4(0u, 0u, 0u, 0u); lo = hi; }
                const int d = (c * a.xrows + i) * 3 + xh;
                x_hi[xw + d] = hi;
                x_lo[xw + d] = lo;
            }
        }
    };
    auto stage_x_slow = [&](int kc0) {      // wide (strided) spans and ragged channel tails: direct, unpipelined
        const int kch_here = min(a.kch, a.kchunks - kc0);
        for (int c = 0; c < kch_here; ++c) {
            const int ch0 = (kc0 + c) * 16 + xh * 8;
            for (int i = xp0; i < span; i += 128) {
                const int pos = lo_pos + i;
                float vv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = 0.f;
                    if (pos >= 0 && pos < a.Tin && ch0 + e < a.Cin_g) {
                        const size_t off = (size_t)(ch0 + e) * a.Tin + pos;
                        v = xb[off];
                        if (gb) v *= svb_gate(gb[off], a.in_slope);
                    }
                    vv[e] = v;
                }
                uint4 hi, lo;
                svbq_split8(vv, hi, lo);
                const int d = (c * a.xrows + i) * 3 + xh;
                x_hi[xw + d] = hi;
                x_lo[xw + d] = lo;
            }
        }
    };
    // ---- Q-input staging (MODE 2).  Unit U = u*256 + tid -> (chunk c, position i, 16-byte part: hi0 hi1 lo0 lo1).
    uint4 qr[QIN ? SVBQ_QUNITS : 1];
    unsigned q_off[QIN ? SVBQ_QUNITS : 1];
    int q_dst[QIN ? SVBQ_QUNITS : 1], q_c[QIN ? SVBQ_QUNITS : 1];
    bool q_ok[QIN ? SVBQ_QUNITS : 1];
    const unsigned short* xqb = nullptr;
    if (QIN) {
        const int kc_all = (a.Cin + 15) >> 4, kc_g = a.kchunks;               // (groups > 1 need Cin_g % 16 == 0: host)
        xqb = a.xq + ((size_t)b * kc_all + (size_t)g * kc_g) * a.Tin * 32;
        const int per_chunk = 4 * span;
#pragma unroll
        for (int u = 0; u < SVBQ_QUNITS; ++u) {
            const int U = u * 256 + tid;
            const int c = U / per_chunk, r = U - c * per_chunk;
            const int i = r >> 2, part = r & 3;
            const int pos = lo_pos + i;
            q_c[u] = c < a.kch ? c : (1 << 20);
            q_ok[u] = pos >= 0 && pos < a.Tin;
            q_off[u] = (c < a.kch && q_ok[u]) ? (unsigned)((c * a.Tin + pos) * 64 + part * 16) : 0u;
            q_dst[u] = ((part >> 1) ? a.x_floats16 : 0) + (c * a.xrows + i) * 3 + (part & 1);
        }
    }
    auto load_xq = [&](int kc0) {
        const int kch_here = min(a.kch, a.kchunks - kc0);
        const char* base = reinterpret_cast<const char*>(xqb + (size_t)kc0 * a.Tin * 32);
#pragma unroll
        for (int u = 0; u < (QIN ? SVBQ_QUNITS : 0); ++u)       // branch-free: idle units read the first row of the phase
            qr[u] = *reinterpret_cast<const uint4*>(base + (q_c[u] < kch_here ? q_off[u] : 0u));
    };
    auto store_xq = [&](int kc0) {
        const int kch_here = min(a.kch, a.kchunks - kc0);
#pragma unroll
        for (int u = 0; u < (QIN ? SVBQ_QUNITS : 0); ++u)
            if (q_c[u] < kch_here) x_hi[xw + q_dst[u]] = q_ok[u] ? qr[u] : make_uint4(0u, 0u, 0u, 0u);
    };
    // ---- direct-A tiles: every wave owns 32 weight rows (and all or half of the BN columns), so its MFMA A operands
    // (row l31, 16-byte half kb of each (tap, chunk) slab) are exactly one coalesced 1-KiB global read per slab -- no LDS
    // staging, no ds_write, no LDS read for the weights.  The fragments of phase s+1 are loaded into the registers of
    // phase s right after their last use (rolling prefetch), so they have a whole phase to arrive.
    uint4 wfh[SLB], wfl[SLB];
    const int wf_row = wm * 32 + l31;
    const int wf_lane16 = (wf_row < m_valid ? wf_row : 0) * 2 + kb;                       // 16-byte units inside a slab
    auto load_wf = [&](int i, int kc0, int tg0) {          // slab i = (tap i / kch, chunk i % kch) of phase (kc0, tg0)
        const int nt_here = min(a.tg, ntap - tg0);
        const int kch_here = min(a.kch, a.kchunks - kc0);
        const int t = i / a.kch, c = i - t * a.kch;
        if (t < nt_here && c < kch_here) {                 // wave-uniform
            const size_t base16 = (w_off0 + ((size_t)(tg0 + t) * tap_step * a.w_tap_slabs + (size_t)(kc0 + c)) * slab_elems) / 8;
            const uint4 th = (reinterpret_cast<const uint4*>(a.wq_hi) + base16)[wf_lane16];
            const uint4 tl = (reinterpret_cast<const uint4*>(a.wq_lo) + base16)[wf_lane16];
            wfh[i] = th;
            wfl[i] = tl;
        }
    };
    const int wbase = (wm * 32 + l31) * 3 + kb;
    int xbase[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) xbase[n] = ((wn * NT + n) * 32 + l31) * a.sx * 3 + kb;
    auto compute = [&](int kc0, int tg0) {
        const int nt_here = min(a.tg, ntap - tg0);
        const int kch_here = min(a.kch, a.kchunks - kc0);
        for (int c = 0; c < kch_here; ++c) {
            for (int t = 0; t < nt_here; ++t) {
                const int woff = (t * a.kch + c) * BM * 3;                                      // scalar
                const int xoff = (c * a.xrows + __builtin_amdgcn_readfirstlane(tap_lds[tg0 + t])) * 3;   // scalar
                const uint4 ah_u = w_hi[wbase + woff], al_u = w_lo[wbase + woff];
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&ah_u), al = *reinterpret_cast<const bf16x8*>(&al_u);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const uint4 bh_u = x_hi[xbase[n] + xoff], bl_u = x_lo[xbase[n] + xoff];
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&bh_u), bl = *reinterpret_cast<const bf16x8*>(&bl_u);
                    if constexpr (!SINGLE) {
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[n], 0, 0, 0);
                    }
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
                }
            }
        }
    };

    // direct-A compute: slabs unrolled (static fragment registers); after a slab's MFMAs its registers are refilled
    // with the same slab of the next phase
    auto compute_direct = [&](int kc0, int tg0, bool has_next, int nkc, int ntg) {
        const int nt_here = min(a.tg, ntap - tg0);
        const int kch_here = min(a.kch, a.kchunks - kc0);
#pragma unroll
        for (int i = 0; i < SLB; ++i) {
            const int t = i / a.kch, c = i - t * a.kch;
            if (t < nt_here && c < kch_here) {
                const int xoff = (c * a.xrows + p.tap_off[t0 + tg0 + t] - min_off) * 3;        // scalar
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&wfh[i]), al = *reinterpret_cast<const bf16x8*>(&wfl[i]);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const uint4 bh_u = x_hi[xbase[n] + xoff], bl_u = x_lo[xbase[n] + xoff];
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&bh_u), bl = *reinterpret_cast<const bf16x8*>(&bl_u);
                    if constexpr (!SINGLE) {
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[n], 0, 0, 0);
                    }
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
                }
            }
            if (has_next) load_wf(i, nkc, ntg);
        }
    };

    int kc0 = 0, tg0 = 0;
    if (DIRECT_A && a.fast) {
        // ---- every K phase has exactly SLB slabs (tg * kch == SLB, tg | ntap, kch | kchunks): the phase body is ONE basic
        // block.  The per-slab tap offsets are fetched before the first LDS read (a scalar load in the slab loop drains the
        // whole LDS queue: lgkmcnt is shared), the B fragments of slab i+1 are requested before the MFMAs of slab i, the three
        // products walk the NT accumulators round-robin, the weight fragments of the next phase are requested branch-free
        // right after their slab's last MFMA, and the next x tile goes into the second LDS buffer (one barrier per tile).
        const int xbuf = 2 * a.x_floats16;
        SVBQ_STAMP(6)
        int sl_t[SLB], sl_c[SLB];
#pragma unroll
        for (int i = 0; i < SLB; ++i) { sl_t[i] = i / a.kch; sl_c[i] = i - sl_t[i] * a.kch; }
        auto load_wf_full = [&](int i, int kcn, int tgn) {
            const size_t base16 = (w_off0 + ((size_t)(tgn + sl_t[i]) * tap_step * a.w_tap_slabs + (size_t)(kcn + sl_c[i])) * slab_elems) / 8;
            wfh[i] = (reinterpret_cast<const uint4*>(a.wq_hi) + base16)[wf_lane16];
            wfl[i] = (reinterpret_cast<const uint4*>(a.wq_lo) + base16)[wf_lane16];
        };
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SLB; ++i) load_wf_full(i, 0, 0);
        if (QIN) { load_xq(0); store_xq(0); }
        else if (phase_fast(0)) { load_x(0); store_x(0); } else { stage_x_slow(0); }
        __syncthreads();
        SVBQ_STAMP(7)
        int cur = 0;
        while (true) {
            int ntg = tg0 + a.tg, nkc = kc0;
            if (ntg >= ntap) { ntg = 0; nkc = kc0 + a.kch; }
            const bool has_next = nkc < a.kchunks;
            const bool new_x = has_next && ntg == 0;
            const bool fastn = new_x && (QIN || phase_fast(nkc));
            const int lkc = has_next ? nkc : kc0, ltg = has_next ? ntg : tg0;      // (last phase: harmless re-request)
            int xoff[SLB];
#pragma unroll
            for (int i = 0; i < SLB; ++i) xoff[i] = cur + (sl_c[i] * a.xrows + p.tap_off[t0 + tg0 + sl_t[i]] - min_off) * 3;
            SVBQ_STAMP(0)
            __builtin_amdgcn_s_waitcnt(0x0F70);          // this phase's weight fragments (requested a phase ago)
            if (fastn) { if (QIN) load_xq(nkc); else load_x(nkc); }
            SVBQ_STAMP(1)
            SVBQ_PRIO(a.prio == 1, 1)
            uint4 bh_u[2][NT], bl_u[2][NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) { bh_u[0][n] = x_hi[xbase[n] + xoff[0]]; bl_u[0][n] = x_lo[xbase[n] + xoff[0]]; }
            {
                // The slab loop with every non-MFMA instruction placed BY HAND in the shadow of an MFMA -- after MFMA j of slab i: one
                // LDS read of slab i+1's B fragments (j < 2 NT), or the reload of this slab's lo weight fragment for the next phase
                // (its hi fragment follows the slab's last MFMA) -- and the order pinned per MFMA.  Left alone, the scheduler sinks
                // every LDS read to just before its MFMA and waits lgkmcnt(0) in between; pinned in clumps (reads of slab i+1, the 9
                // MFMAs of slab i, the reloads -- the r02 form) a wave leaves ~200 idle matrix-pipe cycles per slab whenever its
                // SIMD partner is not in its own MFMA stage: 5.04k -> 4.35k cycles per K phase on 256->256 k5, 96.8 -> 88.8 us
                // (profiles/r03_conv_dense_slab_loop.log).  Same MFMA order per accumulator: bit-identical results.
                // (Tried on top and rejected, same log: the next x tile's 32 dword loads spread over the MFMA gaps instead of
                // issued in one burst in front of the stage -- a VMEM issue holds the in-order wave far longer than an LDS read,
                // the MFMA stage grew from 2.56k to 4.08k cycles and the kernel lost 15 %.  Ablations of this loop, same log: without
                // its LDS reads the stage is as long (they are free); without the 10 weight-fragment reloads it is 1.74k instead of
                // 2.56k cycles -- each global_load_dwordx4 holds the wave ~80 cycles -- but issuing them behind the stage instead
                // costs more in the store stage than it saves here.)
                constexpr int NM = 3 * NT;
#pragma unroll
                for (int i = 0; i < SLB; ++i) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&wfh[i]), al = *reinterpret_cast<const bf16x8*>(&wfl[i]);
                    const size_t base16 = (w_off0 + ((size_t)(ltg + sl_t[i]) * tap_step * a.w_tap_slabs + (size_t)(lkc + sl_c[i])) * slab_elems) / 8;
#pragma unroll
                    for (int j = 0; j < NM; ++j) {
                        const int n = j % NT, prod = j / NT;            // products: lo*hi, hi*lo, hi*hi
                        if (prod == 0) { if constexpr (!SINGLE) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, *reinterpret_cast<const bf16x8*>(&bh_u[i & 1][n]), acc[n], 0, 0, 0); }
                        else if (prod == 1) { if constexpr (!SINGLE) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, *reinterpret_cast<const bf16x8*>(&bl_u[i & 1][n]), acc[n], 0, 0, 0); }
                        else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, *reinterpret_cast<const bf16x8*>(&bh_u[i & 1][n]), acc[n], 0, 0, 0);
                        if (i + 1 < SLB && j < 2 * NT) {               // fillers 0 .. 2NT-1: next slab's B fragments
                            const int nn = j >> 1;
                            if (j & 1) bl_u[(i + 1) & 1][nn] = x_lo[xbase[nn] + xoff[i + 1]];
                            else bh_u[(i + 1) & 1][nn] = x_hi[xbase[nn] + xoff[i + 1]];
                        }
                        if (j == NT) wfl[i] = (reinterpret_cast<const uint4*>(a.wq_lo) + base16)[wf_lane16];   // `al` is dead after product 0
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    wfh[i] = (reinterpret_cast<const uint4*>(a.wq_hi) + base16)[wf_lane16];   // `ah` after the slab's last MFMA
                }
            }
            SVBQ_STAMP(2)
            SVBQ_PRIO(a.prio == 1, 0)
            SVBQ_PRIO(a.prio == 2, 1)
            if (!has_next) break;
            if (new_x) {
                xw = xbuf - cur;
                if (QIN) store_xq(nkc); else if (fastn) store_x(nkc); else stage_x_slow(nkc);
                SVBQ_STAMP(4)
                __syncthreads();
                SVBQ_STAMP(5)
                cur = xw;
            }
            kc0 = nkc; tg0 = ntg;
            SVBQ_NEXT_STAGE
            SVBQ_PRIO(a.prio == 2, 0)
        }
    } else if (DIRECT_A) {
        if (ntap > 0) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SLB; ++i) load_wf(i, 0, 0);
            if (QIN) { load_xq(0); store_xq(0); }
            else if (phase_fast(0)) { load_x(0); store_x(0); } else { stage_x_slow(0); }
            __syncthreads();
            while (true) {
                int ntg = tg0 + a.tg, nkc = kc0;
                if (ntg >= ntap) { ntg = 0; nkc = kc0 + a.kch; }
                const bool has_next = nkc < a.kchunks;
                const bool new_x = has_next && ntg == 0;
                const bool fastn = new_x && (QIN || phase_fast(nkc));
                SVBQ_STAMP(0)
                // all weight fragments of this phase were requested a phase ago: drain them here, so that the x loads
                // issued next are the only outstanding requests and no MFMA waits behind them (vmcnt counts in order)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (fastn) { if (QIN) load_xq(nkc); else load_x(nkc); }
                SVBQ_STAMP(1)
                compute_direct(kc0, tg0, has_next, nkc, ntg);
                SVBQ_STAMP(2)
                if (!has_next) break;
                if (new_x) {                                   // the x tile changes: all waves must be done reading it
                    __syncthreads();
                    SVBQ_STAMP(3)
                    if (QIN) store_xq(nkc); else if (fastn) store_x(nkc); else stage_x_slow(nkc);
                    SVBQ_STAMP(4)
                    __syncthreads();
                    SVBQ_STAMP(5)
                }
                kc0 = nkc; tg0 = ntg;
                SVBQ_NEXT_STAGE
            }
        }
    } else if (ntap > 0) {
        __syncthreads();
        SVBQ_STAMP(6)
        if (QIN) { load_xq(0); store_xq(0); }
        else if (phase_fast(0)) { load_x(0); store_x(0); } else { stage_x_slow(0); }
        load_w(0, 0);
        store_w(0);
        __syncthreads();
        SVBQ_STAMP(7)
        while (true) {
            int ntg = tg0 + a.tg, nkc = kc0;
            if (ntg >= ntap) { ntg = 0; nkc = kc0 + a.kch; }
            const bool has_next = nkc < a.kchunks;
            SVBQ_STAMP(0)
            if (has_next) {
                if (ntg == 0) { if (QIN) load_xq(nkc); else if (phase_fast(nkc)) load_x(nkc); }
                load_w(nkc, ntg);
            }
            SVBQ_STAMP(1)
            compute(kc0, tg0);
            SVBQ_STAMP(2)
            if (!has_next) break;
            __syncthreads();
            SVBQ_STAMP(3)
            if (ntg == 0) { if (QIN) store_xq(nkc); else if (phase_fast(nkc)) store_x(nkc); else stage_x_slow(nkc); }
            store_w(ntg);
            SVBQ_STAMP(4)
            __syncthreads();
            SVBQ_STAMP(5)
            kc0 = nkc; tg0 = ntg;
            SVBQ_NEXT_STAGE
        }
    }

    // ---- epilogue.  Everything that does not depend on the accumulator element is hoisted: per accumulator row r the channel's
    // 32-bit offset and bias, per column block n the position; the store is `uniform base + 32-bit lane offset`.  (Written
    // naively -- 64-bit index products and the activation switch per element -- this block cost ~45 VALU instructions per
    // element, a fifth of the workgroup's time.)  The feature switches are kernel arguments, i.e. scalar branches.
    // (Round 3 tried the stores as 16-byte quads through a per-wave LDS transpose -- 4 VMEM instructions per 32x32 tile and lane
    // instead of 16, gate / residual / mask as 16-byte loads: the epilogue went from 7.1-8.0k to 11-14k cycles, every shape got
    // 2-8 % slower, profiles/r03_conv_vector_epilogue_ab.log.  The epilogue is not bound by its store-instruction count.)
    const int out_base = p.phase_out_base[ph];
    SVBQ_LAST_STAGE
    SVBQ_STAMP(6)
    {
        const size_t yb_off = (size_t)b * a.Cout * a.Tout;
        float* yb = a.y + yb_off;
        int rowoff[16];
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = m_base + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
            const int co = g * a.Cout_g + ml;
            const bool ok = ml < a.Cout_g;
            rowoff[r] = ok ? co * a.Tout : -1;
            bv[r] = (a.bias && ok) ? a.bias[co] : 0.f;
        }
        const bool plain = !a.out_gate && !a.residual && !a.mask;
        const int act = a.out_act;
        const float slope = a.out_slope;
        if (plain && act != SVB_ACT_TANH) {
            const float neg = act == SVB_ACT_RELU ? 0.f : (act == SVB_ACT_LRELU ? slope : 1.f);      // v > 0 ? v : v * neg
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int ql = q0 + (wn * NT + n) * 32 + l31;
                const int pos = ql * a.out_stride + out_base;
                if (ql < nq) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[n][r] + bv[r];
                        if (rowoff[r] >= 0) yb[rowoff[r] + pos] = v > 0.f ? v : v * neg;
                    }
                }
            }
        } else {
            const float* gateb = a.out_gate ? a.out_gate + yb_off : nullptr;
            const float* resb = a.residual ? a.residual + yb_off : nullptr;
            const float* maskb = a.mask ? a.mask + (size_t)b * a.Tout : nullptr;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int ql = q0 + (wn * NT + n) * 32 + l31;
                const int pos = ql * a.out_stride + out_base;
                if (ql < nq) {
                    const float mk = maskb ? maskb[pos] : 1.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (rowoff[r] >= 0) {
                            const int oi = rowoff[r] + pos;
                            float v = svb_apply_act(acc[n][r] + bv[r], act, slope);
                            if (gateb) v *= svb_gate(gateb[oi], a.out_gate_slope);
                            if (resb) v += resb[oi];
                            yb[oi] = v * mk;
                        }
                    }
                }
            }
        }
    }
    SVBQ_STAMP(7)
}

// ------------------------------------------------------------------------------------------------------------------
// Weight pack: v [d0][d1][k] (reference layout, WeightNorm over dim-0 rows) -> bf16 hi/lo in the two operand layouts
//   qa[tap][ceil(d1/16)][d0][16]            (k-dim = d1: Conv1d forward, ConvTranspose1d data-gradient)
//   qb[tap][G][ceil(d0g/16)][d1][16]        (k-dim = d0 within its group: ConvTranspose1d forward, Conv1d data-gradient)
// Padding entries are written as zeros by the block of the row (qa) / of the group's last row (qb): buffers need no prior fill.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void svbq_pack_row(const float* v, const float* gnorm, unsigned short* qa_hi, unsigned short* qa_lo,
                                              unsigned short* qb_hi, unsigned short* qb_lo, int d0, int d1, int k, int G,
                                              int weight_norm, int row, float* red) {
    const int rowlen = d1 * k;
    const float* vr = v + (size_t)row * rowlen;
    float scale = 1.f;
    if (weight_norm) {
        float vv = 0.f;
        for (int e = threadIdx.x; e < rowlen; e += 256) vv += vr[e] * vr[e];
        vv = svb_block_sum<256>(vv, red);
        scale = gnorm[row] / sqrtf(vv);
    }
    const int kcha = (d1 + 15) / 16;
    const int d0g = d0 / G, kchb = (d0g + 15) / 16;
    const int gq = row / d0g, rl = row - gq * d0g;
    for (int e = threadIdx.x; e < rowlen; e += 256) {
        const int c1 = e / k, j = e - c1 * k;
        unsigned hi, lo;
        svbq_split(vr[e] * scale, hi, lo);
        if (qa_hi) {
            const size_t ia = (((size_t)j * kcha + (c1 >> 4)) * d0 + row) * 16 + (c1 & 15);
            qa_hi[ia] = (unsigned short)hi; qa_lo[ia] = (unsigned short)lo;
        }
        if (qb_hi) {
            const size_t ib = ((((size_t)j * G + gq) * kchb + (rl >> 4)) * d1 + c1) * 16 + (rl & 15);
            qb_hi[ib] = (unsigned short)hi; qb_lo[ib] = (unsigned short)lo;
        }
    }
    // padding (round 6: written here, so a fresh buffer needs no zero fill -- the per-call packs of the spectral-normalised convs paid a
    // memset launch each): qa's channels d1 .. 16 kcha - 1 of this row; qb's rows d0g .. 16 kchb - 1 of this row's group (its last row's block)
    const int pad_a = 16 * kcha - d1;
    if (qa_hi && pad_a)
        for (int e = threadIdx.x; e < pad_a * k; e += 256) {
            const int c1 = d1 + e / k, j = e - (e / k) * k;
            const size_t ia = (((size_t)j * kcha + (c1 >> 4)) * d0 + row) * 16 + (c1 & 15);
            qa_hi[ia] = 0; qa_lo[ia] = 0;
        }
    const int pad_b = 16 * kchb - d0g;
    if (qb_hi && pad_b && rl == d0g - 1)
        for (int e = threadIdx.x; e < pad_b * rowlen; e += 256) {
            const int r2 = d0g + e / rowlen, e2 = e - (e / rowlen) * rowlen;
            const int c1 = e2 / k, j = e2 - c1 * k;
            const size_t ib = ((((size_t)j * G + gq) * kchb + (r2 >> 4)) * d1 + c1) * 16 + (r2 & 15);
            qb_hi[ib] = 0; qb_lo[ib] = 0;
        }
}

__global__ __launch_bounds__(256) void svb_weight_pack_bf16x3_kernel(const float* v, const float* gnorm, unsigned short* qa_hi,
                                                                     unsigned short* qa_lo, unsigned short* qb_hi,
                                                                     unsigned short* qb_lo, int d0, int d1, int k, int G,
                                                                     int weight_norm) {
    __shared__ float red[8];
    svbq_pack_row(v, gnorm, qa_hi, qa_lo, qb_hi, qb_lo, d0, d1, k, G, weight_norm, blockIdx.x, red);
}

// Many weights in ONE launch (all convs of an optimizer after its step): block -> (tensor, row) through the descriptor
// table's cumulative row counts.
__global__ __launch_bounds__(256) void svb_weight_pack_bf16x3_multi_kernel(const SvbPackDesc* descs, int n) {
    __shared__ float red[8];
    int lo = 0, hi = n - 1;
    const int r = blockIdx.x;
    while (lo < hi) {                        // last descriptor with row_start <= r
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].row_start <= r) lo = mid; else hi = mid - 1;
    }
    const SvbPackDesc d = descs[lo];
    svbq_pack_row(d.v, d.g, d.qa_hi, d.qa_lo, d.qb_hi, d.qb_lo, d.d0, d.d1, d.k, d.groups, d.weight_norm, r - d.row_start, red);
}

// ==================================================================================================================
// `conv_precision: bf16` (single-product arithmetic of the forward / data-gradient convs; weight gradients keep the split).  A
// process-wide arithmetic mode like the hparam it mirrors: set once before the work is issued, not per call.
static int g_svbq_single = 0;
extern "C" void svb_conv_set_single_product(int on) { g_svbq_single = on ? 1 : 0; }
extern "C" int svb_conv_get_single_product(void) { return g_svbq_single; }

struct QCfg { int BM, BN; };
#define SVBQ_NCFG 12
static const QCfg kQCfgs[SVBQ_NCFG] = {{64, 128}, {128, 96}, {128, 128}, {64, 64}, {32, 128}, {64, 192}, {64, 256},
                                       {64, 128}, {64, 192}, {64, 256},       // 7..9: the direct-A forms of 0, 5, 6
                                       {128, 64}, {128, 32}};                 // 10, 11: narrow direct-A tiles for short sequences
                                                                              // (T = 281: more, shorter workgroups per launch)

static int q_pick(int cout_g, int nq_max, long nz) {
    long best_cost = -1;
    int best = 0;
    for (int i = 0; i < 5; ++i) {       // the wide-N tiles (5, 6) are only picked by measurement (force_cfg)
        const long mt = svb_cdiv(cout_g, kQCfgs[i].BM), qt = svb_cdiv(nq_max, kQCfgs[i].BN);
        const long area = (long)kQCfgs[i].BM * kQCfgs[i].BN;
        const long cost = ((mt * qt * nz + 255) / 256) * area;
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && area > (long)kQCfgs[best].BM * kQCfgs[best].BN)) {
            best_cost = cost;
            best = i;
        }
    }
    return best;
}

template <int WM, int WN, int NT, int SLB, int MODE>
static void q_launch_kernel(const SvbConvQArgs& a, const SvbConvPlan& p, dim3 grid, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&svb_conv1d_bf16x3_kernel<WM, WN, NT, SLB, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((svb_conv1d_bf16x3_kernel<WM, WN, NT, SLB, MODE>), grid, dim3(256), lds + g_svbq_lds_pad, stream, a, p);
}

template <int WM, int WN, int NT, int SLB>
static int q_launch(SvbConvQArgs& a, const SvbConvPlan& p, int nq_max, int span_off_max, int ntap_max, hipStream_t stream,
                    bool probe_fast = false) {
    constexpr int BM = 32 * WM, BN = 32 * WN * NT;
    const int span_max = (BN - 1) * a.sx + span_off_max + 1;
    a.xrows = span_max;
    a.fast_x = span_max <= 128 * 3 ? 1 : 0;
    a.xit = a.fast_x ? svb_cdiv(span_max, 128) : 1;
    a.kchunks = svb_cdiv(a.Cin_g, 16);
    // phase = tg taps x kch chunks, tg*kch <= SLB slabs, kch*xit <= SVBQ_XUNITS, LDS budget ~78 KB (2 blocks per CU)
    a.tg = ntap_max < 1 ? 1 : (ntap_max > SLB ? SLB : ntap_max);
    int kch = SLB / a.tg;
    // Q input: usable without an input gate, with whole 16-channel chunks per group, and when one chunk's span fits the
    // per-thread unit budget
    const bool qin = a.xq && !a.in_gate && (a.G == 1 || a.Cin_g % 16 == 0) && 4 * span_max <= SVBQ_QUNITS * 256;
    const int kch_cap = qin ? (SVBQ_QUNITS * 256) / (4 * span_max) : (a.fast_x ? SVBQ_XUNITS / a.xit : 2);
    if (kch > kch_cap) kch = kch_cap;
    if (kch > a.kchunks) kch = a.kchunks;
    if (kch < 1) kch = 1;
    // direct-A tiles: when all output phases have the same tap count and (tap group) x (chunk group) can be made exactly SLB
    // slabs with whole phases, the straight-line pipelined loop runs
    a.fast = 0;
    if (SLB <= 5 && !g_svbq_nofast && ntap_max >= 1) {
        bool same = true;
        for (int ph = 0; ph < p.n_phase; ++ph) same = same && (p.phase_start[ph + 1] - p.phase_start[ph] == ntap_max);
        for (int tg = SLB; same && tg >= 1 && !a.fast; --tg) {
            if (ntap_max % tg || SLB % tg) continue;
            const int kc = SLB / tg;
            if (kc > kch_cap || a.kchunks % kc) continue;
            if ((size_t)4 * kc * span_max * 48 + SVB_MAX_TAPS * 4 > 78 * 1024) continue;
            a.fast = 1; a.tg = tg; kch = kc;
        }
    }
    // (direct-A tiles read their weight fragments straight from global memory: no weight tile in LDS)
    if (probe_fast) return a.fast;
    auto lds_bytes = [&](int kc) {
        return (size_t)2 * ((SLB <= 5 ? 0 : a.tg * kc * BM) + (a.fast ? 2 : 1) * kc * a.xrows) * 48 + SVB_MAX_TAPS * 4;
    };
    while (!a.fast && kch > 1 && lds_bytes(kch) > 78 * 1024) --kch;
    if (lds_bytes(kch) > 150 * 1024) return SVB_ERR_UNSUPPORTED;
    a.kch = kch;
    a.w_floats16 = SLB <= 5 ? 0 : a.tg * a.kch * BM * 3;
    a.x_floats16 = a.kch * a.xrows * 3;
    dim3 grid(a.G * svb_cdiv(a.Cout_g, BM), svb_cdiv(nq_max, BN), a.B * p.n_phase);
    if (g_svbq_single && !qin) {
        if (a.in_gate) q_launch_kernel<WM, WN, NT, SLB, 4>(a, p, grid, lds_bytes(a.kch), stream);
        else q_launch_kernel<WM, WN, NT, SLB, 3>(a, p, grid, lds_bytes(a.kch), stream);
    } else if (a.in_gate && a.in_gate == a.x) q_launch_kernel<WM, WN, NT, SLB, 5>(a, p, grid, lds_bytes(a.kch), stream);
    else if (a.in_gate) q_launch_kernel<WM, WN, NT, SLB, 1>(a, p, grid, lds_bytes(a.kch), stream);
    else if (qin) q_launch_kernel<WM, WN, NT, SLB, 2>(a, p, grid, lds_bytes(a.kch), stream);
    else q_launch_kernel<WM, WN, NT, SLB, 0>(a, p, grid, lds_bytes(a.kch), stream);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// direct-A tile with 5-slab phases, or with 4-slab phases when only those divide the K extent into whole phases (the straight-
// line pipelined loop needs that: e.g. 1x1 convs, whose 12 / 16 / 24 / 64 channel chunks are multiples of 4 but not of 5)
#define SVBQ_DIRECT(WM_, WN_, NT_)                                                                                   \
    if (!q_launch<WM_, WN_, NT_, 5>(a, p, nq_max, span_off_max, ntap_max, stream, true) &&                            \
        q_launch<WM_, WN_, NT_, 4>(a, p, nq_max, span_off_max, ntap_max, stream, true))                               \
        rc = q_launch<WM_, WN_, NT_, 4>(a, p, nq_max, span_off_max, ntap_max, stream);                                \
    else                                                                                                             \
        rc = q_launch<WM_, WN_, NT_, 5>(a, p, nq_max, span_off_max, ntap_max, stream);

static int q_dispatch(SvbConvQArgs& a, const SvbConvPlan& p, hipStream_t stream) {
    int nq_max = 0, span_off_max = 0, ntap_max = 0;
    for (int ph = 0; ph < p.n_phase; ++ph) {
        if (p.phase_nq[ph] > nq_max) nq_max = p.phase_nq[ph];
        if (p.phase_span_off[ph] > span_off_max) span_off_max = p.phase_span_off[ph];
        const int nt = p.phase_start[ph + 1] - p.phase_start[ph];
        if (nt > ntap_max) ntap_max = nt;
    }
    if (nq_max <= 0) return SVB_OK;
    if ((long)a.B * p.n_phase > 65535) return SVB_ERR_UNSUPPORTED;
    if ((long)a.Cout * a.Tout > 0x7fffffffL) return SVB_ERR_UNSUPPORTED;      // the epilogue's per-clip offsets are 32-bit
    int cfg = q_pick(a.Cout_g, nq_max, (long)a.B * p.n_phase * a.G);
    if (a.force_cfg >= SVBQ_NCFG && a.force_cfg < SVBQ_NCFG + SVB_TW_NVARIANTS && !g_svbq_single) {
        // configurations 12 .. 17: the 8-wave tile-walking kernel (conv1d_tw.hip); outside its domain the heuristic tile runs
        const int rc = svb_tw_launch(a, p, a.force_cfg - SVBQ_NCFG, stream);
        if (rc != SVB_ERR_UNSUPPORTED) return rc;
    }
    if (a.force_cfg >= 0 && a.force_cfg < SVBQ_NCFG) cfg = a.force_cfg;
    // configurations 15, 16 (round 5, after the tile-walking variants so that earlier table entries keep their meaning): 32-row
    // tiles twice as wide along time for the narrow stages of the vocoder (32 / 64 output channels, 2-4 K chunks: a 32 x 128
    // workgroup runs three short phases with a barrier pair each and ~24 MFMAs per wave in between; the wide tile doubles the
    // work per staged phase).  Only ever picked by measurement.
    // configurations 17 .. 22 (round 6): the pointwise GEMM form of conv1d_pw.hip (1-tap convs; bit-identical results)
    const int pw = a.force_cfg - (SVBQ_NCFG + SVB_TW_NVARIANTS + 2);
    if (pw >= 0 && pw < SVB_PW_NVARIANTS && !g_svbq_single) {
        const int rc = svb_pw_launch(a, p, pw, stream);
        if (rc != SVB_ERR_UNSUPPORTED) return rc;
    }
    const int wide32 = a.force_cfg - (SVBQ_NCFG + SVB_TW_NVARIANTS);
    if (wide32 == 0 || wide32 == 1) {
        int rc;
        if (wide32 == 0) rc = q_launch<1, 4, 2, 8>(a, p, nq_max, span_off_max, ntap_max, stream);
        else { SVBQ_DIRECT(1, 4, 2) }
        if (rc != SVB_ERR_UNSUPPORTED) return rc;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc;
        switch (cfg) {
            case 0: rc = q_launch<2, 2, 2, 8>(a, p, nq_max, span_off_max, ntap_max, stream); break;
            case 1: SVBQ_DIRECT(4, 1, 3) break;
            case 2: SVBQ_DIRECT(4, 1, 4) break;
            case 3: rc = q_launch<2, 2, 1, 8>(a, p, nq_max, span_off_max, ntap_max, stream); break;
            case 4: rc = q_launch<1, 4, 1, 8>(a, p, nq_max, span_off_max, ntap_max, stream); break;
            case 5: rc = q_launch<2, 2, 3, 8>(a, p, nq_max, span_off_max, ntap_max, stream); break;
            case 6: rc = q_launch<2, 2, 4, 8>(a, p, nq_max, span_off_max, ntap_max, stream); break;
            case 7: SVBQ_DIRECT(2, 2, 2) break;
            case 8: SVBQ_DIRECT(2, 2, 3) break;
            case 9: SVBQ_DIRECT(2, 2, 4) break;
            case 10: SVBQ_DIRECT(4, 1, 2) break;
            default: SVBQ_DIRECT(4, 1, 1) break;
        }
        if (rc != SVB_ERR_UNSUPPORTED || cfg == 3) return rc;
        cfg = 3;        // strided convs with very wide input spans: the narrowest tile has the smallest LDS footprint
    }
    return SVB_ERR_UNSUPPORTED;
}

static void q_fill(SvbConvQArgs& a, const SvbConvEpilogue* e) {
    a.bias = e ? e->bias : nullptr;
    a.in_gate = e ? e->in_gate : nullptr;
    a.in_slope = e ? e->in_slope : 0.f;
    a.out_act = e ? e->out_act : 0;
    a.out_slope = e ? e->out_slope : 0.f;
    a.out_gate = e ? e->out_gate : nullptr;
    a.out_gate_slope = e ? e->out_gate_slope : 0.f;
    a.residual = e ? e->residual : nullptr;
    a.mask = e ? e->mask : nullptr;
    a.force_cfg = e ? e->force_cfg - 1 : -1;
    a.xq = e ? e->x_q : nullptr;
#ifdef SVB_INSTRUMENT
    a.dbg = g_svbq_dbg;
    a.dbg_block0 = g_svbq_dbg_block0;
    a.prio = g_svbq_prio;
#endif
}

extern "C" int svb_weight_pack_bf16x3(const float* v, const float* g, unsigned short* qa_hi, unsigned short* qa_lo,
                                      unsigned short* qb_hi, unsigned short* qb_lo, int d0, int d1, int k, int groups,
                                      int weight_norm, void* stream) {
    if (!v || (!qa_hi && !qb_hi) || (qa_hi && !qa_lo) || (qb_hi && !qb_lo) || d0 <= 0 || d1 <= 0 || k <= 0 || groups <= 0 ||
        d0 % groups || (weight_norm && !g))
        return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_weight_pack_bf16x3_kernel, dim3(d0), dim3(256), 0, (hipStream_t)stream, v, g, qa_hi, qa_lo, qb_hi,
                       qb_lo, d0, d1, k, groups, weight_norm);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_weight_pack_bf16x3_multi(const SvbPackDesc* descs, int n, int total_rows, void* stream) {
    if (!descs || n <= 0 || total_rows <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_weight_pack_bf16x3_multi_kernel, dim3(total_rows), dim3(256), 0, (hipStream_t)stream, descs, n);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_conv1d_forward_bf16x3(const float* x, const unsigned short* qa_hi, const unsigned short* qa_lo, float* y,
                                         int B, int Cin, int Cout, int groups, int Tin, int Tout, int k, int stride, int pad,
                                         int dil, const SvbConvEpilogue* epi, void* stream) {
    if (!x || !qa_hi || !qa_lo || !y || B <= 0 || groups <= 0 || Cin % groups || Cout % groups || k <= 0 || k > SVB_MAX_TAPS ||
        stride <= 0 || dil <= 0)
        return SVB_ERR_ARG;
    if (Tout != (Tin + 2 * pad - dil * (k - 1) - 1) / stride + 1 || Tout <= 0) return SVB_ERR_ARG;
    SvbConvQArgs a;
    SvbConvPlan p;
    memset(&p, 0, sizeof(p));
    a.x = x; a.wq_hi = qa_hi; a.wq_lo = qa_lo; a.y = y;
    q_fill(a, epi);
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.G = groups; a.Cin_g = Cin / groups; a.Cout_g = Cout / groups;
    a.Tin = Tin; a.Tout = Tout; a.sx = stride; a.out_stride = 1;
    a.w_tap_slabs = svb_cdiv(a.Cin_g, 16); a.w_g_slabs = 0; a.w_slab_rows = Cout; a.w_goff_m = a.Cout_g;
    p.n_phase = 1;
    p.phase_start[0] = 0; p.phase_start[1] = k;
    for (int j = 0; j < k; ++j) { p.tap_off[j] = j * dil - pad; p.tap_w[j] = j; }
    p.phase_nq[0] = Tout; p.phase_out_base[0] = 0; p.phase_min_off[0] = -pad; p.phase_span_off[0] = (k - 1) * dil;
    return q_dispatch(a, p, (hipStream_t)stream);
}

extern "C" int svb_conv1d_taps_bf16x3(const float* x, const unsigned short* q_hi, const unsigned short* q_lo, float* y, int B,
                                      int Cin, int Cout, int Tin, int Tout, int ntaps, const int* tap_off,
                                      const SvbConvEpilogue* epi, void* stream) {
    if (!x || !q_hi || !q_lo || !y || !tap_off || B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0 || ntaps <= 0 ||
        ntaps > SVB_MAX_TAPS)
        return SVB_ERR_ARG;
    SvbConvQArgs a;
    SvbConvPlan p;
    memset(&p, 0, sizeof(p));
    a.x = x; a.wq_hi = q_hi; a.wq_lo = q_lo; a.y = y;
    q_fill(a, epi);
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.G = 1; a.Cin_g = Cin; a.Cout_g = Cout;
    a.Tin = Tin; a.Tout = Tout; a.sx = 1; a.out_stride = 1;
    a.w_tap_slabs = svb_cdiv(Cin, 16); a.w_g_slabs = 0; a.w_slab_rows = Cout; a.w_goff_m = Cout;
    p.n_phase = 1;
    p.phase_start[0] = 0; p.phase_start[1] = ntaps;
    int mn = tap_off[0], mx = tap_off[0];
    for (int j = 0; j < ntaps; ++j) {
        p.tap_off[j] = tap_off[j]; p.tap_w[j] = j;
        if (tap_off[j] < mn) mn = tap_off[j];
        if (tap_off[j] > mx) mx = tap_off[j];
    }
    p.phase_nq[0] = Tout; p.phase_out_base[0] = 0; p.phase_min_off[0] = mn; p.phase_span_off[0] = mx - mn;
    return q_dispatch(a, p, (hipStream_t)stream);
}

extern "C" int svb_conv1d_transposed_bf16x3(const float* x, const unsigned short* qb_hi, const unsigned short* qb_lo, float* y,
                                            int B, int Cin, int Cout, int groups, int Tin, int Tout, int k, int stride, int pad,
                                            int dil, const SvbConvEpilogue* epi, void* stream) {
    if (!x || !qb_hi || !qb_lo || !y || B <= 0 || groups <= 0 || Cin % groups || Cout % groups || k <= 0 || k > SVB_MAX_TAPS ||
        stride <= 0 || stride > SVB_MAX_PHASE || dil <= 0 || Tout <= 0)
        return SVB_ERR_ARG;
    SvbConvQArgs a;
    SvbConvPlan p;
    memset(&p, 0, sizeof(p));
    a.x = x; a.wq_hi = qb_hi; a.wq_lo = qb_lo; a.y = y;
    q_fill(a, epi);
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.G = groups; a.Cin_g = Cin / groups; a.Cout_g = Cout / groups;
    a.Tin = Tin; a.Tout = Tout; a.sx = 1; a.out_stride = stride;
    const int kchb = svb_cdiv(a.Cin_g, 16);
    a.w_tap_slabs = groups * kchb; a.w_g_slabs = kchb; a.w_slab_rows = a.Cout_g; a.w_goff_m = 0;
    p.n_phase = stride;
    int nt = 0;
    for (int r = 0; r < stride; ++r) {
        p.phase_start[r] = nt;
        int umin = (pad - r) > 0 ? (pad - r + stride - 1) / stride : 0;
        const int pos0 = stride * umin + r - pad;
        p.phase_out_base[r] = pos0;
        p.phase_nq[r] = pos0 < Tout ? (Tout - 1 - pos0) / stride + 1 : 0;
        int mn = 0, mx = 0, first = 1;
        for (int j = 0; j < k; ++j) {
            if ((j * dil) % stride != r) continue;
            const int off = umin + (r - j * dil) / stride;
            p.tap_off[nt] = off; p.tap_w[nt] = j;
            if (first || off < mn) mn = off;
            if (first || off > mx) mx = off;
            first = 0;
            ++nt;
        }
        p.phase_min_off[r] = mn; p.phase_span_off[r] = mx - mn;
    }
    p.phase_start[stride] = nt;
    return q_dispatch(a, p, (hipStream_t)stream);
}

// ==================================================================================================================
// Weight gradient on the bf16 matrix cores (stride-1 convs):
//   part[split][ca][cb][j] = sum over this split's (batch, 64-position chunk)s of  A[n,ca,q] * Bt[n,cb,q + j*dil - pad]
// The reduction runs over positions, so both MFMA operands are position-major in LDS: tiles [64 channels][positions]
// of packed bf16 pairs (one dword = positions 2m, 2m+1), hi and lo.  Tap j needs the Bt operand shifted by j*dil
// positions, which is not 16-byte aligned -- so operands are read dword-wise and the shifted fragment is assembled in
// registers with v_alignbit (shift 0 or 16 bits).  For dil == 1 the taps share one window of 4+TGW/2 dwords per step.
// K order inside an MFMA is free as long as A and B agree: lane group kb takes positions 32*kb + 8*s + e.  Row pitches
// are 2*odd dwords: the 8-byte-aligned ds_read_b64 of the dil == 1 path and the dword reads of the general path are
// both bank-conflict free (each is serviced per 32-lane half, i.e. per kb).
// ==================================================================================================================
// (SvbWgradQArgs: conv1d_q.h -- shared with the direct-operand 1-tap kernel, conv1d_wgrad_pw.hip)
#define SVBQ_WG_QC 64
#define SVBQ_WG_NXIT 3

__device__ __forceinline__ unsigned svbq_funnel(unsigned hi, unsigned lo, unsigned sh) {
    return (unsigned)((((unsigned long long)hi << 32) | lo) >> sh);
}

// AT x BT: 32x32 accumulator tiles per wave along the A rows / B rows (workgroup tile 64*AT x 64*BT).  Wide tiles raise the
// MFMA work per staged element for the tap-poor gradients (1x1 convs: 12 MFMAs per wave and 64-position chunk at 1x1).
// GATED: 0 = plain operands; 1 = activation-derivative gates loaded from their own tensors; 2 = only Bt is gated and its gate tensor
// is Bt itself (the weight gradient of `conv(leaky_relu(x))`, b_gate == b: every conv of the HifiGAN generator) -- the factor comes
// from the value just loaded: a third fewer global loads and no staging registers for the gate (the general form of the 3 / 4 / 5-tap
// instantiations spills: 36 ... 200 bytes of scratch per lane at 256 VGPRs).  Same arithmetic: bit-identical.
// (Round 5, measured: the general gated 3 / 4 / 5-tap instantiations spill 36 ... 200 bytes per lane at two workgroups per CU; at ONE per
//  CU -- accumulators in AGPRs, no spills -- the vocoder step is 6 % SLOWER (118.8 / 119.3 against 112.0 ms): occupancy is worth more
//  than the spills cost.  The self-gated variants (GATED 2) need neither.)
// GATED 3 (round 6) = GATED 1 with the compile-time knowledge that only A carries a gate tensor (the discriminators' convs: the
// activation sits in the producing conv's epilogue, so dy is gated by the saved OUTPUT and x is used as it is): the staging registers
// of a second gate go away -- the 3-tap dilated instantiation no longer spills.
template <int TGW, int DIL, int GATED, int AT, int BT>
__global__ __launch_bounds__(256, 2) void svb_conv1d_wgrad_bf16x3_kernel(SvbWgradQArgs a) {
    HIP_DYNAMIC_SHARED(unsigned, wg_smem)
    unsigned* A_hi = wg_smem;
    unsigned* A_lo = A_hi + 64 * AT * a.pa;
    unsigned* B_hi = A_lo + 64 * AT * a.pa;
    unsigned* B_lo = B_hi + 64 * BT * a.pb;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int kb = lane >> 5, l31 = lane & 31;

    // XCD-aware bijective remap (round 4; block id b runs on XCD b % 8): consecutive work ids stay on one XCD, and the work id
    // walks the tiles of ONE split first -- the (a_tiles x b_tiles x tap groups) workgroups that read the same position chunks
    // in the same order share that XCD's L2 instead of each fetching its A / Bt rows through the fabric (with the identity map
    // tile t of every split ran on XCD t % 8: 16 tiles of a 256 x 256 gradient = every chunk fetched by 8 L2s, 4-16 times).
    const int orig = blockIdx.x + gridDim.x * blockIdx.y, nwg = gridDim.x * gridDim.y;
    const int qd = nwg >> 3, rd = nwg & 7, xcd = orig & 7;
    const int wgid = a.xcd_map ? (xcd < rd ? xcd * (qd + 1) : rd * (qd + 1) + (xcd - rd) * qd) + (orig >> 3) : orig;
    const int vsplit = wgid / (int)gridDim.x;
    int idx = wgid - vsplit * (int)gridDim.x;
    const int tgi = idx % a.n_tg; idx /= a.n_tg;
    const int bt = idx % a.b_tiles; idx /= a.b_tiles;
    const int at = idx % a.a_tiles;
    const int g = idx / a.a_tiles;
    const int a0 = at * 64 * AT, b0 = bt * 64 * BT;
    const int j0 = a.tg_j0[tgi];
    const int ntap = a.tg_ntap[tgi];
    const int min_off = a.tg_o0[tgi];                        // first Bt tile index, in units of the (phase) sequence
    const int ph_r = a.tg_r[tgi];
    const int span = SVBQ_WG_QC + (ntap - 1) * a.dil;
    const int nxp = (span + 1) / 2 - 32;                    // Bt pairs beyond the first 32 of a row

    f32x16 acc[AT][BT][TGW];
#pragma unroll
    for (int ia = 0; ia < AT; ++ia)
#pragma unroll
        for (int ib = 0; ib < BT; ++ib)
#pragma unroll
            for (int t = 0; t < TGW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ia][ib][t][r] = 0.f;

    constexpr int RA = 8 * AT, RB = 8 * BT;                 // staged rows per thread (row = srow + 8 * rr)
    float ar[RA][2], br[RB][2], bx[SVBQ_WG_NXIT][2];
    const int srow = tid >> 5, spair = tid & 31;            // staging role: row srow + 8*it, pair spair
    const bool do_bias = a.bias_part != nullptr && bt == 0 && tgi == 0;
    float bsum[RA];
#pragma unroll
    for (int rr = 0; rr < RA; ++rr) bsum[rr] = 0.f;
    const float* a_base = a.a + (size_t)g * a.CA_g * a.TA;
    const float* ag_base = (GATED == 1 || GATED == 3) && a.a_gate ? a.a_gate + (size_t)g * a.CA_g * a.TA : nullptr;
    const float* b_base = a.b + (size_t)g * a.CB_g * a.TB;
    const float* bg_base = GATED == 1 && a.b_gate ? a.b_gate + (size_t)g * a.CB_g * a.TB : nullptr;

    // ---- staging, branch-free: per-thread row offsets and masks are hoisted; per chunk only the (clamped) position
    // changes.  Out-of-range rows / positions read a valid element and are zeroed when written to LDS, so all loads of a
    // chunk are in flight before the first wait.
    unsigned a_roff[RA], b_roff[RB];      // byte offsets of this thread's A rows / B rows
    unsigned a_rmask = 0, b_rmask = 0;    // bit rr: row valid
#pragma unroll
    for (int rr = 0; rr < RA; ++rr) {
        const int r = srow + 8 * rr;
        const bool av = (a0 + r) < a.CA_g;
        a_roff[rr] = 4u * (unsigned)((av ? a0 + r : a0) * a.TA);
        a_rmask |= (av ? 1u : 0u) << rr;
    }
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) {
        const int r = srow + 8 * rr;
        const bool bv = (b0 + r) < a.CB_g;
        b_roff[rr] = 4u * (unsigned)((bv ? b0 + r : b0) * a.TB);
        b_rmask |= (bv ? 1u : 0u) << rr;
    }
    unsigned x_roff[SVBQ_WG_NXIT];        // extra Bt pairs (beyond the first 32 of a row): row offset, pair, validity
    int x_pr[SVBQ_WG_NXIT], x_r[SVBQ_WG_NXIT];
    bool x_rv[SVBQ_WG_NXIT], x_on[SVBQ_WG_NXIT];
#pragma unroll
    for (int e = 0; e < SVBQ_WG_NXIT; ++e) {
        const int task = tid + 256 * e;
        x_on[e] = task < 64 * BT * nxp;
        const int r = x_on[e] ? task / nxp : 0;
        x_r[e] = r;
        x_pr[e] = 32 + (x_on[e] ? task - r * nxp : 0);
        x_rv[e] = (b0 + r) < a.CB_g;
        x_roff[e] = 4u * (unsigned)((x_rv[e] ? b0 + r : b0) * a.TB);
    }
    bool a_ok0, a_ok1, b_ok0, b_ok1, xk0[SVBQ_WG_NXIT], xk1[SVBQ_WG_NXIT];     // position validity of the staged chunk

    // (Round 6 tried the two adjacent positions of a row as ONE 8-byte load at a dword-aligned address -- half the staging loads of
    //  the tap-poor gradients, 48 dword loads per 24 MFMAs at one tap: the single-tap gradients got 25-50 % SLOWER (1536 x 256:
    //  250 -> 373 us), the 5-tap ones did not move -- profiles/r06_wgrad_pair_loads.log.  Lanes of a load instruction walk ROWS here
    //  (one 4-byte piece of 64 different rows per instruction); an 8-byte piece that is not 8-byte aligned is two requests.)
    auto load_tiles = [&](int chunk) {
        const int bb = chunk / a.chunks_per_b;
        const int q0 = (chunk - bb * a.chunks_per_b) * SVBQ_WG_QC;
        const int lo = q0 + min_off;
        const float* ab = a_base + (size_t)bb * a.CA * a.TA;               // wave-uniform bases
        const float* bbp = b_base + (size_t)bb * a.CB * a.TB;
        const int qa = q0 + 2 * spair;
        const int pb0 = (lo + 2 * spair) * a.sx + ph_r, pb1 = pb0 + a.sx;          // Bt positions of the pair
        a_ok0 = qa < a.TA; a_ok1 = qa + 1 < a.TA;
        b_ok0 = pb0 >= 0 && pb0 < a.TB; b_ok1 = pb1 >= 0 && pb1 < a.TB;
        const unsigned ao0 = 4u * (unsigned)min(qa, a.TA - 1), ao1 = 4u * (unsigned)min(qa + 1, a.TA - 1);
        const unsigned bo0 = 4u * (unsigned)min(max(pb0, 0), a.TB - 1), bo1 = 4u * (unsigned)min(max(pb1, 0), a.TB - 1);
#pragma unroll
        for (int rr = 0; rr < RA; ++rr) {
            ar[rr][0] = svbq_ld(ab, a_roff[rr] + ao0);
            ar[rr][1] = svbq_ld(ab, a_roff[rr] + ao1);
        }
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            br[rr][0] = svbq_ld(bbp, b_roff[rr] + bo0);
            br[rr][1] = svbq_ld(bbp, b_roff[rr] + bo1);
        }
#pragma unroll
        for (int e = 0; e < SVBQ_WG_NXIT; ++e) {
            const int p0 = (lo + 2 * x_pr[e]) * a.sx + ph_r, p1 = p0 + a.sx;
            xk0[e] = x_on[e] && x_rv[e] && p0 >= 0 && p0 < a.TB;
            xk1[e] = x_on[e] && x_rv[e] && p1 >= 0 && p1 < a.TB;
            bx[e][0] = svbq_ld(bbp, x_roff[e] + 4u * (unsigned)min(max(p0, 0), a.TB - 1));
            bx[e][1] = svbq_ld(bbp, x_roff[e] + 4u * (unsigned)min(max(p1, 0), a.TB - 1));
        }
        if constexpr (GATED == 2) {
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                br[rr][0] *= svb_gate(br[rr][0], a.b_slope);
                br[rr][1] *= svb_gate(br[rr][1], a.b_slope);
            }
#pragma unroll
            for (int e = 0; e < SVBQ_WG_NXIT; ++e) {
                bx[e][0] *= svb_gate(bx[e][0], a.b_slope);
                bx[e][1] *= svb_gate(bx[e][1], a.b_slope);
            }
        } else if (GATED) {
            if (ag_base) {
                const float* gp = ag_base + (size_t)bb * a.CA * a.TA;
#pragma unroll
                for (int rr = 0; rr < RA; ++rr) {
                    ar[rr][0] *= svb_gate(svbq_ld(gp, a_roff[rr] + ao0), a.a_slope);
                    ar[rr][1] *= svb_gate(svbq_ld(gp, a_roff[rr] + ao1), a.a_slope);
                }
            }
            if (GATED == 1 && bg_base) {
                const float* gp = bg_base + (size_t)bb * a.CB * a.TB;
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    br[rr][0] *= svb_gate(svbq_ld(gp, b_roff[rr] + bo0), a.b_slope);
                    br[rr][1] *= svb_gate(svbq_ld(gp, b_roff[rr] + bo1), a.b_slope);
                }
#pragma unroll
                for (int e = 0; e < SVBQ_WG_NXIT; ++e) {
                    const int p0 = (lo + 2 * x_pr[e]) * a.sx + ph_r, p1 = p0 + a.sx;
                    bx[e][0] *= svb_gate(svbq_ld(gp, x_roff[e] + 4u * (unsigned)min(max(p0, 0), a.TB - 1)), a.b_slope);
                    bx[e][1] *= svb_gate(svbq_ld(gp, x_roff[e] + 4u * (unsigned)min(max(p1, 0), a.TB - 1)), a.b_slope);
                }
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int rr = 0; rr < RA; ++rr) {
            const int r = srow + 8 * rr;
            const bool av = (a_rmask >> rr) & 1u;
            const float a0v = av && a_ok0 ? ar[rr][0] : 0.f, a1v = av && a_ok1 ? ar[rr][1] : 0.f;
            unsigned hi, lo;
            bsum[rr] += a0v + a1v;
            svbq_split2(a0v, a1v, hi, lo);
            A_hi[r * a.pa + spair] = hi; A_lo[r * a.pa + spair] = lo;
        }
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int r = srow + 8 * rr;
            const bool bv = (b_rmask >> rr) & 1u;
            const float b0v = bv && b_ok0 ? br[rr][0] : 0.f, b1v = bv && b_ok1 ? br[rr][1] : 0.f;
            unsigned hi, lo;
            svbq_split2(b0v, b1v, hi, lo);
            B_hi[r * a.pb + spair] = hi; B_lo[r * a.pb + spair] = lo;
        }
#pragma unroll
        for (int e = 0; e < SVBQ_WG_NXIT; ++e) {
            if (x_on[e]) {
                unsigned hi, lo;
                svbq_split2(xk0[e] ? bx[e][0] : 0.f, xk1[e] ? bx[e][1] : 0.f, hi, lo);
                B_hi[x_r[e] * a.pb + x_pr[e]] = hi; B_lo[x_r[e] * a.pb + x_pr[e]] = lo;
            }
        }
    };
    auto mma3 = [&](const uint4& ah_u, const uint4& al_u, const uint4& bh_u, const uint4& bl_u, f32x16& c) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&ah_u), al = *reinterpret_cast<const bf16x8*>(&al_u);
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&bh_u), bl = *reinterpret_cast<const bf16x8*>(&bl_u);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    };
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 ahf[AT], alf[AT];
#pragma unroll
            for (int ia = 0; ia < AT; ++ia) {
                const unsigned* ahp = A_hi + ((wm + 2 * ia) * 32 + l31) * a.pa + 16 * kb;
                const unsigned* alp = A_lo + ((wm + 2 * ia) * 32 + l31) * a.pa + 16 * kb;
                const uint2 ah0 = *reinterpret_cast<const uint2*>(ahp + 4 * s), ah1 = *reinterpret_cast<const uint2*>(ahp + 4 * s + 2);
                const uint2 al0 = *reinterpret_cast<const uint2*>(alp + 4 * s), al1 = *reinterpret_cast<const uint2*>(alp + 4 * s + 2);
                ahf[ia] = make_uint4(ah0.x, ah0.y, ah1.x, ah1.y);
                alf[ia] = make_uint4(al0.x, al0.y, al1.x, al1.y);
            }
#pragma unroll
            for (int ib = 0; ib < BT; ++ib) {
                const unsigned* bhp = B_hi + ((wn + 2 * ib) * 32 + l31) * a.pb + 16 * kb;
                const unsigned* blp = B_lo + ((wn + 2 * ib) * 32 + l31) * a.pb + 16 * kb;
                if (DIL == 1) {
                    constexpr int NU2 = (4 + TGW / 2 + 1) / 2, NU = 2 * NU2;
                    unsigned uh[NU], ul[NU];
#pragma unroll
                    for (int d = 0; d < NU2; ++d) {
                        const uint2 th = *reinterpret_cast<const uint2*>(bhp + 4 * s + 2 * d);
                        const uint2 tl = *reinterpret_cast<const uint2*>(blp + 4 * s + 2 * d);
                        uh[2 * d] = th.x; uh[2 * d + 1] = th.y;
                        ul[2 * d] = tl.x; ul[2 * d + 1] = tl.y;
                    }
#pragma unroll
                    for (int t = 0; t < TGW; ++t) {
                        if (t < ntap) {
                            uint4 bh, bl;
                            if (t & 1) {
                                constexpr unsigned S16 = 16;
                                const int o = t >> 1;
                                bh = make_uint4(svbq_funnel(uh[o + 1], uh[o], S16), svbq_funnel(uh[o + 2], uh[o + 1], S16),
                                                svbq_funnel(uh[o + 3], uh[o + 2], S16), svbq_funnel(uh[o + 4], uh[o + 3], S16));
                                bl = make_uint4(svbq_funnel(ul[o + 1], ul[o], S16), svbq_funnel(ul[o + 2], ul[o + 1], S16),
                                                svbq_funnel(ul[o + 3], ul[o + 2], S16), svbq_funnel(ul[o + 4], ul[o + 3], S16));
                            } else {
                                const int o = t >> 1;
                                bh = make_uint4(uh[o], uh[o + 1], uh[o + 2], uh[o + 3]);
                                bl = make_uint4(ul[o], ul[o + 1], ul[o + 2], ul[o + 3]);
                            }
#pragma unroll
                            for (int ia = 0; ia < AT; ++ia) mma3(ahf[ia], alf[ia], bh, bl, acc[ia][ib][t]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < TGW; ++t) {
                        if (t < ntap) {
                            const int bp = 8 * s + t * a.dil;
                            const int dw = bp >> 1;
                            const unsigned sh = (unsigned)(bp & 1) * 16u;
                            unsigned uh[5], ul[5];
#pragma unroll
                            for (int d = 0; d < 5; ++d) { uh[d] = bhp[dw + d]; ul[d] = blp[dw + d]; }
                            const uint4 bh = make_uint4(svbq_funnel(uh[1], uh[0], sh), svbq_funnel(uh[2], uh[1], sh),
                                                        svbq_funnel(uh[3], uh[2], sh), svbq_funnel(uh[4], uh[3], sh));
                            const uint4 bl = make_uint4(svbq_funnel(ul[1], ul[0], sh), svbq_funnel(ul[2], ul[1], sh),
                                                        svbq_funnel(ul[3], ul[2], sh), svbq_funnel(ul[4], ul[3], sh));
#pragma unroll
                            for (int ia = 0; ia < AT; ++ia) mma3(ahf[ia], alf[ia], bh, bl, acc[ia][ib][t]);
                        }
                    }
                }
            }
        }
    };

    int chunk = vsplit;
    if (chunk < a.total_chunks) {
        load_tiles(chunk);
        store_tiles();
        __syncthreads();
        while (true) {
            const int next = chunk + a.nsplit;
            const bool has_next = next < a.total_chunks;
            if (has_next) load_tiles(next);
            compute();
            if (!has_next) break;
            __syncthreads();
            store_tiles();
            __syncthreads();
            chunk = next;
        }
    }

    const int cb_real = a.gp_cb ? a.gp_cb : a.CB_g;         // row pitch of the gradient: REAL input channels per group
    float* part = a.part + (size_t)vsplit * a.CA * cb_real * a.k;
#pragma unroll
    for (int ia = 0; ia < AT; ++ia)
#pragma unroll
        for (int ib = 0; ib < BT; ++ib)
#pragma unroll
            for (int t = 0; t < TGW; ++t) {
                if (t < ntap) {
                    const int bl = b0 + (wn + 2 * ib) * 32 + l31;
                    const int bgrp = a.gp_cb ? bl / a.gp_cb : 0;
                    const int bcol = a.gp_cb ? bl - bgrp * a.gp_cb : bl;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
                        const int al = a0 + (wm + 2 * ia) * 32 + row;
                        const bool diag = !a.gp_ca || al / a.gp_ca == bgrp;        // packed groups: only the diagonal blocks exist
                        if (al < a.CA_g && bl < a.CB_g && diag)
                            part[((size_t)(g * a.CA_g + al) * cb_real + bcol) * a.k + (j0 + t * a.sx)] = acc[ia][ib][t][r];
                    }
                }
            }
    if (do_bias) {        // row sums of this split's A tiles: the 32 lanes of a half-wave staged one row
#pragma unroll
        for (int rr = 0; rr < RA; ++rr) {
            float v = bsum[rr];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            const int r = srow + 8 * rr;
            if (spair == 0 && (a0 + r) < a.CA_g) a.bias_part[(size_t)vsplit * a.CA + g * a.CA_g + a0 + r] = v;
        }
    }
}

// ==================================================================================================================
// Weight gradient of GROUPED convs with few channels per group (round 3; the multi-scale discriminator's k41 convs have 16 x 8,
// 32 x 16, 32 x 32, 64 x 32 channels per group -- reference modules/hifigan/hifigan.py:259-268).  The kernel above tiles
// (output channels) x (input channels) in 32 x 32 MFMA blocks and walks the taps as shifted operands; with 8 input channels per
// group 3-12 % of its MFMA work lands on the block diagonal that exists.  Here the GEMM is turned: a 16 x 16 MFMA tile is
// 16 output channels of ONE group x 16 TAPS of one input channel (v_mfma_f32_16x16x32_bf16, K = 32 positions), so every
// product is one the gradient needs (k = 41: 41 of 48 columns).
//   workgroup = (group, 16-row block of its output channels); wave w owns input channels [w*CPW, (w+1)*CPW) of the group
//   A operand  = dy rows [16][64 positions] of the chunk, bf16 hi / lo pairs in LDS (16-byte fragment reads)
//   B operand  = x row of one input channel, stored per stride phase (position P0 + s*q + r -> phase r, index q): tap j = s*u + r
//                of output position t reads phase r at q = t + u, i.e. 8 consecutive halfwords at an arbitrary halfword offset,
//                assembled from 5 dwords with a funnel shift (as the general path above)
//   part[split][ca][cb][j] as above; bias partials = row sums of the (gated) dy rows this workgroup stages.
// Envelope (host): groups > 1, CA_g % 16 == 0, CB_g in {4, 8, 16, 32}, dil == 1, stride in {1, 2, 4}, k <= 48.
// ==================================================================================================================
struct SvbWgradG16Args {
    const float* a;
    const float* b;
    float* part;
    float* bias_part;
    const float* a_gate;
    const float* b_gate;
    float a_slope, b_slope;
    int B, CA, CB, G, CA_g, CB_g, TA, TB, k, pad, sx, sxs;       // sxs = log2(sx)
    int a_tiles, chunks_per_b, total_chunks, nsplit;
    int q_len;            // positions per phase row (even)
    int px;               // phase-row pitch in dwords
    int cb_blocks;        // workgroups per (group, 16-row block): CB_g / (4 * CPW)
};
#define SVBQ_G16_PA 36    // dy row pitch in dwords (16-byte aligned rows)

template <int CPW, int NJT>
__global__ __launch_bounds__(256, 2) void svb_conv1d_wgrad_g16_kernel(SvbWgradG16Args a) {
    constexpr int XE = 5 * CPW;                  // staged x elements per thread and chunk (bound; see the envelope)
    constexpr int TPC = 64 / CPW;                // threads per input-channel row (256 / CB_g)
    HIP_DYNAMIC_SHARED(unsigned, g16_smem)
    unsigned* A_hi = g16_smem;
    unsigned* A_lo = A_hi + 16 * SVBQ_G16_PA;
    unsigned* X_hi = A_lo + 16 * SVBQ_G16_PA;
    const int x_rows = 4 * CPW * a.sx;
    unsigned* X_lo = X_hi + x_rows * a.px;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, l15 = lane & 15;
    int bidx = blockIdx.x;
    const int cbb = bidx % a.cb_blocks; bidx /= a.cb_blocks;      // input channels are split over cb_blocks workgroups of 4*CPW
    const int at = bidx % a.a_tiles, g = bidx / a.a_tiles;
    const int ca0 = g * a.CA_g + at * 16;        // first output channel of this workgroup
    const int cbl0 = cbb * 4 * CPW;              // first input channel of this workgroup, within the group
    const int cb0 = g * a.CB_g + cbl0;

    f32x4 acc[CPW][NJT];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][jt][r] = 0.f;

    // staging roles
    const int arow = tid >> 4, at4 = (tid & 15) * 4;                 // dy: row, first of 4 consecutive positions
    const int xcb = tid / TPC, xl = tid - xcb * TPC;                  // x: input channel row, first offset
    const int span = a.sx * a.q_len;                                 // staged positions per row: P0 .. P0 + span
    const unsigned a_roff = 4u * (unsigned)((ca0 + arow) * a.TA);
    const unsigned x_roff = 4u * (unsigned)((cb0 + xcb) * a.TB);
    float ar[4], xr[XE];
    float bsum = 0.f;
    const bool do_bias = a.bias_part != nullptr && cbb == 0;
    // B-fragment constants per tap tile: phase row offset and position shift of this lane's tap
    int b_row[NJT], b_u[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const int j = 16 * jt + l15;
        b_row[jt] = (j & (a.sx - 1)) * a.px;
        b_u[jt] = j >> a.sxs;
    }

    auto load_tiles = [&](int chunk) {
        const int bb = chunk / a.chunks_per_b;
        const int t0 = (chunk - bb * a.chunks_per_b) * 64;
        const float* ab = a.a + (size_t)bb * a.CA * a.TA;
        const float* bbp = a.b + (size_t)bb * a.CB * a.TB;
        const int P0 = t0 * a.sx - a.pad;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = t0 + at4 + e;
            float v = svbq_ld(ab, a_roff + 4u * (unsigned)min(t, a.TA - 1));
            if (a.a_gate) v *= svb_gate(svbq_ld(a.a_gate + (size_t)bb * a.CA * a.TA, a_roff + 4u * (unsigned)min(t, a.TA - 1)), a.a_slope);
            ar[e] = t < a.TA ? v : 0.f;
        }
#pragma unroll
        for (int e = 0; e < XE; ++e) {
            const int i = xl + TPC * e;
            const int p = P0 + i;
            const bool ok = i < span && p >= 0 && p < a.TB;
            const unsigned off = x_roff + 4u * (unsigned)min(max(p, 0), a.TB - 1);
            float v = svbq_ld(bbp, off);
            if (a.b_gate) v *= svb_gate(svbq_ld(a.b_gate + (size_t)bb * a.CB * a.TB, off), a.b_slope);
            xr[e] = ok ? v : 0.f;
        }
    };
    auto store_tiles = [&]() {
        unsigned h0, l0, h1, l1;
        svbq_split2(ar[0], ar[1], h0, l0);
        svbq_split2(ar[2], ar[3], h1, l1);
        bsum += (ar[0] + ar[1]) + (ar[2] + ar[3]);
        const int d = arow * SVBQ_G16_PA + (at4 >> 1);
        A_hi[d] = h0; A_hi[d + 1] = h1;
        A_lo[d] = l0; A_lo[d + 1] = l1;
        unsigned short* xh = reinterpret_cast<unsigned short*>(X_hi);
        unsigned short* xl16 = reinterpret_cast<unsigned short*>(X_lo);
#pragma unroll
        for (int e = 0; e < XE; ++e) {
            const int i = xl + TPC * e;
            if (i < span) {
                unsigned hi, lo;
                svbq_split2(xr[e], 0.f, hi, lo);
                const int idx = ((xcb * a.sx + (i & (a.sx - 1))) * a.px) * 2 + (i >> a.sxs);
                xh[idx] = (unsigned short)(hi & 0xFFFFu);
                xl16[idx] = (unsigned short)(lo & 0xFFFFu);
            }
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 ahu = *reinterpret_cast<const uint4*>(A_hi + l15 * SVBQ_G16_PA + 16 * ks + 4 * kg);
            const uint4 alu = *reinterpret_cast<const uint4*>(A_lo + l15 * SVBQ_G16_PA + 16 * ks + 4 * kg);
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&ahu), al = *reinterpret_cast<const bf16x8*>(&alu);
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int rowbase = (wave * CPW + c) * a.sx * a.px;
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    const int q0 = 32 * ks + 8 * kg + b_u[jt];
                    const unsigned* hp = X_hi + rowbase + b_row[jt] + (q0 >> 1);
                    const unsigned* lp = X_lo + rowbase + b_row[jt] + (q0 >> 1);
                    const unsigned sh = (unsigned)(q0 & 1) * 16u;
                    unsigned uh[5], ul[5];
#pragma unroll
                    for (int d = 0; d < 5; ++d) { uh[d] = hp[d]; ul[d] = lp[d]; }
                    const uint4 bhu = make_uint4(svbq_funnel(uh[1], uh[0], sh), svbq_funnel(uh[2], uh[1], sh),
                                                 svbq_funnel(uh[3], uh[2], sh), svbq_funnel(uh[4], uh[3], sh));
                    const uint4 blu = make_uint4(svbq_funnel(ul[1], ul[0], sh), svbq_funnel(ul[2], ul[1], sh),
                                                 svbq_funnel(ul[3], ul[2], sh), svbq_funnel(ul[4], ul[3], sh));
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&bhu), bl = *reinterpret_cast<const bf16x8*>(&blu);
                    acc[c][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[c][jt], 0, 0, 0);
                    acc[c][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[c][jt], 0, 0, 0);
                    acc[c][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[c][jt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);       // (without it hipcc hoists every fragment read of the k-step: spills)
            }
        }
    };

    int chunk = blockIdx.y;
    if (chunk < a.total_chunks) {
        load_tiles(chunk);
        store_tiles();
        __syncthreads();
        while (true) {
            const int next = chunk + a.nsplit;
            const bool has_next = next < a.total_chunks;
            if (has_next) load_tiles(next);
            compute();
            if (!has_next) break;
            __syncthreads();
            store_tiles();
            __syncthreads();
            chunk = next;
        }
    }

    float* part = a.part + (size_t)blockIdx.y * a.CA * a.CB_g * a.k;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int cb = cbl0 + wave * CPW + c;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const int j = 16 * jt + l15;
            if (j < a.k) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    part[((size_t)(ca0 + 4 * kg + r) * a.CB_g + cb) * a.k + j] = acc[c][jt][r];
            }
        }
    }
    if (do_bias) {            // the 16 lanes that staged one dy row hold its partial sums
        float v = bsum;
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if ((tid & 15) == 0) a.bias_part[(size_t)blockIdx.y * a.CA + ca0 + arow] = v;
    }
}

static const bool g_svbq_g16_off = SVB_ENV_FLAG("SVB_WG_NO_G16");          // A/B switch
// the grouped 16-row kernel's envelope; returns the tap tiles (0 = not eligible)
static int wgq_g16_njt(int groups, int CA_g, int CB_g, int k, int sx, int dil) {
    if (g_svbq_g16_off || groups <= 1 || CA_g % 16 || dil != 1 || k > 48) return 0;
    if (!(CB_g == 4 || CB_g == 8 || CB_g == 16 || CB_g == 32)) return 0;
    if (!(sx == 1 || sx == 2 || sx == 4)) return 0;
    return svb_cdiv(k, 16);
}
static int wgq_g16_qlen(int njt, int sx) {
    int q = 64 + ((16 * njt - 1) / sx) + 2;          // positions per phase row: 64 outputs + the furthest tap + the fragment's reach
    return q + (q & 1);
}

template <int CPW, int NJT>
static void wgq_g16_launch_k(const SvbWgradG16Args& a, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&svb_conv1d_wgrad_g16_kernel<CPW, NJT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((svb_conv1d_wgrad_g16_kernel<CPW, NJT>), grid, dim3(256), lds, st, a);
}
template <int CPW>
static void wgq_g16_launch(const SvbWgradG16Args& a, int njt, dim3 grid, size_t lds, hipStream_t st) {
    if (njt == 1) wgq_g16_launch_k<CPW, 1>(a, grid, lds, st);
    else if (njt == 2) wgq_g16_launch_k<CPW, 2>(a, grid, lds, st);
    else wgq_g16_launch_k<CPW, 3>(a, grid, lds, st);
}

// Tap-group width cap of dilated weight gradients: the general shifted-operand path reads 5 + 5 dwords per tap and MFMA step, and
// with 5 accumulator sets per wave it runs at 70 TF on the period discriminators' 1024 -> 1024 layers where 3 + 2 taps reach 125
// (vocoder step 134.3 -> 130.5 ms with 3, 137.0 with 2; profiles/r03_vocoder_period_layout.log).
static const int g_svbq_wg_dil_tgw = SVB_ENV_INT("SVB_WG_DIL_TGW", 3);
static int wgq_tgw(int k) { return k <= 5 ? k : (k % 5 == 0 ? 5 : (svb_cdiv(k, 4) <= svb_cdiv(k, 5) ? 4 : 5)); }

// Tap groups of a weight gradient (see SvbWgradQArgs).  Returns the number of groups (0 = outside the envelope) and the
// taps-per-group template width in *tgw_out.
static int wgq_groups(int k, int sx, int pad, int dil, int* tgw_out, short* j0, short* ntap, short* r, short* o0) {
    if (sx > 1 && dil != 1) return 0;
    int n = 0;
    if (sx == 1) {
        int tgw = wgq_tgw(k);
        // a tap group's Bt window is 64 + (tgw - 1) * dil positions, at most 24 beyond the chunk: wide dilations (the period
        // discriminators' (5,1) convs run as dilation-p convs over [H, p] planes) get narrower groups, not the fp32 kernel
        while (tgw > 1 && ((tgw - 1) * dil + 1) / 2 > 4 * SVBQ_WG_NXIT) --tgw;
        if (dil > 1 && g_svbq_wg_dil_tgw > 0 && tgw > g_svbq_wg_dil_tgw) tgw = g_svbq_wg_dil_tgw;
        for (int j = 0; j < k; j += tgw, ++n) {
            if (j0) { j0[n] = (short)j; ntap[n] = (short)(k - j < tgw ? k - j : tgw); r[n] = 0; o0[n] = (short)(j * dil - pad); }
        }
        *tgw_out = tgw;
        return n;
    }
    const int tgw = wgq_tgw(svb_cdiv(k, sx));
    for (int ph = 0; ph < sx; ++ph) {
        int first = -1, cnt = 0;                              // taps j with (j - pad) mod sx == ph: first, first+sx, ...
        for (int j = 0; j < k; ++j)
            if ((((j - pad) % sx) + sx) % sx == ph) { if (first < 0) first = j; ++cnt; }
        for (int i = 0; i < cnt; i += tgw, ++n) {
            if (n >= SVB_MAX_TAPS) return 0;
            if (j0) {
                const int jj = first + i * sx;
                j0[n] = (short)jj; ntap[n] = (short)(cnt - i < tgw ? cnt - i : tgw); r[n] = (short)ph;
                o0[n] = (short)((jj - pad - ph) / sx);       // exact: (jj - pad - ph) is a multiple of sx
            }
        }
    }
    *tgw_out = tgw;
    return n;
}

// Wave tile multiplicity: doubling ONE side (128x64 / 64x128 workgroup tiles) pays for the single-tap gradients of the 1x1
// convs, whose 64x64 tile has only 12 MFMAs per wave and staged chunk; the staging registers of wider tiles, of gated
// operands or of more taps do not fit 256 VGPRs (measured: spills), so those keep the 64x64 tile.  A side is doubled only
// when that leaves no fully empty 64-row block.
// (Round 6, measured: the 64x128 tile for TWO taps with only the A operand gated -- the mel critic's towers; 36 bytes of scratch per
//  lane -- is 10-25 % SLOWER than the 64x64 tile on every tower shape (profiles/r06_conv_per_shape_bt2.log): not enabled.)
static void wgq_tile(int CA_g, int CB_g, int tgw, bool gated, int* at, int* bt) {
    *at = *bt = 1;
    if (tgw != 1 || gated || g_svbq_wg_narrow) return;
    const bool a2 = CA_g >= 128 && (svb_cdiv(CA_g, 128) * 128 - CA_g) < 64;
    const bool b2 = CB_g >= 128 && (svb_cdiv(CB_g, 128) * 128 - CB_g) < 64;
    if (a2) *at = 2;
    else if (b2) *bt = 2;
}

// split-K target of SHORT problems (fewer than 8 chunks per workgroup at the target above: the T = 281 layers of the step).
// Measured on the train step: 16.0 ms with 512 for all, 15.8 with 384, 15.7-15.8 with 256, 15.9 with 192, 16.15 with 128,
// 17.6 with 64; the vocoder step does not move (tools/r03_runs/r03_gpu45.sh, r03_gpu46.sh); on a third box 16.00 / 16.06 / 16.04
// without the rule against 15.84 / 15.81 / 15.82 with it (r03_gpu49.sh); the chunk threshold is flat between 6 and 16.  0 = off.
static const long g_svbq_wg_small_chunks = SVB_ENV_LONG("SVB_WG_SMALL_CHUNKS", 8);
static const long g_svbq_wg_small_blocks = SVB_ENV_LONG("SVB_WG_SMALL_BLOCKS", 256);
static const long g_svbq_wg_blocks = SVB_ENV_LONG("SVB_WG_BLOCKS", 512);   // split-K target: workgroups per launch
// Group packing factor (see SvbWgradQArgs::gp_ca): the largest power of two m dividing `groups` with m*CA_g <= 64 and m*CB_g <= 64.
static const bool g_svbq_wg_nopack = SVB_ENV_FLAG("SVB_WG_NO_GROUP_PACK");      // A/B switch
static const bool g_svbq_wg_no_xcd = SVB_ENV_FLAG("SVB_WG_NO_XCD");             // A/B switch: identity block -> work map
static const bool g_svbq_wg_no_pw = SVB_ENV_FLAG("SVB_WG_NO_PW");               // A/B switch: 1-tap gradients on the tap-group kernel
static int wgq_pack(int groups, int CA_g, int CB_g) {
    int m = 1;
    if (g_svbq_wg_nopack) return 1;
    while (groups % (2 * m) == 0 && 2 * m * CA_g <= 64 && 2 * m * CB_g <= 64) m *= 2;
    return m;
}

// 0 floats (and *nsplit = 0) when the shape is outside this kernel's envelope: the caller uses svb_conv1d_wgrad.
extern "C" size_t svb_conv1d_wgrad_bf16x3_workspace_floats(int B, int CA, int CB, int groups, int TA, int k, int sx, int pad,
                                                           int dil, int* nsplit_out) {
    if (nsplit_out) *nsplit_out = 0;
    if (B <= 0 || groups <= 0 || CA % groups || CB % groups || k <= 0 || k > SVB_MAX_TAPS || dil <= 0 || TA <= 0 || sx <= 0)
        return 0;
    int tgw = 0;
    const int n_tg = wgq_groups(k, sx, pad, dil, &tgw, nullptr, nullptr, nullptr, nullptr);
    if (n_tg <= 0) return 0;
    const long slab = (long)CA * (CB / groups) * k;
    if (wgq_g16_njt(groups, CA / groups, CB / groups, k, sx, dil)) {       // grouped 16-row kernel: one workgroup per 16 output channels
        const long tiles16 = ((long)CA / 16) * (CB / groups > 16 ? 2 : 1);
        const long chunks16 = (long)B * svb_cdiv(TA, 64);
        long cap = g_svbq_wg_blocks / tiles16;
        if (cap < 1) cap = 1;
        if (cap > chunks16) cap = chunks16;
        while (cap > 1 && cap * slab > (16L << 20)) --cap;
        const long per16 = svb_cdiv(chunks16, cap);
        const long ns16 = svb_cdiv(chunks16, per16);
        if (nsplit_out) *nsplit_out = (int)ns16;
        return (size_t)ns16 * slab;
    }
    const int gp = wgq_pack(groups, CA / groups, CB / groups);
    groups /= gp;
    const int CA_g = CA / groups, CB_g = CB / groups;
    int at = 1, bt = 1;
    if (!(gp == 1 && !g_svbq_wg_no_pw && svb_wgrad_pw_plan(CA, CB, groups, k, sx, pad, dil, TA, &at, &bt)))
        wgq_tile(CA_g, CB_g, tgw, false, &at, &bt);
    const long tiles = (long)groups * svb_cdiv(CA_g, 64 * at) * svb_cdiv(CB_g, 64 * bt) * n_tg;
    const long chunks = (long)B * svb_cdiv(TA, SVBQ_WG_QC);
    long ns_cap = g_svbq_wg_blocks / tiles;                      // one resident wave of blocks at 2 per CU
    if (ns_cap < 1) ns_cap = 1;
    // short problems: a workgroup that would see fewer than 8 chunks spends most of its life writing its 82 KB partial tile;
    // fewer, longer workgroups then (the other streams fill the CUs)
    if (g_svbq_wg_small_blocks > 0 && chunks < g_svbq_wg_small_chunks * ns_cap) {
        ns_cap = g_svbq_wg_small_blocks / tiles;
        if (ns_cap < 1) ns_cap = 1;
    }
    if (ns_cap > chunks) ns_cap = chunks;
    while (ns_cap > 1 && ns_cap * slab > (16L << 20)) --ns_cap;  // <= 64 MB of partials
    const long per = svb_cdiv(chunks, ns_cap);
    const long ns = svb_cdiv(chunks, per);
    if (nsplit_out) *nsplit_out = (int)ns;
    return (size_t)ns * slab;
}

template <int TGW, int DIL, int GATED, int AT, int BT>
static void wgq_launch_kernel(const SvbWgradQArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&svb_conv1d_wgrad_bf16x3_kernel<TGW, DIL, GATED, AT, BT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((svb_conv1d_wgrad_bf16x3_kernel<TGW, DIL, GATED, AT, BT>), grid, dim3(256), lds, st, a);
}

template <int TGW, int AT, int BT>
static void wgq_launch_t(const SvbWgradQArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    const bool gated = a.a_gate || a.b_gate;
    const bool self_b = !a.a_gate && a.b_gate && a.b_gate == a.b;
    if (a.dil == 1) {
        if (self_b) wgq_launch_kernel<TGW, 1, 2, AT, BT>(a, grid, lds, st);
        else if (gated && !a.b_gate) wgq_launch_kernel<TGW, 1, 3, AT, BT>(a, grid, lds, st);
        else if (gated) wgq_launch_kernel<TGW, 1, 1, AT, BT>(a, grid, lds, st);
        else wgq_launch_kernel<TGW, 1, 0, AT, BT>(a, grid, lds, st);
    } else {
        if (self_b) wgq_launch_kernel<TGW, 0, 2, AT, BT>(a, grid, lds, st);
        else if (gated && !a.b_gate) wgq_launch_kernel<TGW, 0, 3, AT, BT>(a, grid, lds, st);
        else if (gated) wgq_launch_kernel<TGW, 0, 1, AT, BT>(a, grid, lds, st);
        else wgq_launch_kernel<TGW, 0, 0, AT, BT>(a, grid, lds, st);
    }
}

template <int TGW>
static void wgq_launch(const SvbWgradQArgs& a, dim3 grid, size_t lds, hipStream_t st) {
    if constexpr (TGW == 1) {
        if (a.at == 2) { wgq_launch_kernel<1, 1, 0, 2, 1>(a, grid, lds, st); return; }       // (dil is irrelevant for one tap)
        if (a.bt == 2) { wgq_launch_kernel<1, 1, 0, 1, 2>(a, grid, lds, st); return; }
    }
    wgq_launch_t<TGW, 1, 1>(a, grid, lds, st);
}

extern "C" int svb_conv1d_wgrad_bf16x3(const float* a_t, const float* b_t, float* part, int B, int CA, int CB, int groups,
                                       int TA, int TB, int k, int sx, int pad, int dil, const float* a_gate, float a_slope,
                                       const float* b_gate, float b_slope, int nsplit, float* bias_part, void* stream) {
    if (!a_t || !b_t || !part || B <= 0 || groups <= 0 || CA % groups || CB % groups || k <= 0 || k > SVB_MAX_TAPS ||
        dil <= 0 || nsplit <= 0 || sx <= 0)
        return SVB_ERR_ARG;
    if (const int njt = wgq_g16_njt(groups, CA / groups, CB / groups, k, sx, dil)) {
        SvbWgradG16Args q;
        q.a = a_t; q.b = b_t; q.part = part; q.bias_part = bias_part; q.a_gate = a_gate; q.b_gate = b_gate;
        q.a_slope = a_slope; q.b_slope = b_slope;
        q.B = B; q.CA = CA; q.CB = CB; q.G = groups; q.CA_g = CA / groups; q.CB_g = CB / groups; q.TA = TA; q.TB = TB;
        q.k = k; q.pad = pad; q.sx = sx; q.sxs = sx == 1 ? 0 : (sx == 2 ? 1 : 2);
        q.a_tiles = q.CA_g / 16;
        q.chunks_per_b = svb_cdiv(TA, 64); q.total_chunks = B * q.chunks_per_b;
        if (nsplit > q.total_chunks) return SVB_ERR_ARG;
        q.nsplit = nsplit;
        q.q_len = wgq_g16_qlen(njt, sx);
        q.px = (q.q_len / 2) | 1;
        const int cb_wg = q.CB_g > 16 ? 16 : q.CB_g;            // input channels per workgroup (32 per group: two workgroups)
        q.cb_blocks = q.CB_g / cb_wg;
        const size_t lds16 = sizeof(unsigned) * ((size_t)2 * 16 * SVBQ_G16_PA + (size_t)2 * cb_wg * sx * q.px);
        dim3 grid16(groups * q.a_tiles * q.cb_blocks, nsplit);
        hipStream_t st16 = (hipStream_t)stream;
        switch (cb_wg) {
            case 4: wgq_g16_launch<1>(q, njt, grid16, lds16, st16); break;
            case 8: wgq_g16_launch<2>(q, njt, grid16, lds16, st16); break;
            default: wgq_g16_launch<4>(q, njt, grid16, lds16, st16); break;
        }
        SVB_CHECK_LAUNCH();
        return SVB_OK;
    }
    SvbWgradQArgs a;
    a.a = a_t; a.b = b_t; a.part = part; a.a_gate = a_gate; a.b_gate = b_gate; a.a_slope = a_slope; a.b_slope = b_slope;
    a.bias_part = bias_part;
    const int gp = wgq_pack(groups, CA / groups, CB / groups);
    a.gp_ca = gp > 1 ? CA / groups : 0; a.gp_cb = gp > 1 ? CB / groups : 0;
    groups /= gp;
    a.B = B; a.CA = CA; a.CB = CB; a.G = groups; a.CA_g = CA / groups; a.CB_g = CB / groups; a.TA = TA; a.TB = TB;
    a.k = k; a.off0 = -pad; a.dil = dil; a.sx = sx;
    int tgw = 0;
    a.n_tg = wgq_groups(k, sx, pad, dil, &tgw, a.tg_j0, a.tg_ntap, a.tg_r, a.tg_o0);
    if (a.n_tg <= 0) return SVB_ERR_UNSUPPORTED;
    wgq_tile(a.CA_g, a.CB_g, tgw, a_gate || b_gate, &a.at, &a.bt);
    a.a_tiles = svb_cdiv(a.CA_g, 64 * a.at); a.b_tiles = svb_cdiv(a.CB_g, 64 * a.bt);
    a.chunks_per_b = svb_cdiv(TA, SVBQ_WG_QC); a.total_chunks = B * a.chunks_per_b;
    if (nsplit > a.total_chunks) return SVB_ERR_ARG;
    a.nsplit = nsplit;
    a.xcd_map = g_svbq_wg_no_xcd ? 0 : 1;
    if (gp == 1 && !g_svbq_wg_no_pw && svb_wgrad_pw_launch(a, (hipStream_t)stream) == SVB_OK) return SVB_OK;   // 1-tap: direct operands
    a.pa = 34;
    a.pb = 32 + ((tgw - 1) * dil + 1) / 2 + 6;
    a.pb += a.pb & 1;
    if (!((a.pb >> 1) & 1)) a.pb += 2;                       // 2 * odd
    const size_t lds = (size_t)64 * (a.at * a.pa + a.bt * a.pb) * 2 * sizeof(unsigned);
    dim3 grid(groups * a.a_tiles * a.b_tiles * a.n_tg, nsplit);
    hipStream_t st = (hipStream_t)stream;
    switch (tgw) {
        case 1: wgq_launch<1>(a, grid, lds, st); break;
        case 2: wgq_launch<2>(a, grid, lds, st); break;
        case 3: wgq_launch<3>(a, grid, lds, st); break;
        case 4: wgq_launch<4>(a, grid, lds, st); break;
        default: wgq_launch<5>(a, grid, lds, st); break;
    }
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
