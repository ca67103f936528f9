// conv1d.hip -- implicit-GEMM 1-D convolution family on the CDNA4 matrix cores (gfx950).
//
// One "tap-offset" kernel covers every dense 1-D conv on the NeuralSVB hot path:
//   * Conv1d forward (dilated / strided / grouped)            reference: modules/fastspeech/fs2_vae.py:44-59 (WN),
//                                                               modules/hifigan/hifigan.py:33-50,117,261-268 ...
//   * ConvTranspose1d forward and Conv1d data-gradient         reference: vae_models.py:115-120, hifigan.py:122-125
//     (decomposed into `stride` output phases, each a stride-1 tap-offset conv -> no zero-insertion work)
// and a second kernel computes weight gradients (split-K over batch x time, deterministic two-stage reduce).
//
// Data layout in HBM: activations [B, C, T] fp32 (time contiguous -- the reference's NCT), weights pre-packed
// by svb_weight_pack() to [tap][k-channel][m-channel] so that both MFMA operands are read with unit stride.
// GEMM view: M = out channels, N = output positions of one batch row, K = (in channel, tap).
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  Lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; accumulator reg r of lane l is D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
// LDS: x tile [kc channels][span] (+ phase-split for strided reads so lanes stay bank-conflict free),
//      w tile [taps][kc][BM].  Both halves of the wave (k=0/1) read different rows -> conflict free.
// Block->tile map is XCD-aware: the M tiles that share one x tile run on the same XCD (shared L2).
#include "svb_common.h"
#include "conv1d.h"

template <int WM, int WN, int NT, int RPW, int WS_ROWS, bool GATE>
__global__ __launch_bounds__(256, 2) void svb_conv1d_mfma_kernel(SvbConvArgs a, SvbConvPlan p) {
    constexpr int BM = 32 * WM, BN = 32 * WN * NT;
    // RPW: x rows staged per wave per chunk (kc <= 4*RPW)
    constexpr int NJ = 3;                                // 64-lane column groups per x row on the fast path (span <= 192)
    constexpr int BM4 = BM / 4;
    constexpr int WUNITS = WS_ROWS * BM4;                // float4 units in the W tile
    constexpr int WU = (WUNITS + 255) / 256;
    static_assert(WM * WN == 4, "256 threads = 4 waves");
    // dynamic LDS, sized per launch: [ws: tg*kc*BM floats | xs: kc*xrow floats | tap table]
    HIP_DYNAMIC_SHARED(float, dyn_smem)
    float* ws = dyn_smem;
    float* xs = dyn_smem + a.ws_floats;
    int* tap_lds = reinterpret_cast<int*>(xs + a.xs_floats);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int kk = lane >> 5, l31 = lane & 31;

    // XCD-aware bijective remap (block id b runs on XCD b%8): consecutive work ids stay on one XCD.
    const int nwg = gridDim.x * gridDim.y;
    const int orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int qd = nwg >> 3, rd = nwg & 7, xcd = orig & 7;
    const int wgid = (xcd < rd ? xcd * (qd + 1) : rd * (qd + 1) + (xcd - rd) * qd) + (orig >> 3);
    const int mt = wgid % gridDim.x, qt = wgid / gridDim.x;

    const int m_tiles_g = gridDim.x / a.G;
    const int g = mt / m_tiles_g, mtile = mt % m_tiles_g;
    const int b = blockIdx.z / p.n_phase, ph = blockIdx.z % p.n_phase;
    const int nq = p.phase_nq[ph];
    const int q0 = qt * BN;
    if (q0 >= nq) return;  // block-uniform

    const int t0 = p.phase_start[ph], ntap = p.phase_start[ph + 1] - t0;
    const int min_off = p.phase_min_off[ph];
    const int lo = q0 * a.sx + min_off;
    const int span = (BN - 1) * a.sx + p.phase_span_off[ph] + 1;
    if (tid < ntap) {
        const int rel = p.tap_off[t0 + tid] - min_off;
        tap_lds[tid] = (a.sx == 1) ? rel : (rel % a.sx) * a.ph_len + rel / a.sx;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    const int m_base = mtile * BM;
    const int m_valid = min(BM, a.Cout_g - m_base);
    const float* xb = a.x + ((size_t)b * a.Cin + (size_t)g * a.Cin_g) * a.Tin;
    const float* gb = GATE ? a.in_gate + ((size_t)b * a.Cin + (size_t)g * a.Cin_g) * a.Tin : nullptr;
    const float* wbase = a.wp + (size_t)g * a.w_goff_k * a.w_ld + (size_t)g * a.w_goff_m + m_base;

    // ---- register staging (all global loads of a stage are issued before any LDS store) ----------------------
    // Everything that does not depend on the K-chunk is hoisted: per-lane column validity / LDS destinations of the
    // x tile, and per-unit (tap,row,col) decode + element offsets of the weight tile.
    float xr[RPW][NJ];
    float4 wr[WU];
    bool xok[NJ];
    int xsrc[NJ], xdst[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int i = lane + 64 * jj;
        const int pos = lo + i;
        xok[jj] = i < span && pos >= 0 && pos < a.Tin;
        xsrc[jj] = xok[jj] ? pos : 0;                     // out-of-range lanes read element 0 and are zeroed when staged
        xdst[jj] = (a.sx == 1) ? i : (i % a.sx) * a.ph_len + i / a.sx;
    }
    const int tap_step = ntap > 1 ? (p.tap_w[t0 + 1] - p.tap_w[t0]) : 0;   // tap slabs form an arithmetic progression
    int woff[WU], wch[WU], wtap[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int unit = u * 256 + tid;
        const int row = unit / BM4, col = (unit - row * BM4) * 4;
        const int t = row / a.kc, c = row - t * a.kc;
        const bool ok = unit < WUNITS && t < a.tg && col < m_valid;
        wtap[u] = ok ? t : 1 << 20;                       // invalid units never pass `t < nt_here`
        wch[u] = c;
        woff[u] = ok ? (p.tap_w[t0 + min(t, max(ntap - 1, 0))] * a.w_tap_stride + c * a.w_ld + col) : 0;
    }

    // Branch-free loads: every load of a stage is issued unconditionally (invalid rows / lanes read a valid dummy element)
    // and masked when the tile is written to LDS, so all of a stage's loads are in flight before the first wait.
    auto load_x = [&](int c0) {
        const int kc = min(a.kc, a.Cin_g - c0);
        const int kcp = (kc + 1) & ~1;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave + 4 * rr;
            if (r < kcp) {                                     // wave-uniform
                const size_t roff = r < kc ? (size_t)(c0 + r) * a.Tin : 0;
                const float* xrow_p = xb + roff;
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) xr[rr][jj] = xrow_p[xsrc[jj]];
                if (GATE) {
                    const float* grow_p = gb + roff;
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) xr[rr][jj] *= svb_gate(grow_p[xsrc[jj]], a.in_slope);
                }
            }
        }
    };
    auto store_x = [&](int c0) {
        const int kc = min(a.kc, a.Cin_g - c0);
        const int kcp = (kc + 1) & ~1;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave + 4 * rr;
            if (r < kcp) {
                float* xd = xs + r * a.xrow;
                const bool rv = r < kc;
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
                    if (lane + 64 * jj < span) xd[xdst[jj]] = (rv && xok[jj]) ? xr[rr][jj] : 0.f;
            }
        }
    };
    // generic (slow) x staging for wide spans / many rows: direct global -> LDS
    auto stage_x_slow = [&](int c0) {
        const int kc = min(a.kc, a.Cin_g - c0);
        const int kcp = (kc + 1) & ~1;
        for (int r = wave; r < kcp; r += 4) {
            const float* xrow_p = xb + (size_t)(c0 + r) * a.Tin;
            const float* grow_p = gb ? gb + (size_t)(c0 + r) * a.Tin : nullptr;
            for (int i = lane; i < span; i += 64) {
                const int pos = lo + i;
                float v = 0.f;
                if (r < kc && pos >= 0 && pos < a.Tin) {
                    v = xrow_p[pos];
                    if (grow_p) v *= svb_gate(grow_p[pos], a.in_slope);
                }
                const int di = (a.sx == 1) ? i : (i % a.sx) * a.ph_len + i / a.sx;
                xs[r * a.xrow + di] = v;
            }
        }
    };
    auto load_w = [&](int c0, int tg) {
        const int kc = min(a.kc, a.Cin_g - c0);
        const int nt_here = min(a.tg, ntap - tg);
        const float* wchunk = wbase + (size_t)c0 * a.w_ld + (size_t)tg * tap_step * a.w_tap_stride;
        if (a.w_vec) {
#pragma unroll
            for (int u = 0; u < WU; ++u) {                    // unconditional; units outside the stage read offset 0
                const bool ok = wtap[u] < nt_here && wch[u] < kc;
                const float4 t = *reinterpret_cast<const float4*>(wchunk + (ok ? woff[u] : 0));
                wr[u] = t;
            }
        } else {
#pragma unroll
            for (int u = 0; u < WU; ++u) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wtap[u] < nt_here && wch[u] < kc) {
                    const int col = ((u * 256 + tid) % BM4) * 4;
                    const float* src = wchunk + woff[u];
                    v.x = src[0];
                    if (col + 1 < m_valid) v.y = src[1];
                    if (col + 2 < m_valid) v.z = src[2];
                    if (col + 3 < m_valid) v.w = src[3];
                }
                wr[u] = v;
            }
        }
    };
    auto store_w = [&](int c0, int tg) {
        const int kc = min(a.kc, a.Cin_g - c0);
        const int nt_here = min(a.tg, ntap - tg);
#pragma unroll
        for (int u = 0; u < WU; ++u)
            if (wtap[u] < nt_here)                             // rows past the chunk (odd-kc padding row) must be zero
                *reinterpret_cast<float4*>(ws + (u * 256 + tid) * 4) = wch[u] < kc ? wr[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto compute = [&](int c0, int tg) {
        const int kc = min(a.kc, a.Cin_g - c0);
        const int kcp = (kc + 1) & ~1;
        const int nt_here = min(a.tg, ntap - tg);
        for (int t = 0; t < nt_here; ++t) {
            const float* wsa = ws + (t * a.kc + kk) * BM + wm * 32 + l31;
            const float* xsb = xs + kk * a.xrow + tap_lds[tg + t] + (wn * NT) * 32 + l31;
            // operands of step c2+2 are read from LDS while the MFMAs of step c2 run.  The read past the last step lands
            // in the two padding rows each tile carries (value unused), so the loop body is branch-free.
            float av = wsa[0];
            float bv[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[n] = xsb[n * 32];
            for (int c2 = 0; c2 < kcp; c2 += 2) {
                const float an = wsa[(c2 + 2) * BM];
                float bn[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) bn[n] = xsb[(c2 + 2) * a.xrow + n * 32];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[n], acc[n], 0, 0, 0);
                av = an;
#pragma unroll
                for (int n = 0; n < NT; ++n) bv[n] = bn[n];
            }
        }
    };

    // ---- software pipeline: global loads of step s+1 are in flight while step s runs on the matrix cores --------
    int c0 = 0, tg = 0;
    if (ntap > 0) {
        __syncthreads();                                   // tap_lds visible
        if (a.fast_x) { load_x(0); store_x(0); } else { stage_x_slow(0); }
        load_w(0, 0);
        store_w(0, 0);
        __syncthreads();
        while (true) {
            int ntg = tg + a.tg, nc0 = c0;
            if (ntg >= ntap) { ntg = 0; nc0 = c0 + a.kc; }
            const bool has_next = nc0 < a.Cin_g;
            if (has_next) {
                if (ntg == 0 && a.fast_x) load_x(nc0);
                load_w(nc0, ntg);
            }
            compute(c0, tg);
            if (!has_next) break;
            __syncthreads();
            if (ntg == 0) { if (a.fast_x) store_x(nc0); else stage_x_slow(nc0); }
            store_w(nc0, ntg);
            __syncthreads();
            c0 = nc0; tg = ntg;
        }
    }

    const int out_base = p.phase_out_base[ph];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ql = q0 + (wn * NT + n) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int ml = m_base + wm * 32 + row;
            if (ml < a.Cout_g && ql < nq) {
                const int co = g * a.Cout_g + ml;
                const int pos = ql * a.out_stride + out_base;
                float v = acc[n][r];
                if (a.bias) v += a.bias[co];
                v = svb_apply_act(v, a.out_act, a.out_slope);
                const size_t oi = ((size_t)b * a.Cout + co) * a.Tout + pos;
                if (a.out_gate) v *= svb_gate(a.out_gate[oi], a.out_gate_slope);
                if (a.residual) v += a.residual[oi];
                if (a.mask) v *= a.mask[(size_t)b * a.Tout + pos];
                a.y[oi] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Weight gradient: part[split][a][b][j] = sum over this split's (batch, q-chunk)s of A[b,a,q] * Bt[b,b,q*sx+off_j]
// A is the q-indexed tensor (dy for Conv1d, x for ConvTranspose1d), Bt the tap-strided one.
// ------------------------------------------------------------------------------------------------------
template <int TGW>
__global__ __launch_bounds__(256) void svb_conv1d_wgrad_kernel(SvbWgradArgs a) {
    constexpr int QCMAX = 64, AROW = QCMAX + 1;
    __shared__ float As[64 * AROW];
    __shared__ float Bs[SVB_WGRAD_BS_TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int kk = lane >> 5, l31 = lane & 31;

    int idx = blockIdx.x;
    const int tgi = idx % a.n_tg; idx /= a.n_tg;
    const int bt = idx % a.b_tiles; idx /= a.b_tiles;
    const int at = idx % a.a_tiles;
    const int g = idx / a.a_tiles;
    const int a0 = at * 64, b0 = bt * 64;
    const int j0 = tgi * TGW;
    const int ntap = min(TGW, a.k - j0);
    const int min_off = a.off0 + j0 * a.dil;
    const int span = (a.qc - 1) * a.sx + (ntap - 1) * a.dil + 1;

    f32x16 acc[TGW];
#pragma unroll
    for (int t = 0; t < TGW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    constexpr int NJB = 2;                                  // 64-lane column groups of the tap-strided tile (span <= 128)
    const bool fast = span <= 64 * NJB;
    float ar[16];
    float br[16][NJB];
    const float* a_base = a.a + (size_t)g * a.CA_g * a.TA;
    const float* ag_base = a.a_gate ? a.a_gate + (size_t)g * a.CA_g * a.TA : nullptr;
    const float* b_base = a.b + (size_t)g * a.CB_g * a.TB;
    const float* bg_base = a.b_gate ? a.b_gate + (size_t)g * a.CB_g * a.TB : nullptr;

    auto load_tiles = [&](int chunk) {
        const int bb = chunk / a.chunks_per_b;
        const int q0 = (chunk - bb * a.chunks_per_b) * a.qc;
        const int lo = q0 * a.sx + min_off;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = wave + 4 * rr;
            {
                const size_t roff = ((size_t)bb * a.CA + a0 + r) * a.TA;
                const int q = q0 + lane;
                float v = 0.f;
                if ((a0 + r) < a.CA_g && lane < a.qc && q < a.TA) {
                    v = a_base[roff + q];
                    if (ag_base) v *= svb_gate(ag_base[roff + q], a.a_slope);
                }
                ar[rr] = v;
            }
            if (fast) {
                const size_t roff = ((size_t)bb * a.CB + b0 + r) * a.TB;
#pragma unroll
                for (int jj = 0; jj < NJB; ++jj) {
                    const int i = lane + 64 * jj;
                    const int pos = lo + i;
                    float v = 0.f;
                    if ((b0 + r) < a.CB_g && i < span && pos >= 0 && pos < a.TB) {
                        v = b_base[roff + pos];
                        if (bg_base) v *= svb_gate(bg_base[roff + pos], a.b_slope);
                    }
                    br[rr][jj] = v;
                }
            }
        }
    };
    auto store_tiles = [&](int chunk) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = wave + 4 * rr;
            if (lane < a.qc) As[r * AROW + lane] = ar[rr];
            if (fast) {
#pragma unroll
                for (int jj = 0; jj < NJB; ++jj) {
                    const int i = lane + 64 * jj;
                    if (i < span) Bs[r * a.brow + i] = br[rr][jj];
                }
            }
        }
        if (!fast) {   // wide (strided) spans: direct global -> LDS
            const int bb = chunk / a.chunks_per_b;
            const int q0 = (chunk - bb * a.chunks_per_b) * a.qc;
            const int lo = q0 * a.sx + min_off;
            for (int r = wave; r < 64; r += 4) {
                const size_t roff = ((size_t)bb * a.CB + b0 + r) * a.TB;
                for (int i = lane; i < span; i += 64) {
                    const int pos = lo + i;
                    float v = 0.f;
                    if ((b0 + r) < a.CB_g && pos >= 0 && pos < a.TB) {
                        v = b_base[roff + pos];
                        if (bg_base) v *= svb_gate(bg_base[roff + pos], a.b_slope);
                    }
                    Bs[r * a.brow + i] = v;
                }
            }
        }
    };

    int chunk = blockIdx.y;
    if (chunk < a.total_chunks) {
        load_tiles(chunk);
        store_tiles(chunk);
        __syncthreads();
        while (true) {
            const int next = chunk + a.nsplit;
            const bool has_next = next < a.total_chunks;
            if (has_next) load_tiles(next);
            const float* asa = As + (wm * 32 + l31) * AROW + kk;
            const float* bsb = Bs + (wn * 32 + l31) * a.brow + kk * a.sx;
            for (int qq = 0; qq < a.qc; qq += 2) {
                const float av = asa[qq];
#pragma unroll
                for (int t = 0; t < TGW; ++t) {
                    if (t < ntap) {
                        const float bv = bsb[qq * a.sx + t * a.dil];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
            }
            if (!has_next) break;
            __syncthreads();
            store_tiles(next);
            __syncthreads();
            chunk = next;
        }
    }

    float* part = a.part + (size_t)blockIdx.y * a.CA * a.CB_g * a.k;
#pragma unroll
    for (int t = 0; t < TGW; ++t) {
        if (t < ntap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                const int al = a0 + wm * 32 + row;
                const int bl = b0 + wn * 32 + l31;
                if (al < a.CA_g && bl < a.CB_g)
                    part[((size_t)(g * a.CA_g + al) * a.CB_g + bl) * a.k + (j0 + t)] = acc[t][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Second stage of the weight gradient: sum the split partials of one weight row; for weight-normalised
// layers also apply d(g v/||v||): dg = <v,dW>/||v||, dv = g/||v|| dW - g <v,dW>/||v||^3 v
// (torch.nn.utils.weight_norm, dim=0 -- reference fs2_vae.py:42,48,58 / hifigan.py:33-50).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void svb_wgrad_reduce_row(const float* part, int nsplit, size_t split_stride, const float* v,
                                                     const float* gnorm, float* dv, float* dg, int rowlen, int weight_norm,
                                                     int accumulate, const float* bias_part, float* db, int rows, int vec,
                                                     int row, float* red) {
    const int acc_b = (accumulate & 3) != 0;      // bit 1: accumulate into db only
    accumulate &= 1;
    const size_t base = (size_t)row * rowlen;
    float dot = 0.f, vv = 0.f;
    float4 sreg[8];                                           // vec path: this thread's first 8 summed float4s of the row (8192 elements)
    // eight partials in flight per thread: a row is one float4 per thread and split, so the loop over the splits is a chain of
    // dependent HBM latencies (round 3: 28 splits two at a time = 14 round trips, 26 us per launch at 1.1 TB/s)
    auto sum_splits = [&](int e) {
        float4 s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = part + base + e;
        for (int sp = 0; sp < nsplit; sp += 8) {
            float4 p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                p[k] = sp + k < nsplit ? *reinterpret_cast<const float4*>(src + (size_t)(sp + k) * split_stride)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k].x += p[k].x; s[k].y += p[k].y; s[k].z += p[k].z; s[k].w += p[k].w; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k].x += s[k + 4].x; s[k].y += s[k + 4].y; s[k].z += s[k + 4].z; s[k].w += s[k + 4].w; }
        s[0].x += s[2].x; s[0].y += s[2].y; s[0].z += s[2].z; s[0].w += s[2].w;
        s[1].x += s[3].x; s[1].y += s[3].y; s[1].z += s[3].z; s[1].w += s[3].w;
        return make_float4(s[0].x + s[1].x, s[0].y + s[1].y, s[0].z + s[1].z, s[0].w + s[1].w);
    };
    if (vec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {                         // elements [0, 8192): kept in registers (1024 channels x 5 taps = 5120)
            const int e = threadIdx.x * 4 + j * 1024;
            sreg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < rowlen) {
                float4 sv = sum_splits(e);
                if (weight_norm) {
                    const float4 ve = *reinterpret_cast<const float4*>(v + base + e);
                    dot += ve.x * sv.x + ve.y * sv.y + ve.z * sv.z + ve.w * sv.w;
                    vv += ve.x * ve.x + ve.y * ve.y + ve.z * ve.z + ve.w * ve.w;
                    sreg[j] = sv;
                } else {
                    float4* dst = reinterpret_cast<float4*>(dv + base + e);
                    if (accumulate) {
                        const float4 o = *dst;
                        sv.x += o.x; sv.y += o.y; sv.z += o.z; sv.w += o.w;
                    }
                    *dst = sv;
                }
            }
        }
        for (int e = threadIdx.x * 4 + 8192; e < rowlen; e += 1024) {      // longer rows: dv is the scratch (no accumulate with WN)
            float4 sv = sum_splits(e);
            float4* dst = reinterpret_cast<float4*>(dv + base + e);
            if (weight_norm) {
                const float4 ve = *reinterpret_cast<const float4*>(v + base + e);
                dot += ve.x * sv.x + ve.y * sv.y + ve.z * sv.z + ve.w * sv.w;
                vv += ve.x * ve.x + ve.y * ve.y + ve.z * ve.z + ve.w * ve.w;
            } else if (accumulate) {
                const float4 o = *dst;
                sv.x += o.x; sv.y += o.y; sv.z += o.z; sv.w += o.w;
            }
            *dst = sv;
        }
    } else {
        for (int e = threadIdx.x; e < rowlen; e += 256) {
            float s = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) s += part[(size_t)sp * split_stride + base + e];
            if (weight_norm) {
                const float ve = v[base + e];
                dot += ve * s;
                vv += ve * ve;
                dv[base + e] = s;  // finalised below (no accumulate on this path: the host rejects it)
            } else {
                dv[base + e] = accumulate ? dv[base + e] + s : s;
            }
        }
    }
    if (bias_part) {      // bias-gradient partials written by the stage-1 kernel: [nsplit][rows]
        float b = 0.f;
        for (int sp = threadIdx.x; sp < nsplit; sp += 256) b += bias_part[(size_t)sp * rows + row];
        b = svb_block_sum<256>(b, red);
        if (threadIdx.x == 0) db[row] = acc_b ? db[row] + b : b;
    }
    if (!weight_norm) return;
    dot = svb_block_sum<256>(dot, red);
    vv = svb_block_sum<256>(vv, red);
    const float nrm = sqrtf(vv);
    const float gg = gnorm[row];
    const float sa = gg / nrm;
    const float sb = gg * dot / (nrm * nrm * nrm);
    if (threadIdx.x == 0) dg[row] = accumulate ? dg[row] + dot / nrm : dot / nrm;
    if (vec) {
        auto finish = [&](int e, const float4& sv) {
            float4* dst = reinterpret_cast<float4*>(dv + base + e);
            const float4 ve = *reinterpret_cast<const float4*>(v + base + e);
            float4 o = make_float4(sa * sv.x - sb * ve.x, sa * sv.y - sb * ve.y, sa * sv.z - sb * ve.z, sa * sv.w - sb * ve.w);
            if (accumulate) {
                const float4 old = *dst;
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *dst = o;
        };
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = threadIdx.x * 4 + j * 1024;
            if (e < rowlen) finish(e, sreg[j]);
        }
        for (int e = threadIdx.x * 4 + 8192; e < rowlen; e += 1024) finish(e, *reinterpret_cast<const float4*>(dv + base + e));
    } else {
        for (int e = threadIdx.x; e < rowlen; e += 256) dv[base + e] = sa * dv[base + e] - sb * v[base + e];
    }
}

__global__ __launch_bounds__(256) void svb_wgrad_reduce_kernel(const float* part, int nsplit, size_t split_stride,
                                                               const float* v, const float* gnorm, float* dv, float* dg,
                                                               int rowlen, int weight_norm, int accumulate,
                                                               const float* bias_part, float* db, int rows, int vec) {
    __shared__ float red[8];
    svb_wgrad_reduce_row(part, nsplit, split_stride, v, gnorm, dv, dg, rowlen, weight_norm, accumulate, bias_part, db, rows, vec,
                         blockIdx.x, red);
}

// Many weight gradients finished by ONE launch (the reduces of a whole backward pass, deferred to its end): the descriptors
// travel in the kernel-argument segment (no device-side table, no upload); block -> (descriptor, row) by the cumulative row
// counts.  `vec` of a descriptor is filled in by the host entry point.
#define SVB_REDUCE_BATCH 24
struct SvbReduceBatch { SvbReduceDesc d[SVB_REDUCE_BATCH]; int vec[SVB_REDUCE_BATCH]; int n; };

__global__ __launch_bounds__(256) void svb_wgrad_reduce_multi_kernel(SvbReduceBatch bt) {
    __shared__ float red[8];
    int i = 0;
    while (i + 1 < bt.n && bt.d[i + 1].row_start <= (int)blockIdx.x) ++i;
    const SvbReduceDesc& r = bt.d[i];
    svb_wgrad_reduce_row(r.part, r.nsplit, (size_t)r.rows * r.rowlen, r.v, r.g, r.dv, r.dg, r.rowlen, r.weight_norm, r.accumulate,
                         r.bias_part, r.db, r.rows, bt.vec[i], (int)blockIdx.x - r.row_start, red);
}

// ------------------------------------------------------------------------------------------------------
// Weight pack (+ WeightNorm forward): source w/v is [d0][d1][k] (reference layout, norm over dim 0 rows);
// pa[j][d1][d0] and pb[j][d0][d1] are the two operand layouts (see conv1d.h).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void svb_weight_pack_kernel(const float* v, const float* gnorm, float* pa, float* pb,
                                                              int d0, int d1, int k, int weight_norm) {
    __shared__ float red[8];
    const int row = blockIdx.x;
    const int rowlen = d1 * k;
    const float* vr = v + (size_t)row * rowlen;
    float scale = 1.f;
    if (weight_norm) {
        float vv = 0.f;
        for (int e = threadIdx.x; e < rowlen; e += 256) vv += vr[e] * vr[e];
        vv = svb_block_sum<256>(vv, red);
        scale = gnorm[row] / sqrtf(vv);
    }
    for (int e = threadIdx.x; e < rowlen; e += 256) {
        const int c1 = e / k, j = e - c1 * k;
        const float w = vr[e] * scale;
        if (pa) pa[((size_t)j * d1 + c1) * d0 + row] = w;
        if (pb) pb[((size_t)j * d0 + row) * d1 + c1] = w;
    }
}

// db[c] = sum_{b,t} dy[b,c,t] * gate'(gate[b,c,t])
__global__ __launch_bounds__(256) void svb_bias_grad_kernel(const float* dy, const float* gate, float slope, float* db,
                                                            int B, int C, int T) {
    __shared__ float red[8];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const size_t off = ((size_t)b * C + c) * T;
        for (int t = threadIdx.x; t < T; t += 256) {
            float v = dy[off + t];
            if (gate) v *= svb_gate(gate[off + t], slope);
            s += v;
        }
    }
    s = svb_block_sum<256>(s, red);
    if (threadIdx.x == 0) db[c] = s;
}

// ======================================================================================================
// Host side: plan construction + launch (C ABI, see include/svb_hip.h)
// ======================================================================================================
struct SvbTileCfg { int BM, BN; };
static const SvbTileCfg kCfgs[5] = {{64, 128}, {128, 96}, {128, 128}, {64, 64}, {32, 128}};

// Tile choice = smallest estimated makespan: blocks are dealt round-robin to the 256 CUs, so the kernel takes
// ceil(blocks / 256) "rounds" of one tile's work each (padding waste and CU under-fill both show up here);
// ties go to the larger tile (fewer weight re-reads).  `nz` = batch * phases (grid.z), `groups` multiplies grid.x.
static int pick_cfg(int cout_g, int nq_max, long nz) {
    long best_cost = -1;
    int best = 0;
    for (int i = 0; i < 5; ++i) {
        const long mt = svb_cdiv(cout_g, kCfgs[i].BM), qt = svb_cdiv(nq_max, kCfgs[i].BN);
        const long area = (long)kCfgs[i].BM * kCfgs[i].BN;
        const long rounds = (mt * qt * nz + 255) / 256;
        const long cost = rounds * area;
        if (best_cost < 0 || cost < best_cost ||
            (cost == best_cost && area > (long)kCfgs[best].BM * kCfgs[best].BN)) {
            best_cost = cost;
            best = i;
        }
    }
    return best;
}

template <int WM, int WN, int NT, int RPW, int WS_ROWS, bool GATE>
static void launch_one_g(const SvbConvArgs& a, const SvbConvPlan& p, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    static bool attr_set = false;   // allow > 64 KiB of dynamic LDS (wide strided tiles); set once per instantiation
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&svb_conv1d_mfma_kernel<WM, WN, NT, RPW, WS_ROWS, GATE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((svb_conv1d_mfma_kernel<WM, WN, NT, RPW, WS_ROWS, GATE>), grid, dim3(256), lds_bytes, stream, a, p);
}

template <int WM, int WN, int NT, int RPW, int WS_ROWS>
static int launch_one(SvbConvArgs& a, const SvbConvPlan& p, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if (a.in_gate) launch_one_g<WM, WN, NT, RPW, WS_ROWS, true>(a, p, grid, lds_bytes, stream);
    else launch_one_g<WM, WN, NT, RPW, WS_ROWS, false>(a, p, grid, lds_bytes, stream);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

template <int WM, int WN, int NT>
static int launch_cfg(SvbConvArgs& a, const SvbConvPlan& p, int nq_max, int span_off_max, int ntap_max, hipStream_t stream) {
    constexpr int BM = 32 * WM, BN = 32 * WN * NT, WS_ROWS = 80, XS_MAX = 8192;
    const int span_max = (BN - 1) * a.sx + span_off_max + 1;
    a.ph_len = svb_cdiv(span_max, a.sx);
    a.xrow = a.ph_len * a.sx;
    a.tg = ntap_max < 5 ? (ntap_max > 0 ? ntap_max : 1) : 5;       // taps per weight stage
    int kc = XS_MAX / a.xrow;                                        // channels per K-chunk
    if (kc > WS_ROWS / a.tg) kc = WS_ROWS / a.tg;
    constexpr int RPW_BIG = (BM >= 128) ? 8 : 16;                   // register budget: 128-row tiles stage fewer x rows
    if (kc > 4 * RPW_BIG) kc = 4 * RPW_BIG;
    if (kc > ((a.Cin_g + 1) & ~1)) kc = (a.Cin_g + 1) & ~1;
    kc &= ~1;
    if (kc < 2) return SVB_ERR_UNSUPPORTED;
    a.kc = kc;
    a.fast_x = span_max <= 192 ? 1 : 0;
    a.w_vec = (a.w_ld % 4 == 0 && a.w_tap_stride % 4 == 0 && a.w_goff_m % 4 == 0 &&
               ((size_t)a.w_goff_k * a.w_ld) % 4 == 0 && ((uintptr_t)a.wp % 16) == 0) ? 1 : 0;
    a.ws_floats = (a.tg * a.kc + 2) * BM;                            // +2 padding rows (operand prefetch over-read)
    a.xs_floats = ((a.kc + 2) * a.xrow + BN + 3) & ~3;               // multiple of 4 floats keeps the carve 16-byte aligned
    const size_t lds_bytes = (size_t)(a.ws_floats + a.xs_floats + SVB_MAX_TAPS) * 4;
    dim3 grid(a.G * svb_cdiv(a.Cout_g, BM), svb_cdiv(nq_max, BN), a.B * p.n_phase);
    if (kc <= 16) return launch_one<WM, WN, NT, 4, WS_ROWS>(a, p, grid, lds_bytes, stream);
    return launch_one<WM, WN, NT, RPW_BIG, WS_ROWS>(a, p, grid, lds_bytes, stream);
}

static int launch_conv(SvbConvArgs& a, const SvbConvPlan& p, hipStream_t stream) {
    int nq_max = 0, span_off_max = 0, ntap_max = 0;
    for (int ph = 0; ph < p.n_phase; ++ph) {
        if (p.phase_nq[ph] > nq_max) nq_max = p.phase_nq[ph];
        if (p.phase_span_off[ph] > span_off_max) span_off_max = p.phase_span_off[ph];
        const int nt = p.phase_start[ph + 1] - p.phase_start[ph];
        if (nt > ntap_max) ntap_max = nt;
    }
    if (nq_max <= 0) return SVB_OK;
    if ((long)a.B * p.n_phase > 65535) return SVB_ERR_UNSUPPORTED;
    int cfg = pick_cfg(a.Cout_g, nq_max, (long)a.B * p.n_phase * a.G);
    if (a.force_cfg >= 0 && a.force_cfg < 5) cfg = a.force_cfg;
    switch (cfg) {
        case 0: return launch_cfg<2, 2, 2>(a, p, nq_max, span_off_max, ntap_max, stream);
        case 1: return launch_cfg<4, 1, 3>(a, p, nq_max, span_off_max, ntap_max, stream);
        case 2: return launch_cfg<4, 1, 4>(a, p, nq_max, span_off_max, ntap_max, stream);
        case 3: return launch_cfg<2, 2, 1>(a, p, nq_max, span_off_max, ntap_max, stream);
        default: return launch_cfg<1, 4, 1>(a, p, nq_max, span_off_max, ntap_max, stream);
    }
}

extern "C" int svb_conv1d_pick_cfg(int cout_g, int nq_max, int nz) { return pick_cfg(cout_g, nq_max, nz); }

static void fill_epilogue(SvbConvArgs& a, const SvbConvEpilogue* e) {
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
}

extern "C" int svb_conv1d_forward(const float* x, const float* wp, float* y, int B, int Cin, int Cout, int groups,
                                  int Tin, int Tout, int k, int stride, int pad, int dil,
                                  const SvbConvEpilogue* epi, void* stream) {
    if (!x || !wp || !y || B <= 0 || groups <= 0 || Cin % groups || Cout % groups || k <= 0 || k > SVB_MAX_TAPS ||
        stride <= 0 || dil <= 0)
        return SVB_ERR_ARG;
    if (Tout != (Tin + 2 * pad - dil * (k - 1) - 1) / stride + 1 || Tout <= 0) return SVB_ERR_ARG;
    SvbConvArgs a;
    SvbConvPlan p;
    memset(&p, 0, sizeof(p));
    a.x = x; a.wp = wp; a.y = y;
    fill_epilogue(a, epi);
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.G = groups; a.Cin_g = Cin / groups; a.Cout_g = Cout / groups;
    a.Tin = Tin; a.Tout = Tout; a.sx = stride; a.out_stride = 1;
    a.w_tap_stride = a.Cin_g * Cout; a.w_ld = Cout; a.w_goff_k = 0; a.w_goff_m = a.Cout_g;
    p.n_phase = 1;
    p.phase_start[0] = 0; p.phase_start[1] = k;
    for (int j = 0; j < k; ++j) { p.tap_off[j] = j * dil - pad; p.tap_w[j] = j; }
    p.phase_nq[0] = Tout; p.phase_out_base[0] = 0; p.phase_min_off[0] = -pad; p.phase_span_off[0] = (k - 1) * dil;
    return launch_conv(a, p, (hipStream_t)stream);
}

// Stride-1 conv with an arbitrary tap table: y[b,co,q] = sum_{ci,t} wp[t][ci][co] x[b,ci,q + tap_off[t]] (zero outside [0,Tin)).
extern "C" int svb_conv1d_taps(const float* x, const float* wp, float* y, int B, int Cin, int Cout, int Tin, int Tout,
                               int ntaps, const int* tap_off, const SvbConvEpilogue* epi, void* stream) {
    if (!x || !wp || !y || !tap_off || B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0 || ntaps <= 0 ||
        ntaps > SVB_MAX_TAPS)
        return SVB_ERR_ARG;
    SvbConvArgs a;
    SvbConvPlan p;
    memset(&p, 0, sizeof(p));
    a.x = x; a.wp = wp; a.y = y;
    fill_epilogue(a, epi);
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.G = 1; a.Cin_g = Cin; a.Cout_g = Cout;
    a.Tin = Tin; a.Tout = Tout; a.sx = 1; a.out_stride = 1;
    a.w_tap_stride = Cin * Cout; a.w_ld = Cout; a.w_goff_k = 0; a.w_goff_m = Cout;
    p.n_phase = 1;
    p.phase_start[0] = 0; p.phase_start[1] = ntaps;
    int mn = tap_off[0], mx = tap_off[0];
    for (int j = 0; j < ntaps; ++j) {
        p.tap_off[j] = tap_off[j]; p.tap_w[j] = j;
        if (tap_off[j] < mn) mn = tap_off[j];
        if (tap_off[j] > mx) mx = tap_off[j];
    }
    p.phase_nq[0] = Tout; p.phase_out_base[0] = 0; p.phase_min_off[0] = mn; p.phase_span_off[0] = mx - mn;
    return launch_conv(a, p, (hipStream_t)stream);
}

// y[b,co,pos] = sum_{ci,j : pos = t*stride - pad + j*dil} wp[j][ci][co] x[b,ci,t]   (gather form, no zero insertion)
extern "C" int svb_conv1d_transposed(const float* x, const float* wp, float* y, int B, int Cin, int Cout, int groups,
                                     int Tin, int Tout, int k, int stride, int pad, int dil,
                                     const SvbConvEpilogue* epi, void* stream) {
    if (!x || !wp || !y || B <= 0 || groups <= 0 || Cin % groups || Cout % groups || k <= 0 || k > SVB_MAX_TAPS ||
        stride <= 0 || stride > SVB_MAX_PHASE || dil <= 0 || Tout <= 0)
        return SVB_ERR_ARG;
    SvbConvArgs a;
    SvbConvPlan p;
    memset(&p, 0, sizeof(p));
    a.x = x; a.wp = wp; a.y = y;
    fill_epilogue(a, epi);
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.G = groups; a.Cin_g = Cin / groups; a.Cout_g = Cout / groups;
    a.Tin = Tin; a.Tout = Tout; a.sx = 1; a.out_stride = stride;
    // packed as [j][Cin (global)][Cout_g]
    a.w_tap_stride = Cin * a.Cout_g; a.w_ld = a.Cout_g; a.w_goff_k = a.Cin_g; a.w_goff_m = 0;
    p.n_phase = stride;
    int nt = 0;
    for (int r = 0; r < stride; ++r) {
        p.phase_start[r] = nt;
        // output positions pos = stride*u + r - pad >= 0  ->  u >= ceil((pad - r)/stride)
        int umin = (pad - r) > 0 ? (pad - r + stride - 1) / stride : 0;
        const int pos0 = stride * umin + r - pad;
        p.phase_out_base[r] = pos0;
        p.phase_nq[r] = pos0 < Tout ? (Tout - 1 - pos0) / stride + 1 : 0;
        int mn = 0, mx = 0, first = 1;
        for (int j = 0; j < k; ++j) {
            if ((j * dil) % stride != r) continue;
            const int off = umin + (r - j * dil) / stride;  // exact division
            p.tap_off[nt] = off; p.tap_w[nt] = j;
            if (first || off < mn) mn = off;
            if (first || off > mx) mx = off;
            first = 0;
            ++nt;
        }
        p.phase_min_off[r] = mn; p.phase_span_off[r] = mx - mn;
    }
    p.phase_start[stride] = nt;
    return launch_conv(a, p, (hipStream_t)stream);
}

extern "C" size_t svb_conv1d_wgrad_workspace_floats(int B, int CA, int CB, int groups, int TA, int k, int sx,
                                                    int* nsplit_out) {
    const int CA_g = CA / groups, CB_g = CB / groups;
    int qc = 64 / sx; if (qc < 2) qc = 2; qc &= ~1;
    const int tgw = k <= 5 ? k : (k % 5 == 0 ? 5 : (svb_cdiv(k, 4) <= svb_cdiv(k, 5) ? 4 : 5));
    const int n_tg = svb_cdiv(k, tgw);
    const long tiles = (long)groups * svb_cdiv(CA_g, 64) * svb_cdiv(CB_g, 64) * n_tg;
    const long chunks = (long)B * svb_cdiv(TA, qc);
    long ns = 768 / tiles; if (ns < 1) ns = 1; if (ns > chunks) ns = chunks; if (ns > 24) ns = 24;
    if (nsplit_out) *nsplit_out = (int)ns;
    return (size_t)ns * CA * CB_g * k;
}

extern "C" int svb_conv1d_wgrad(const float* a_t, const float* b_t, float* part, int B, int CA, int CB, int groups,
                                int TA, int TB, int k, int sx, int pad, int dil, const float* a_gate, float a_slope,
                                const float* b_gate, float b_slope, int nsplit, void* stream) {
    if (!a_t || !b_t || !part || B <= 0 || groups <= 0 || CA % groups || CB % groups || k <= 0 || k > SVB_MAX_TAPS ||
        sx <= 0 || dil <= 0 || nsplit <= 0)
        return SVB_ERR_ARG;
    SvbWgradArgs a;
    a.a = a_t; a.b = b_t; a.part = part; a.a_gate = a_gate; a.b_gate = b_gate; a.a_slope = a_slope; a.b_slope = b_slope;
    a.B = B; a.CA = CA; a.CB = CB; a.G = groups; a.CA_g = CA / groups; a.CB_g = CB / groups; a.TA = TA; a.TB = TB;
    a.k = k; a.sx = sx; a.off0 = -pad; a.dil = dil;
    int qc = 64 / sx; if (qc < 2) qc = 2; qc &= ~1;
    a.qc = qc;
    const int tgw = k <= 5 ? k : (k % 5 == 0 ? 5 : (svb_cdiv(k, 4) <= svb_cdiv(k, 5) ? 4 : 5));
    a.n_tg = svb_cdiv(k, tgw);
    a.a_tiles = svb_cdiv(a.CA_g, 64); a.b_tiles = svb_cdiv(a.CB_g, 64);
    a.chunks_per_b = svb_cdiv(TA, qc); a.total_chunks = B * a.chunks_per_b;
    if (nsplit > a.total_chunks) return SVB_ERR_ARG;
    a.nsplit = nsplit;
    a.brow = ((qc - 1) * sx + (tgw - 1) * dil + 1) | 1;
    if ((long)a.brow * 64 > SVB_WGRAD_BS_TOTAL) return SVB_ERR_UNSUPPORTED;
    dim3 grid(groups * a.a_tiles * a.b_tiles * a.n_tg, nsplit);
    hipStream_t st = (hipStream_t)stream;
    switch (tgw) {
        case 1: hipLaunchKernelGGL((svb_conv1d_wgrad_kernel<1>), grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((svb_conv1d_wgrad_kernel<2>), grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((svb_conv1d_wgrad_kernel<3>), grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((svb_conv1d_wgrad_kernel<4>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((svb_conv1d_wgrad_kernel<5>), grid, dim3(256), 0, st, a); break;
    }
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_wgrad_reduce(const float* part, int nsplit, const float* v, const float* g, float* dv, float* dg,
                                int rows, int rowlen, int weight_norm, int accumulate, const float* bias_part, float* db,
                                void* stream) {
    if (!part || !dv || rows <= 0 || rowlen <= 0 || nsplit <= 0) return SVB_ERR_ARG;
    if (weight_norm && (!v || !g || !dg)) return SVB_ERR_ARG;
    if (bias_part && !db) return SVB_ERR_ARG;
    // 16-byte loads when every row of every operand is 16-byte aligned
    const int vec = (rowlen & 3) == 0 && (((uintptr_t)part | (uintptr_t)dv | (uintptr_t)v) & 15) == 0;
    // accumulating into dv with WeightNorm needs the summed row in registers (dv is not free to be scratch)
    if (weight_norm && accumulate && !(vec && rowlen <= 8192)) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_wgrad_reduce_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, part, nsplit,
                       (size_t)rows * rowlen, v, g, dv, dg, rowlen, weight_norm, accumulate, bias_part, db, rows, vec);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_wgrad_reduce_multi(const SvbReduceDesc* descs, int n, void* stream) {
    if (!descs || n <= 0) return SVB_ERR_ARG;
    for (int i0 = 0; i0 < n; i0 += SVB_REDUCE_BATCH) {
        SvbReduceBatch bt;
        bt.n = n - i0 < SVB_REDUCE_BATCH ? n - i0 : SVB_REDUCE_BATCH;
        int rows = 0;
        for (int i = 0; i < bt.n; ++i) {
            SvbReduceDesc r = descs[i0 + i];
            if (!r.part || !r.dv || r.rows <= 0 || r.rowlen <= 0 || r.nsplit <= 0) return SVB_ERR_ARG;
            if (r.weight_norm && (!r.v || !r.g || !r.dg)) return SVB_ERR_ARG;
            if (r.bias_part && !r.db) return SVB_ERR_ARG;
            const int vec = (r.rowlen & 3) == 0 && (((uintptr_t)r.part | (uintptr_t)r.dv | (uintptr_t)r.v) & 15) == 0;
            if (r.weight_norm && (r.accumulate & 1) && !(vec && r.rowlen <= 8192)) return SVB_ERR_UNSUPPORTED;
            r.row_start = rows;
            rows += r.rows;
            bt.d[i] = r;
            bt.vec[i] = vec;
        }
        hipLaunchKernelGGL(svb_wgrad_reduce_multi_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, bt);
    }
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_weight_pack(const float* v, const float* g, float* pa, float* pb, int d0, int d1, int k,
                               int weight_norm, void* stream) {
    if (!v || (!pa && !pb) || d0 <= 0 || d1 <= 0 || k <= 0 || (weight_norm && !g)) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_weight_pack_kernel, dim3(d0), dim3(256), 0, (hipStream_t)stream, v, g, pa, pb, d0, d1, k,
                       weight_norm);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_bias_grad(const float* dy, const float* gate, float slope, float* db, int B, int C, int T,
                             void* stream) {
    if (!dy || !db || B <= 0 || C <= 0 || T <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_bias_grad_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, gate, slope, db, B, C, T);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
