// attn.hip -- the glue between the three matrix products of the conformer's relative-position self-attention
// (reference modules/commons/espnet_transformer_attn.py:125-186, RelPositionMultiHeadedAttention, legacy rel_shift):
//
//   scores[i][j] = (ac[i][j] + shift(bd)[i][j]) / sqrt(d_k);  masked keys -> -FLT_MAX;  attn = softmax_j(scores);  masked -> 0
//
// where shift() is the "pad one zero column, view [T+1,T], drop the first row" trick (:125-148), i.e.
//   shift(bd)[i][j] = bd[i][T-1-i+j]        for j <= i
//                   = 0                      for j == i+1
//                   = bd[i+1][j-i-2]         for j >  i+1
// torch runs this as pad / view / slice-copy / add / div / masked_fill / softmax / masked_fill: eight passes over the
// [B,h,T,T] tensors.  Here one wave owns one query row: it reads the ac row and the two bd rows once, keeps the row in
// registers, does the max / sum reductions with wave shuffles and writes attn once.  HBM-bound: 12 B per score element.
#include "svb_common.h"
#include "../../include/svb_hip.h"

#define SVB_ATTN_MAXPL 32      /* row elements per lane kept in registers: T <= 2048 */

__global__ __launch_bounds__(256) void svb_relpos_softmax_kernel(const float* ac, const float* bd, const float* keep,
                                                                 float* attn, int B, int H, int T, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (b*H + h)*T + i
    if (row >= (long)B * H * T) return;
    const int i = (int)(row % T);
    const int b = (int)(row / ((long)H * T));
    const float* acr = ac + row * T;
    const float* bdr = bd + row * T;                                      // bd[i][.]; bd[i+1][.] follows at +T
    const float* kp = keep + (long)b * T;
    float* out = attn + row * T;
    float v[SVB_ATTN_MAXPL];
    float mx = -3.402823466e+38f;
#pragma unroll
    for (int m = 0; m < SVB_ATTN_MAXPL; ++m) {
        const int j = lane + 64 * m;
        float s = -3.402823466e+38f;
        if (j < T) {
            float sh;
            if (j <= i) sh = bdr[T - 1 - i + j];
            else if (j == i + 1) sh = 0.f;
            else sh = bdr[T + j - i - 2];
            s = (acr[j] + sh) * scale;
            if (kp[j] == 0.f) s = -3.402823466e+38f;                      // masked_fill(min)
        }
        v[m] = s;
        mx = fmaxf(mx, s);
    }
    mx = svb_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < SVB_ATTN_MAXPL; ++m) {
        const int j = lane + 64 * m;
        const float e = j < T ? expf(v[m] - mx) : 0.f;
        v[m] = e;
        sum += e;
    }
    sum = svb_wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int m = 0; m < SVB_ATTN_MAXPL; ++m) {
        const int j = lane + 64 * m;
        if (j < T) out[j] = kp[j] == 0.f ? 0.f : v[m] * inv;
    }
}

extern "C" int svb_relpos_softmax(const float* ac, const float* bd, const float* keep, float* attn, int B, int H, int T,
                                  float scale, void* stream) {
    if (!ac || !bd || !keep || !attn || B <= 0 || H <= 0 || T <= 0) return SVB_ERR_ARG;
    if (T > 64 * SVB_ATTN_MAXPL) return SVB_ERR_UNSUPPORTED;
    const long rows = (long)B * H * T;
    hipLaunchKernelGGL(svb_relpos_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ac, bd,
                       keep, attn, B, H, T, scale);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Conformer convolution module, the part between its two pointwise convs (reference
// modules/fastspeech/conformer/layers.py:47-63, eval mode):
//     u = GLU(y, dim=channels) = y[:, :C] * sigmoid(y[:, C:])
//     v = depthwise_conv1d(u; w[C][K], bias, pad (K-1)/2)
//     z = BatchNorm1d(v) with running statistics          (the PPG encoder is frozen and always in eval mode)
//     out = z * sigmoid(z)                                 (Swish)
// torch runs it as glu + MIOpen depthwise conv + batch_norm + sigmoid + mul (five passes, 2C*T reads + 4 x C*T round trips);
// here one workgroup owns one (batch, channel) row tile: the GLU'd row segment + halo goes to LDS once, every thread then
// produces one output position from K LDS reads.  HBM-bound: 12 B per output element (2 reads + 1 write).
// ------------------------------------------------------------------------------------------------------------------
#define SVB_DW_TILE 512
#define SVB_DW_MAXK 63
__global__ __launch_bounds__(256) void svb_glu_dwconv_bn_swish_kernel(const float* y, const float* w, const float* bias,
                                                                      const float* bn_w, const float* bn_b,
                                                                      const float* bn_mean, const float* bn_var, float eps,
                                                                      float* out, int B, int C, int T, int K) {
    __shared__ float u[SVB_DW_TILE + SVB_DW_MAXK];
    __shared__ float wk[SVB_DW_MAXK];
    const int tiles = (T + SVB_DW_TILE - 1) / SVB_DW_TILE;
    const int tile = blockIdx.x % tiles;
    const int bc = blockIdx.x / tiles;
    const int c = bc % C, b = bc / C;
    const int t0 = tile * SVB_DW_TILE, pad = (K - 1) / 2;
    const float* ya = y + ((size_t)b * 2 * C + c) * T;
    const float* yg = ya + (size_t)C * T;
    for (int i = threadIdx.x; i < SVB_DW_TILE + K - 1; i += 256) {
        const int t = t0 - pad + i;
        u[i] = (t >= 0 && t < T) ? ya[t] * svb_sigmoid(yg[t]) : 0.f;
    }
    if (threadIdx.x < K) wk[threadIdx.x] = w[(size_t)c * K + threadIdx.x];
    __syncthreads();
    const float scale = (bn_w ? bn_w[c] : 1.f) / sqrtf(bn_var[c] + eps);
    const float shift = (bn_b ? bn_b[c] : 0.f) - bn_mean[c] * scale + (bias ? bias[c] * scale : 0.f);
    for (int i = threadIdx.x; i < SVB_DW_TILE; i += 256) {
        const int t = t0 + i;
        if (t >= T) break;
        float acc = 0.f;
        for (int j = 0; j < K; ++j) acc = fmaf(wk[j], u[i + j], acc);
        const float z = acc * scale + shift;
        out[((size_t)b * C + c) * T + t] = z * svb_sigmoid(z);
    }
}

extern "C" int svb_glu_dwconv_bn_swish(const float* y, const float* w, const float* bias, const float* bn_w,
                                       const float* bn_b, const float* bn_mean, const float* bn_var, float eps, float* out,
                                       int B, int C, int T, int K, void* stream) {
    if (!y || !w || !bn_mean || !bn_var || !out || B <= 0 || C <= 0 || T <= 0 || K <= 0 || !(K & 1)) return SVB_ERR_ARG;
    if (K > SVB_DW_MAXK) return SVB_ERR_UNSUPPORTED;
    const long blocks = (long)B * C * ((T + SVB_DW_TILE - 1) / SVB_DW_TILE);
    if (blocks > 0x7fffffffL) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_glu_dwconv_bn_swish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, w, bias,
                       bn_w, bn_b, bn_mean, bn_var, eps, out, B, C, T, K);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Fused relative-position self-attention (forward; the PPG encoder is frozen): the content scores, the rel-shifted
// position scores, scale, key mask, softmax and the value product of espnet_transformer_attn.py:150-186 in ONE pass --
// the [B,h,T,T] `ac` and `attn` tensors never exist.
//
//   q, k, v, out : [B][H*dk][T] (the conv layout; head hh = channels hh*dk ..), dk = 64
//   pos_u        : [H][dk]  (pos_bias_u, added to q for the content term)
//   bd           : position scores (q + pos_bias_v) . linear_pos(pos_emb), UNSHIFTED, element strides (batch, head, row);
//                  the legacy rel_shift (:125-148) is applied while reading (see the header of this file)
//   keep         : [B][T] float, 0 = padded key
//
// One 64-lane wave owns 32 queries and walks the keys in blocks of 32 with an online softmax; waves do not cooperate
// (no LDS, no barrier): K and V tiles are read straight from global memory (a (batch, head)'s K/V are 2 x 144 KB, shared
// by its 18 query tiles through L2).  Every product is an MFMA 32x32x16 in split-bf16 arithmetic (hi*hi + hi*lo + lo*hi,
// fp32 accumulate), computed TRANSPOSED so that nothing has to change lanes between the two products:
//   S^T[j][i] = sum_d K[j][d] * Qu[i][d]     A = K rows j (8 strided loads per lane, coalesced along j), B = Qu (registers)
//   acc[r] of lane (i = lane & 31, kb = lane >> 5) is key j = j0 + 8*(r>>2) + 4*kb + (r&3): consecutive registers 8s'..8s'+7
//   are exactly the 8 k-slots of that lane in the B operand of
//   O^T[d][i] = sum_j V[d][j] * P[i][j]      A = V rows d: slot e of k-step s' is key j0 + 16*s' + 8*(e>>2) + 4*kb + (e&3),
//                                            i.e. two 16-byte loads along the contiguous T axis.
// bd and V are read with lanes across rows (16-byte pieces of 32..64 different rows per load instruction); measured, the kernel
// is bound by those requests, not by the MFMAs (200 us for B32 H4 T562 = 51 TFLOP/s; staging bd row-wise through LDS was
// slower).  HBM traffic: 4 B per score element + the K/V/Q/out tiles, a third of the unfused sequence's.
// ------------------------------------------------------------------------------------------------------------------
#include "svb_q.h"
#include <type_traits>
typedef __bf16 svba_bf16x8 __attribute__((ext_vector_type(8)));
typedef float svba_f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void svba_split8(const float* v, svba_bf16x8& hi, svba_bf16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) svbq_split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
    __builtin_memcpy(&hi, h, 16);
    __builtin_memcpy(&lo, l, 16);
}

// three-way split v = p0 + p1 + p2 (each bf16, 24 mantissa bits together): the exact-parity mode's operands
__device__ __forceinline__ void svba_split8x3(const float* v, svba_bf16x8& p0, svba_bf16x8& p1, svba_bf16x8& p2) {
    unsigned a[4], b[4], c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const svbq_f2 x = {v[2 * e], v[2 * e + 1]};
        const svbq_bf2 h = __builtin_convertvector(x, svbq_bf2);
        const svbq_f2 r1 = x - __builtin_convertvector(h, svbq_f2);
        const svbq_bf2 m = __builtin_convertvector(r1, svbq_bf2);
        const svbq_f2 r2 = r1 - __builtin_convertvector(m, svbq_f2);
        const svbq_bf2 l = __builtin_convertvector(r2, svbq_bf2);
        __builtin_memcpy(&a[e], &h, 4);
        __builtin_memcpy(&b[e], &m, 4);
        __builtin_memcpy(&c[e], &l, 4);
    }
    __builtin_memcpy(&p0, a, 16);
    __builtin_memcpy(&p1, b, 16);
    __builtin_memcpy(&p2, c, 16);
}
template <int NS>
__device__ __forceinline__ void svba_split(const float* v, svba_bf16x8 (&p)[NS]) {
    if constexpr (NS == 3) svba_split8x3(v, p[0], p[1], p[2]);
    else svba_split8(v, p[0], p[1]);
}
// acc += A . B over the parts: NS = 2: lo*hi + hi*lo + hi*hi (+ lo*lo first with LL); NS = 3: the six products down to 2^-16 of the
// largest (p2*q0 + p0*q2 + p1*q1 + p1*q0 + p0*q1 + p0*q0), smallest first
template <int NS, bool LL>
__device__ __forceinline__ void svba_mma(f32x16& acc, const svba_bf16x8 (&a)[NS], const svba_bf16x8 (&b)[NS]) {
    if constexpr (NS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    } else {
        if (LL) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
}

#define SVB_ATTN_DK 64
#define SVB_ATTN_SKEW_P 34      // row pitch (floats) of the skew buffer: lane il reads row jl + 31 - il -> word stride 33, conflict-free

// POS = true (round 4): the position scores are computed HERE instead of being read from a [B,H,T,T] tensor a GEMM wrote:
//   bd[i][c] = sum_d (q[i][d] + pos_v[d]) * p[d][c],   p = linear_pos(pos_emb) of the frozen encoder, handed over transposed and
//   pre-split (pt_hi / pt_lo: [H][T][dk] bf16).
// A key block needs bd[i][T-1-i+j] (j <= i) or bd[i+1][j-i-2] (j > i+1): for the wave's 32 queries and the block's 32 keys that is
// a band of 63 columns c = cb + (jl + 31 - il).  The band is one MFMA product  M[cl][il] = sum_d p^T[cb + cl][d] * Qv[il][d]
// (64 x 32, two A tiles of p^T rows -- 16-byte loads, the rows are contiguous in d), whose element for (key jl, query il) sits at
// row jl + 31 - il: the accumulators go through a per-wave LDS buffer once (written [cl][il], read skewed, conflict-free with a
// pitch of 34 floats).  The second form uses the frags of the queries i + 1.  +24 MFMAs per block against 36, and neither the
// 4 B per score element of HBM traffic nor the rocBLAS GEMM that wrote them.
// NS (round 6): parts of the operand split.  2 = hi + lo, three products per pair (bf16x3: 2^-16 per product); 3 = the exact-parity
// mode (`conv_precision: fp32`, whose reference evaluates this attention in fp32: svb_attn_set_split3): three bf16 parts, six products,
// ~2^-23 per product -- the q fragments then take 1.5x the LDS, so a workgroup is ONE wave.
template <bool POS, int NS>
__global__ __launch_bounds__(NS == 3 ? 64 : 128, 2) void svb_relpos_attn_fwd_kernel(const float* q, const float* k, const float* v, const float* pos_u,
                                                                  const float* bd, long bd_sb, long bd_sh, long bd_sr,
                                                                  const float* keep, float* out, int B, int H, int T,
                                                                  float scale, long qkv_sb, const float* pos_v,
                                                                  const unsigned short* pt_hi, const unsigned short* pt_lo,
                                                                  const unsigned short* pt_lo2) {
    constexpr int NW = NS == 3 ? 1 : 2;              // waves per workgroup
    constexpr int DK = SVB_ATTN_DK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kb = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, hh = bh - b * H;
    const int i0 = (blockIdx.x * NW + wave) * 32;
    if (i0 >= T) return;                                            // (waves are independent: no barrier below)
    const int i = i0 + l31;
    const bool iv = i < T;
    const int ic = iv ? i : T - 1;
    const size_t head = ((size_t)b * H + hh) * DK * T;                // (out: contiguous [B][H*dk][T])
    const size_t head_in = (size_t)b * qkv_sb + (size_t)hh * DK * T;  // (q, k, v: batch pitch qkv_sb -- slices of a fused projection)
    const float* qh = q + head_in;
    const float* kh = k + head_in;
    const float* vh = v + head_in;
    const float* bdh = POS ? nullptr : bd + (size_t)b * bd_sb + (size_t)hh * bd_sh;
    const float* keepb = keep + (size_t)b * T;

    // ---- B operand of the score product: Qu[i][d], d = 16s + 8kb + e.  The 8 fragments are this lane's alone; they live in a
    // private LDS slot (conflict-free 16-byte rows, no barrier) rather than in 32 registers: the kernel is bound by memory
    // latency, i.e. by how many waves fit on a SIMD.
    __shared__ uint4 q_frag[NW][4 * NS][64];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = 16 * s + 8 * kb + e;
            t[e] = qh[(size_t)d * T + ic] + pos_u[hh * DK + d];
        }
        svba_bf16x8 f[NS];
        svba_split<NS>(t, f);
#pragma unroll
        for (int u = 0; u < NS; ++u) __builtin_memcpy(&q_frag[wave][NS * s + u][lane], &f[u], 16);
    }
    // POS: Qv[i][d] = q[i][d] + pos_v[d] of this lane's query and of the next one (the j > i + 1 form reads row i + 1 of bd)
    __shared__ uint4 qv_frag[POS ? NW : 1][POS ? 2 : 1][POS ? 4 * NS : 1][POS ? 64 : 1];       // [wave][query i / i+1][fragment][lane]
    __shared__ float m_skew[POS ? NW : 1][POS ? 64 * SVB_ATTN_SKEW_P : 1];
    if (POS) {
        const int i1 = i + 1 < T ? i + 1 : T - 1;
#pragma unroll
        for (int w1 = 0; w1 < 2; ++w1)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int d = 16 * s + 8 * kb + e;
                    t[e] = qh[(size_t)d * T + (w1 ? i1 : ic)] + pos_v[hh * DK + d];
                }
                svba_bf16x8 f[NS];
                svba_split<NS>(t, f);
#pragma unroll
                for (int u = 0; u < NS; ++u) __builtin_memcpy(&qv_frag[POS ? wave : 0][POS ? w1 : 0][POS ? NS * s + u : 0][POS ? lane : 0], &f[u], 16);
            }
    }
    // the band product + skew (see the header of the kernel): sh[r] = position score of key j0 + 8*(r>>2) + 4*kb + (r&3) for this
    // lane's query, taken from band column cb + (jl + 31 - il)
    auto pos_band = [&](int which, int cb, float* sh) {
        float* ms = m_skew[POS ? wave : 0];
        __builtin_amdgcn_wave_barrier();                                  // (the previous band's reads are done)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {                                  // one 32-row tile of the band at a time: 16 accumulators
            f32x16 m;
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = 0.f;
            int rc = cb + 32 * tt + l31;
            rc = rc < 0 ? 0 : (rc > T - 1 ? T - 1 : rc);                 // (columns outside [0, T) are never selected below)
            const size_t prow = ((size_t)hh * T + rc) * DK + 8 * kb;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                svba_bf16x8 a[NS], qv[NS];
                __builtin_memcpy(&a[0], pt_hi + prow + 16 * s, 16);
                __builtin_memcpy(&a[1], pt_lo + prow + 16 * s, 16);
                if constexpr (NS == 3) __builtin_memcpy(&a[2], pt_lo2 + prow + 16 * s, 16);
#pragma unroll
                for (int u = 0; u < NS; ++u) __builtin_memcpy(&qv[u], &qv_frag[POS ? wave : 0][POS ? which : 0][POS ? NS * s + u : 0][POS ? lane : 0], 16);
                // NS = 2: four products, not three: the position scores used to come from an fp32 GEMM, and the exact-parity mode's
                // gradient check at the bench shape (1e-4 on norms) sees a three-term band (2^-16 per product) as 1.6e-4 on the
                // first layer behind the encoder
                svba_mma<NS, true>(m, a, qv);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * kb;
                ms[cl * SVB_ATTN_SKEW_P + l31] = m[r];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jl = (r & 3) + 8 * (r >> 2) + 4 * kb;
            sh[r] = ms[(jl + 31 - l31) * SVB_ATTN_SKEW_P + l31];
        }
    };
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float* bd_i = POS ? nullptr : bdh + (size_t)ic * bd_sr;                    // bd[i][.]
    const float* bd_i1 = POS ? nullptr : bdh + (size_t)(ic + 1 < T ? ic + 1 : ic) * bd_sr;      // bd[i+1][.] (never used for i = T-1)

    // One key block.  CLS (wave-uniform): 0 = the whole 32x32 block lies on or below the diagonal (every position score comes
    // from bd[i][T-1-i+j]: one 16-byte read per 4 keys), 1 = the whole block lies beyond the zero diagonal (bd[i+1][j-i-2]),
    // 2 = the block straddles the diagonal or the end of the sequence (per-element reads).  FULL: all 32 keys exist.  The two
    // common bodies are branch-free, so all of a block's loads are in flight together.
    auto key_block = [&](int j0, auto cls_c, auto full_c) {
        constexpr int CLS = decltype(cls_c)::value;
        constexpr bool FULL = decltype(full_c)::value;
        // ---- S^T block: rows j0 .. j0+31, columns i0 .. i0+31
        f32x16 s_acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
        const int jr = (FULL || j0 + l31 < T) ? j0 + l31 : T - 1;   // this lane's A row (clamped: masked below)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = kh[(size_t)(16 * s + 8 * kb + e) * T + jr];
            svba_bf16x8 a[NS], qu[NS];
            svba_split<NS>(t, a);
#pragma unroll
            for (int u = 0; u < NS; ++u) __builtin_memcpy(&qu[u], &q_frag[wave][NS * s + u][lane], 16);
            svba_mma<NS, false>(s_acc, a, qu);
        }
        // ---- + shifted position scores, scale, key mask; block maximum per query
        float p[16];
        unsigned okm = 0u;                                          // bit r: key of register r is a real, unpadded key
        float mb = -INFINITY;
        if (POS) {                                                  // the block's position scores go straight into s_acc
            float shp[16];
            if (CLS != 1) {
                pos_band(0, T - 1 - (i0 + 31) + j0, shp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                    s_acc[r] += (CLS == 0 || (j < T && j <= ic)) ? shp[r] : 0.f;
                }
            }
            if (CLS != 0) {
                pos_band(1, j0 - (i0 + 31) - 2, shp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                    s_acc[r] += (CLS == 1 || (j < T && j > ic + 1)) ? shp[r] : 0.f;
                }
            }
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int jb = j0 + 8 * g4 + 4 * kb;                    // 4 consecutive keys jb .. jb+3
            float sh[4], kp[4];
            if (FULL) {
                const svba_f4u t = *reinterpret_cast<const svba_f4u*>(keepb + jb);
                kp[0] = t.x; kp[1] = t.y; kp[2] = t.z; kp[3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) kp[e] = jb + e < T ? keepb[jb + e] : 0.f;
            }
            if (POS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sh[e] = 0.f;                // (already in s_acc)
            } else if (CLS == 0) {
                const svba_f4u t = *reinterpret_cast<const svba_f4u*>(bd_i + (T - 1 - ic + jb));
                sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w;
            } else if (CLS == 1) {
                const svba_f4u t = *reinterpret_cast<const svba_f4u*>(bd_i1 + (jb - ic - 2));
                sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = jb + e;
                    sh[e] = (j >= T || j == ic + 1) ? 0.f : (j <= ic ? bd_i[T - 1 - ic + j] : bd_i1[j - ic - 2]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                p[r] = (s_acc[r] + sh[e]) * scale;
                if (kp[e] != 0.f) {
                    okm |= 1u << r;
                    mb = fmaxf(mb, p[r]);
                }
            }
        }
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));                     // the two lane halves hold the same query
        const float m_new = fmaxf(m_run, mb);
        // (no branch here: the MFMAs and shuffles below need the whole wave.  While a query has seen nothing but padded keys
        //  m_new is -inf, every p is 0 and o, l are still 0 -- alpha only must not be NaN)
        const float alpha = m_new == -INFINITY ? 1.f : __expf(m_run - m_new);       // (m_run = -inf, m_new finite -> 0)
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = (okm >> r) & 1u ? __expf(p[r] - m_new) : 0.f;
            psum += p[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        // ---- O^T += V . P^T
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            svba_bf16x8 pp[NS];
            svba_split<NS>(p + 8 * s2, pp);
            const int ja = j0 + 16 * s2 + 4 * kb;                   // slots 0-3: ja .. ja+3, slots 4-7: ja+8 .. ja+11
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const float* vr = vh + (size_t)(db * 32 + l31) * T;
                float t[8];
                if (FULL) {
                    const svba_f4u t0 = *reinterpret_cast<const svba_f4u*>(vr + ja);
                    const svba_f4u t1 = *reinterpret_cast<const svba_f4u*>(vr + ja + 8);
                    t[0] = t0.x; t[1] = t0.y; t[2] = t0.z; t[3] = t0.w;
                    t[4] = t1.x; t[5] = t1.y; t[6] = t1.z; t[7] = t1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int j = ja + 8 * (e >> 2) + (e & 3);
                        t[e] = j < T ? vr[j] : 0.f;
                    }
                }
                svba_bf16x8 a[NS];
                svba_split<NS>(t, a);
                svba_mma<NS, false>(o[db], a, pp);
            }
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    const int i_hi = min(i0 + 31, T - 1);                            // largest query of this wave
    for (int j0 = 0; j0 < T; j0 += 32) {
        const bool full = j0 + 32 <= T;
        if (full && j0 + 31 <= i0) key_block(j0, C0{}, std::true_type{});
        else if (full && j0 > i_hi + 1) key_block(j0, C1{}, std::true_type{});
        else if (full) key_block(j0, C2{}, std::true_type{});
        else key_block(j0, C2{}, std::false_type{});
    }
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;              // (a query whose keys are all padded gets zeros, as :183-186)
    if (iv) {
        float* oh = out + head;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                oh[(size_t)d * T + i] = o[db][r] * inv;
            }
    }
}

// process-wide arithmetic mode of the attention products (see NS above); the host side sets it with `conv_precision`
static int g_svb_attn_split3 = 1;      // (default = the host side's default precision, fp32)
extern "C" void svb_attn_set_split3(int on) { g_svb_attn_split3 = on ? 1 : 0; }
extern "C" int svb_attn_get_split3(void) { return g_svb_attn_split3; }

template <bool POS, int NS>
static void attn_launch(const float* q, const float* k, const float* v, const float* pos_u, const float* bd, long bd_sb, long bd_sh,
                        long bd_sr, const float* keep, float* out, int B, int H, int T, float scale, long qkv_sb, const float* pos_v,
                        const unsigned short* pt_hi, const unsigned short* pt_lo, const unsigned short* pt_lo2, void* stream) {
    constexpr int NW = NS == 3 ? 1 : 2;
    hipLaunchKernelGGL((svb_relpos_attn_fwd_kernel<POS, NS>), dim3((T + 32 * NW - 1) / (32 * NW), B * H), dim3(64 * NW), 0,
                       (hipStream_t)stream, q, k, v, pos_u, bd, bd_sb, bd_sh, bd_sr, keep, out, B, H, T, scale, qkv_sb, pos_v, pt_hi, pt_lo,
                       pt_lo2);
}

extern "C" int svb_relpos_attn_fwd(const float* q, const float* k, const float* v, long qkv_sb, const float* pos_u, const float* bd,
                                   long bd_sb, long bd_sh, long bd_sr, const float* keep, float* out, int B, int H, int dk, int T,
                                   float scale, void* stream) {
    if (!q || !k || !v || !pos_u || !bd || !keep || !out || B <= 0 || H <= 0 || T <= 0) return SVB_ERR_ARG;
    if (dk != SVB_ATTN_DK || (long)B * H > 65535) return SVB_ERR_UNSUPPORTED;
    if (qkv_sb < (long)H * dk * T) return SVB_ERR_ARG;
    if (g_svb_attn_split3)
        attn_launch<false, 3>(q, k, v, pos_u, bd, bd_sb, bd_sh, bd_sr, keep, out, B, H, T, scale, qkv_sb, nullptr, nullptr, nullptr, nullptr,
                              stream);
    else
        attn_launch<false, 2>(q, k, v, pos_u, bd, bd_sb, bd_sh, bd_sr, keep, out, B, H, T, scale, qkv_sb, nullptr, nullptr, nullptr, nullptr,
                              stream);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_relpos_attn_pos_fwd(const float* q, const float* k, const float* v, long qkv_sb, const float* pos_u,
                                       const float* pos_v, const unsigned short* pt_hi, const unsigned short* pt_lo,
                                       const unsigned short* pt_lo2, const float* keep, float* out, int B, int H, int dk, int T,
                                       float scale, void* stream) {
    if (!q || !k || !v || !pos_u || !pos_v || !pt_hi || !pt_lo || !keep || !out || B <= 0 || H <= 0 || T <= 0) return SVB_ERR_ARG;
    if (dk != SVB_ATTN_DK || (long)B * H > 65535) return SVB_ERR_UNSUPPORTED;
    if (qkv_sb < (long)H * dk * T) return SVB_ERR_ARG;
    // (the three-way form needs the table's third part; a two-part table runs the two-way kernel whatever the mode)
    if (g_svb_attn_split3 && pt_lo2)
        attn_launch<true, 3>(q, k, v, pos_u, nullptr, 0L, 0L, 0L, keep, out, B, H, T, scale, qkv_sb, pos_v, pt_hi, pt_lo, pt_lo2, stream);
    else
        attn_launch<true, 2>(q, k, v, pos_u, nullptr, 0L, 0L, 0L, keep, out, B, H, T, scale, qkv_sb, pos_v, pt_hi, pt_lo, nullptr, stream);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
