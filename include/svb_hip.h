/* svb_hip.h -- C ABI of libsvb_hip.so, the MI355X (gfx950) kernel library under the NeuralSVB hot path.
 *
 * The reference (MoonInTheRiver/NeuralSVB) has no FFI: every op below replaces a *stock torch op sequence*
 * inside one of the reference's nn.Modules.  Each entry point cites the reference
 * interface it sits under.  Conventions:
 *   - plain pointers + sizes only; all pointers are DEVICE pointers to contiguous fp32 unless noted;
 *   - activations are [B, C, T] (time contiguous), exactly the reference's Conv1d layout;
 *   - no allocation inside: workspaces are passed in, sized by the matching *_workspace_* query;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous and re-entrant per stream;
 *   - return 0 on success, <0 on error (-1 bad argument, -2 launch failure, -3 unsupported shape); never throws.
 */
#ifndef SVB_HIP_H
#define SVB_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SVB_ABI_VERSION 10
int svb_abi_version(void);

/* ---- fused conv epilogue / prologue description ------------------------------------------------------
 * y = mask * ( residual + out_gate' * act( conv(x * in_gate') + bias ) )
 *   in_gate'  = 1 where in_gate  > 0 else in_slope        (in_gate == x  -> LeakyReLU(x) fused on the load;
 *                                                          in_gate == saved y -> activation backward)
 *   out_gate' = 1 where out_gate > 0 else out_gate_slope  (data-gradient through a fused input LeakyReLU)
 * Any pointer may be NULL (term skipped).  out_act: 0 none, 1 ReLU, 2 LeakyReLU(out_slope), 3 tanh.
 * force_cfg: 0 = auto tile choice, 1..5 (bf16x3 variants: 1..7) = force tile configuration.                              */
typedef struct SvbConvEpilogue {
    const float* bias;      /* [Cout] */
    const float* in_gate;   /* same shape as x */
    const float* out_gate;  /* same shape as y */
    const float* residual;  /* same shape as y */
    const float* mask;      /* [B, Tout] */
    float in_slope, out_slope, out_gate_slope;
    int out_act;
    int force_cfg;
    const unsigned short* x_q; /* bf16x3 entry points only: optional Q image of x (svb_split_q layout).  When given (and
                                * in_gate is NULL) the kernel stages its input tiles from it with plain 16-byte copies
                                * instead of splitting fp32 x again in every consuming workgroup; results are bit-identical. */
} SvbConvEpilogue;

/* Weight pack (+ WeightNorm forward  w = g * v / ||v||, norm over all dims but 0).
 * Replaces torch.nn.utils.weight_norm's per-forward recompute (reference modules/fastspeech/fs2_vae.py:42,48,58;
 * modules/hifigan/hifigan.py:33-50,117,124,140).  v: [d0][d1][k] reference layout (Conv1d: d0=Cout, d1=Cin/groups;
 * ConvTranspose1d: d0=Cin, d1=Cout).  pa: [k][d1][d0], pb: [k][d0][d1] (either may be NULL).                */
int svb_weight_pack(const float* v, const float* g, float* pa, float* pb, int d0, int d1, int k, int weight_norm,
                    void* stream);

/* Conv1d forward: y[b,co,q] = sum_{ci,j} w[co,ci,j] x[b,ci,q*stride + j*dil - pad].
 * Replaces F.conv1d at reference fs2_vae.py:73,83,109-114,125; vae_models.py:86-94; common_layers.py:739-773;
 * hifigan.py:33-61,117,140,202-223 (Conv2d (k,1) over period-major layout),261-286.
 * wp = pa of svb_weight_pack ([k][Cin/groups][Cout]).  Also computes ConvTranspose1d's data gradient.         */
int svb_conv1d_forward(const float* x, const float* wp, float* y, int B, int Cin, int Cout, int groups, int Tin,
                       int Tout, int k, int stride, int pad, int dil, const SvbConvEpilogue* epi, void* stream);

/* Stride-1 conv with an arbitrary tap table (groups 1):  y[b,co,q] = sum_{ci,t} w[t][ci][co] x[b,ci,q + tap_off[t]], zero
 * outside [0,Tin).  wp = a [ntaps][Cin][Cout] pack (pa of a [Cout][Cin][ntaps] weight; pb of it, with Cin/Cout swapped and
 * the offsets negated, gives the data gradient).  Used for the mel critic's 3x3 stride-2 Conv2d (reference
 * modules/fastspeech/multi_window_disc.py:14-31): after a space-to-depth of the input it is a 2x2 stride-1 conv, i.e. a 1-D
 * conv over the row-padded flattened image with taps {-P-1, -P, -1, 0} -- no 9x im2col expansion.                      */
int svb_conv1d_taps(const float* x, const float* wp, float* y, int B, int Cin, int Cout, int Tin, int Tout, int ntaps,
                    const int* tap_off, const SvbConvEpilogue* epi, void* stream);
int svb_conv1d_taps_bf16x3(const float* x, const unsigned short* q_hi, const unsigned short* q_lo, float* y, int B, int Cin,
                           int Cout, int Tin, int Tout, int ntaps, const int* tap_off, const SvbConvEpilogue* epi, void* stream);

/* Tile configuration (0..4) the conv launcher picks for Cout/groups output channels and nq_max output positions
 * per phase: {64x128, 128x96, 128x128, 64x64, 32x128} = svb_conv1d_mfma_kernel<2,2,2>, <4,1,3>, <4,1,4>, <2,2,1>, <1,4,1>.
 * (profiling aid: lets a caller name the kernel instantiation a launch used)                                   */
int svb_conv1d_pick_cfg(int cout_g, int nq_max, int nz /* batch * phases * groups */);

/* Transposed conv (gather form): y[b,co,p] = sum_{ci,j : p = t*stride - pad + j*dil} w[ci,co,j] x[b,ci,t].
 * Replaces F.conv_transpose1d (reference vae_models.py:115-120,126; hifigan.py:122-125,156) and the data
 * gradient of F.conv1d.  wp: [k][Cin][Cout/groups] (= pb of a ConvTranspose1d weight, = pb of a Conv1d weight
 * when used as that conv's dgrad with Cin:=conv.Cout, Cout:=conv.Cin).                                       */
int svb_conv1d_transposed(const float* x, const float* wp, float* y, int B, int Cin, int Cout, int groups, int Tin,
                          int Tout, int k, int stride, int pad, int dil, const SvbConvEpilogue* epi, void* stream);

/* ---- "bf16x3" variants: same operations on the bf16 matrix cores with an fp32-class operand split
 * (v = hi + lo in bf16; products hi*hi + hi*lo + lo*hi, fp32 accumulate; relative error ~1e-5 per conv).
 * svb_weight_pack_bf16x3 writes bf16 hi/lo weights in the two operand layouts
 *   qa: [k][ceil(d1/16)][d0][16]  (Conv1d forward, ConvTranspose1d data-gradient)
 *   qb: [k][groups][ceil((d0/groups)/16)][d1][16]  (ConvTranspose1d forward, Conv1d data-gradient);
 * padding entries (d1 or d0/groups not a multiple of 16) are written as zeros by the pack itself (ABI v10): buffers need no prior fill. */
int svb_weight_pack_bf16x3(const float* v, const float* g, unsigned short* qa_hi, unsigned short* qa_lo,
                           unsigned short* qb_hi, unsigned short* qb_lo, int d0, int d1, int k, int groups, int weight_norm,
                           void* stream);
/* The same pack for MANY weights in one launch (e.g. every conv weight of an optimizer right after its step).  `descs` is a
 * DEVICE array of n descriptors; row_start = number of d0 rows of all earlier descriptors (descs[0].row_start = 0),
 * total_rows = sum of d0.  Replaces ~70 per-forward pack launches of a training step by one per optimizer step.      */
typedef struct SvbPackDesc {
    const float* v;
    const float* g;               /* NULL without weight norm */
    unsigned short *qa_hi, *qa_lo, *qb_hi, *qb_lo;   /* either layout may be NULL */
    int d0, d1, k, groups, weight_norm, row_start;
} SvbPackDesc;
int svb_weight_pack_bf16x3_multi(const SvbPackDesc* descs, int n, int total_rows, void* stream);
int svb_conv1d_forward_bf16x3(const float* x, const unsigned short* qa_hi, const unsigned short* qa_lo, float* y, int B,
                              int Cin, int Cout, int groups, int Tin, int Tout, int k, int stride, int pad, int dil,
                              const SvbConvEpilogue* epi, void* stream);
int svb_conv1d_transposed_bf16x3(const float* x, const unsigned short* qb_hi, const unsigned short* qb_lo, float* y, int B,
                                 int Cin, int Cout, int groups, int Tin, int Tout, int k, int stride, int pad, int dil,
                                 const SvbConvEpilogue* epi, void* stream);

/* Instrumentation build only (`make -C neuralsvb_amd/csrc instr` -> libsvb_hip_instr.so, -DSVB_INSTRUMENT; tools/ load it, the
 * product never does): cycle stamps of the conv kernels' stages and experiment switches.  svb_debug_set_timing_buffer: a device
 * buffer of 64*32*8 uint64 for the first 64 workgroups; svb_debug_set_tw: stamp buffer + ablation mask of conv1d_tw.hip.       */
#ifdef SVB_INSTRUMENT
void svb_debug_set_timing_buffer(void* buf);
void svb_debug_set_tw(void* buf, int ablate, int drain_per_pair);
#endif

/* Weight gradient, stage 1 (split-K partials): part[s][a][b][j] += A[n,a,q] * Bt[n,b,q*sx + j*dil - pad].
 * Conv1d: A = dy (CA=Cout), Bt = x (CB=Cin); ConvTranspose1d: A = x (CA=Cin), Bt = dy (CB=Cout).
 * a_gate/b_gate: optional activation-derivative gates (see SvbConvEpilogue).  Workspace = floats returned by
 * svb_conv1d_wgrad_workspace_floats(), which also returns the split count to pass as `nsplit`.                */
size_t svb_conv1d_wgrad_workspace_floats(int B, int CA, int CB, int groups, int TA, int k, int sx, int* nsplit_out);
int svb_conv1d_wgrad(const float* a, const float* b, float* part, int B, int CA, int CB, int groups, int TA, int TB,
                     int k, int sx, int pad, int dil, const float* a_gate, float a_slope, const float* b_gate,
                     float b_slope, int nsplit, void* stream);
/* The same stage 1 on the bf16 matrix cores (bf16x3 split, see above).  Strided convs (sx > 1, dil 1) are decomposed
 * into sx stride-1 problems over the phase subsequences of Bt.  The workspace query returns 0 floats (and nsplit 0) for
 * shapes outside its envelope (sx 1: (taps per pass - 1)*dil > 23; sx > 1: dil != 1): use svb_conv1d_wgrad then.
 * Partials have the same layout, so svb_wgrad_reduce finishes either.                                           */
size_t svb_conv1d_wgrad_bf16x3_workspace_floats(int B, int CA, int CB, int groups, int TA, int k, int sx, int pad, int dil,
                                                int* nsplit_out);
int svb_conv1d_wgrad_bf16x3(const float* a, const float* b, float* part, int B, int CA, int CB, int groups, int TA, int TB,
                            int k, int sx, int pad, int dil, const float* a_gate, float a_slope, const float* b_gate,
                            float b_slope, int nsplit, float* bias_part, void* stream);
/* bias_part (optional, [nsplit][CA] floats): per-split row sums of the gated A operand -- with A = dy these are the bias
 * gradient partials (reference: autograd of the `bias` argument of F.conv1d); svb_wgrad_reduce sums them into db.     */
/* Stage 2: reduce partials (+ WeightNorm backward: dv, dg from dW).  rows = d0, rowlen = d1*k.
 * bias_part/db (optional): also sum the [nsplit][rows] bias-gradient partials of stage 1 into db[rows].
 * accumulate: 1 = add into dv / dg / db instead of overwriting them (gradients written straight into `.grad` buffers; with
 * weight_norm this needs 16-byte aligned rows and rowlen <= 8192, SVB_ERR_UNSUPPORTED otherwise); 2 = add into db only. */
int svb_wgrad_reduce(const float* part, int nsplit, const float* v, const float* g, float* dv, float* dg, int rows,
                     int rowlen, int weight_norm, int accumulate, const float* bias_part, float* db, void* stream);
/* Several stage-2 reduces in one launch per 24 descriptors (a backward pass can defer all of its reduces to its end): each
 * descriptor = the arguments of one svb_wgrad_reduce call (part is [nsplit][rows][rowlen]); `descs` is a HOST array (it is
 * passed through the kernel-argument segment); row_start is filled in by the call.                                    */
typedef struct SvbReduceDesc {
    const float* part; const float* v; const float* g; float* dv; float* dg; const float* bias_part; float* db;
    int nsplit, rows, rowlen, weight_norm, accumulate, row_start;
} SvbReduceDesc;
int svb_wgrad_reduce_multi(const SvbReduceDesc* descs, int n, void* stream);
/* db[c] = sum_{b,t} dy[b,c,t] * gate'(gate[b,c,t])                                                          */
int svb_bias_grad(const float* dy, const float* gate, float slope, float* db, int B, int C, int T, void* stream);

/* ---- GAN feature-matching loss as multi-tensor launches (reference modules/hifigan/hifigan.py:328-335 `feature_loss`:
 * 2 * sum_p mean(|r_p - g_p|) over the discriminators' feature-map pairs).  fwd: out[0] (+)= sum_p scale_p * sum |a_p - b_p|
 * (scale_p = 2 / n_p for the reference's loss); partials: floats of workspace, one per block (svb_l1_pairs_blocks).  bwd: for each
 * pair  db = gout[0] * scale_p * sign(b - a),  da = -db  (either may be NULL).  At most SVB_L1_MAX_PAIRS pairs per call; the
 * descriptors are HOST memory (they travel in the kernel-argument segment); block0 is filled in by the library.             */
#define SVB_L1_MAX_PAIRS 32
typedef struct SvbL1Pair {
    const float* a; const float* b; float* da; float* db; long n; float scale; int block0;
    int mode; float target;      /* mode 1 (ABI v10): the term is sum (a - target)^2 (b, db unused) -- the LS-GAN terms of
                                  * discriminator_loss / generator_loss, modules/hifigan/hifigan.py:338-365; da = g * scale * 2 (a - target) */
} SvbL1Pair;
int svb_l1_pairs_blocks(const SvbL1Pair* pairs, int n);
int svb_l1_pairs_fwd(const SvbL1Pair* pairs, int n, float* partials, float* out, int accumulate, void* stream);
int svb_l1_pairs_bwd(const SvbL1Pair* pairs, int n, const float* gout, void* stream);
/* out = scale * (a + b [+ c]) over n contiguous floats (c may be NULL; ABI v10): the HifiGAN generator's mean over its parallel
 * ResBlocks (reference modules/hifigan/hifigan.py:157-163) in one pass instead of two adds and a division.                       */
int svb_sum_scale(const float* a, const float* b, const float* c, float scale, float* out, long n, void* stream);

/* ---- WaveNet-style gated layer pieces (reference modules/fastspeech/fs2_vae.py:10-16,61-91) ---------------
 * gate fwd: acts[b,c,t] = tanh(xin[b,c,t] + g[b,goff+c,t]) * sigmoid(xin[b,C+c,t] + g[b,goff+C+c,t])
 * gate bwd: d_xin from d_acts; the same values are the gradient of g's slice and are optionally also written
 * into dg[b, g_off : g_off+2C, t] (dg has g_channels channels; dxin or dg may be NULL).                      */
int svb_wn_gate_fwd(const float* xin, const float* g, float* acts, unsigned short* acts_q, int B, int C, int T,
                    int g_channels, int g_off, void* stream);
int svb_wn_gate_bwd(const float* xin, const float* g, const float* dacts, float* dxin, float* dg, unsigned short* dxin_q,
                    int B, int C, int T, int g_channels, int g_off, void* stream);
/* The *_q arguments of the four gated-stack kernels (all optional, NULL = skip): also emit the result's Q image (see
 * svb_split_q) for the bf16x3 conv that consumes it next.  dxin_q / drs_q describe 2C-channel tensors and need C % 16 == 0. */
/* res/skip update: x_new = (x + rs[:, :C]) * mask ; out_new = out + rs[:, C:]   (last layer: out += rs, x untouched;
 * rs then has C channels and x/x_new may be NULL).  In-place allowed (x_new == x, out_new == out).            */
int svb_wn_res_skip(const float* x, const float* rs, const float* mask, const float* out, float* x_new, float* out_new,
                    unsigned short* x_new_q, int B, int C, int T, int last, void* stream);
/* backward of the (non-last) res/skip update: drs[:, :C] = dx_new*mask, drs[:, C:] = dout, dxm = dx_new*mask
 * (dx_new NULL = zeros, dxm may be NULL).                                                                    */
int svb_wn_res_skip_bwd(const float* dx_new, const float* dout, const float* mask, float* drs, float* dxm,
                        unsigned short* drs_q, int B, int C, int T, void* stream);

/* ---- Pre-split activations ("Q image") for the bf16x3 convs: xq[b][chunk][t][0..15] = hi, [16..31] = lo of the 16
 * channels of chunk `chunk` at position t (bf16; v = hi + lo, hi = rne(v), lo = rne(v - hi)); ceil(C/16) chunks, channels
 * beyond C are zero; optional mask[b,t] multiplies x first.  xq: B * ceil(C/16) * T * 32 bf16, 16-byte aligned.
 * Replaces the per-consumer fp32 -> bf16 split of the conv kernels (no reference counterpart: layout plumbing). */
int svb_split_q(const float* x, const float* mask, unsigned short* xq, int B, int C, int T, void* stream);

/* ---- The whole gated stack (`WN.forward`, reference modules/fastspeech/fs2_vae.py:61-91; p_dropout = 0) and its backward as
 * ONE call each: a host-side loop over the entry points above (bf16x3 convs, gate, res/skip, weight gradients + their reduces),
 * in the order the Python layer issues them -- bit-identical results, ~130 Python-issued launches per generator pass less.
 * All tensors fp32 [B][channels][T] contiguous; layer i has dilation dil_rate^i and padding (k*d - d)/2.                        */
#define SVB_WN_MAX_LAYERS 16
typedef struct SvbWnLayer {
    const unsigned short *in_a_hi, *in_a_lo;   /* in-layer conv C -> 2C, k taps: packed A image (forward) ...                */
    const unsigned short *in_b_hi, *in_b_lo;   /* ... and B image (data gradient); svb_weight_pack_bf16x3 layouts            */
    const unsigned short *rs_a_hi, *rs_a_lo;   /* res/skip 1x1 conv C -> rs_cout                                              */
    const unsigned short *rs_b_hi, *rs_b_lo;
    const float *in_bias, *rs_bias;            /* [2C], [rs_cout]                                                             */
    const float *in_v, *in_g, *rs_v, *rs_g;    /* backward: the weights (v, and g when weight-normalised, else g = NULL)      */
    float *d_in_v, *d_in_g, *d_in_b;           /* backward: gradient buffers ACCUMULATED into (d_in_v NULL: in-layer frozen)   */
    float *d_rs_v, *d_rs_g, *d_rs_b;
    int rs_cout;                               /* 2C; C for the last layer                                                    */
    int cfg_in_fwd, cfg_rs_fwd, cfg_in_bwd, cfg_rs_bwd;   /* force_cfg of the four convs (0 = heuristic tile)                */
} SvbWnLayer;
typedef struct SvbWnStack {
    int B, C, T, n_layers, k, dil_rate, g_channels;
    const float* x0;      /* stack input [B][C][T]                                                                            */
    const float* mask;    /* [B][T] or NULL                                                                                   */
    const float* G;       /* conditioning [B][g_channels][T] (layer i reads channels i*2C .. (i+1)*2C) or NULL                */
    float* xbuf;          /* [n_layers-1][B][C][T]: the inputs of layers 1.. (written forward, read backward)                 */
    float* xin;           /* [n_layers][B][2C][T]: in-layer conv outputs (saved for the gate's backward)                      */
    float* acts;          /* [n_layers][B][C][T]: gate outputs (saved for the res/skip conv's weight gradient)                */
    SvbWnLayer layer[SVB_WN_MAX_LAYERS];
} SvbWnStack;
/* out [B][C][T] receives the skip sum WITHOUT the final mask (the caller multiplies); rs_scratch: [B][2C][T].               */
int svb_wn_stack_forward(const SvbWnStack* s, float* rs_scratch, float* out, void* stream);
typedef struct SvbWnBackward {
    const float* dout;    /* gradient of the (masked) output, already multiplied by the mask: [B][C][T]                       */
    float* dx;            /* [B][C][T]: on return the gradient of x0 (valid if need_dx0)                                       */
    float* dG;            /* [B][g_channels][T] gradient of G (every slice written) or NULL                                   */
    float *drs, *dxin;    /* [n_layers][B][2C][T] each: per layer, because the weight gradients read them on the side stream  */
    float *dacts, *dxm;   /* [B][C][T] scratch                                                                                */
    float* arena;         /* split-K partials of the pending weight gradients; a full arena is reduced and reused             */
    size_t arena_floats;
    int need_dx0;
} SvbWnBackward;
/* side_stream (optional): the weight gradients and their reduces are issued there after it has caught up with `stream`; the
 * caller joins it.  Weight gradients whose d_*_v is NULL are skipped.  SVB_ERR_UNSUPPORTED: a weight gradient does not fit
 * the bf16x3 kernel's envelope or the arena (nothing has been launched for that gradient; earlier launches stand).        */
int svb_wn_stack_backward(const SvbWnStack* s, const SvbWnBackward* b, void* stream, void* side_stream);

/* ---- Host-side executor of one window's tower of the mel critic (reference modules/voice_conversion/multi_window_disc.py:14-64:
 * blocks of [Conv2d 3x3 s2 + LeakyReLU -> Dropout2d -> InstanceNorm2d] chained through the conv's space-to-depth layout, then the
 * Linear score layer): ONE call issues every launch of the forward / of the backward through the entry points above
 * (svb_conv1d_taps_bf16x3, svb_crop_drop_inorm_*, svb_plane_score_*, svb_conv1d_wgrad_bf16x3 + svb_wgrad_reduce, svb_s2_weight_bwd).
 * Layouts are those entry points': x4 = space-to-depth planes [4C][N][H/2+1][W/2+1] of the (N, C, H, W) input; block b's conv
 * output y4 [cout][N][Ho+1][Wo+1]; `out` = the next block's x4, or [cout][N][Ho][Wo] for the last block.                          */
#define SVB_CT_MAX_BLOCKS 4
typedef struct SvbCtBlock {
    const unsigned short *a_hi, *a_lo;   /* packed derived 2x2 kernel [cout][4C][4] (svb_s2_weight + svb_weight_pack_bf16x3), forward image */
    const unsigned short *b_hi, *b_lo;   /* its data-gradient image (backward with dx4 only)                                           */
    const float* bias;                   /* [cout] or NULL                                                                             */
    const float* keep;                   /* Dropout2d factors [N][cout] or NULL                                                        */
    const float *gamma, *beta;           /* InstanceNorm2d affine or NULL (no norm)                                                    */
    float eps;
    int cout, cfg_fwd, cfg_bwd;          /* tile configurations of the two convs (0 = the kernel's heuristic)                          */
    float *y4, *out, *stats;             /* forward results (stats [cout][N][2] with gamma)                                            */
    float *dy4, *dgb, *dx4;              /* backward: conv output gradient, per-plane sums [2][N][cout] (with gamma), input gradient (NULL:
                                          * not needed -- block 0 only)                                                                 */
    float *d_weight, *d_bias;            /* gradient buffers of the 3x3 weight [cout][C][3][3] / bias to ACCUMULATE into (NULL: skipped)  */
} SvbCtBlock;
typedef struct SvbCriticTower {
    int nb, N, C, H, W;                  /* blocks; planes of the tower's input                                                        */
    int has_slope; float slope;          /* LeakyReLU after every conv                                                                  */
    const float* x4;
    const float *score_w, *score_b;      /* [C_last * H_last * W_last], [1] or NULL                                                    */
    float* score;                        /* [N]                                                                                         */
    SvbCtBlock blk[SVB_CT_MAX_BLOCKS];
} SvbCriticTower;
int svb_critic_tower_forward(const SvbCriticTower* t, void* stream);
typedef struct SvbCtBackward {
    const float* ds; long ds_stride;     /* score cotangents ds[n * ds_stride]                                                         */
    float* dh;                           /* [C_last][N][H_last][W_last] scratch: gradient of the last block's output                   */
    float *d_score_w, *d_score_b;        /* outputs (overwritten) or NULL                                                               */
    float* ws; size_t ws_floats;         /* workspace of the weight gradients (partials + two [cout][4C][2] results of the largest block) */
} SvbCtBackward;
/* side_stream (optional): the weight gradients, their reduces and the gather into d_weight are issued there after it has caught up
 * with `stream` (one event per block); the caller joins it.  SVB_ERR_UNSUPPORTED: a weight gradient does not fit the workspace. */
int svb_critic_tower_backward(const SvbCriticTower* t, const SvbCtBackward* g, void* stream, void* side_stream);

/* ---- LayerNorm over the channel (last) dim of [rows, C] (reference modules/fastspeech/conformer/layers.py:160-170,
 * conformer.py:30; torch.nn.LayerNorm eps 1e-5).                                                             */
int svb_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int rows, int C, float eps, void* stream);
/* Same normalisation over the channel dim of an NCT tensor [B, C, T] (forward only: the PPG encoder is frozen,
 * reference tasks/singing/svb_vae_task.py:558-561).                                                          */
int svb_layernorm_nct_fwd(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int T, float eps,
                          void* stream);
int svb_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* mean, const float* rstd,
                      float* dx, float* dgamma_part, float* dbeta_part, int rows, int C, int n_part, void* stream);

/* ---- Relative-position self-attention glue of the frozen conformer PPG encoder (reference
 * modules/commons/espnet_transformer_attn.py:125-186): attn = masked softmax over keys of (ac + rel_shift(bd)) * scale,
 * ac = (q+u) k^T and bd = (q+v) p^T both [B,H,T,T]; keep [B,T] (1 = real frame, 0 = padding); T <= 2048.        */
int svb_relpos_softmax(const float* ac, const float* bd, const float* keep, float* attn, int B, int H, int T, float scale,
                       void* stream);

/* ---- the same attention fused end to end (forward only; the PPG encoder is frozen): content scores (q + pos_u) . k, the
 * rel-shifted position scores read from `bd` (UNSHIFTED (q + pos_v) . linear_pos(pos_emb), element strides batch / head /
 * row), 1/sqrt(dk) scale, key mask, softmax and the product with v in one kernel -- no [B,h,T,T] score or attention
 * tensor is written.  q, k, v: [B][H*dk][T] with batch pitch qkv_sb elements (>= H*dk*T: they may be slices of one fused
 * projection's output, ABI v7); out: contiguous [B][H*dk][T] (conv layout); pos_u [H][dk]; keep [B][T]; dk must be 64.
 * Split-bf16 MFMA arithmetic (fp32-class), online softmax in fp32.                                                       */
int svb_relpos_attn_fwd(const float* q, const float* k, const float* v, long qkv_sb, const float* pos_u, const float* bd,
                        long bd_sb, long bd_sh, long bd_sr, const float* keep, float* out, int B, int H, int dk, int T, float scale,
                        void* stream);

/* The same attention with the position scores computed in the kernel (no [B,H,T,T] tensor at all): pos_v [H][dk] (the `v` bias of
 * espnet_transformer_attn.py:150-186), pt_hi / pt_lo [H][T][dk] bf16 = linear_pos(pos_emb) transposed and split into
 * bf16 hi + lo (p = hi + lo to fp32 class; constant for a frozen encoder and a given T: prepared once by the host).
 * Each key block computes the 63-column band of (q + pos_v) . p it needs as one MFMA product and skews it through LDS.      */
int svb_relpos_attn_pos_fwd(const float* q, const float* k, const float* v, long qkv_sb, const float* pos_u, const float* pos_v,
                            const unsigned short* pt_hi, const unsigned short* pt_lo, const unsigned short* pt_lo2, const float* keep,
                            float* out, int B, int H,
                            int dk, int T, float scale, void* stream);

/* ---- Conformer convolution module between its pointwise convs, eval mode (reference
 * modules/fastspeech/conformer/layers.py:47-63): out = Swish(BatchNorm_eval(depthwise_conv1d(GLU(y)))).  y [B, 2C, T],
 * w [C][K] (K odd <= 63, padding (K-1)/2), bias [C] or NULL, BatchNorm weight / bias (NULL = 1 / 0), running mean / var,
 * out [B, C, T].  Forward only (the PPG encoder is frozen).                                                        */
int svb_glu_dwconv_bn_swish(const float* y, const float* w, const float* bias, const float* bn_w, const float* bn_b,
                            const float* bn_mean, const float* bn_var, float eps, float* out, int B, int C, int T, int K,
                            void* stream);

/* ---- im2col / col2im for small strided Conv2d layers (reference modules/fastspeech/multi_window_disc.py:14-31).
 * The column matrix is addressed as cols[b*cols_sb + row*cols_sk + pos] (row = (c,jh,jw), pos = ho*Wo+wo): per clip
 * ([B][K][L]: cols_sb = K*L, cols_sk = L) or with the batch folded into the position axis ([K][B*L]: cols_sb = L,
 * cols_sk = B*L), x as x[b*x_sb + c*x_sc + h*W + w].  The GEMM runs as svb_conv1d_forward(k=1) over cols, its data
 * gradient as svb_conv1d_transposed(k=1) followed by svb_col2im (dx: contiguous [B,C,H,W]), its weight gradient as
 * svb_conv1d_wgrad.                                                                                              */
int svb_im2col(const float* x, float* cols, int B, int C, int H, int W, int KH, int KW, int SH, int SW, int PH, int PW,
               int Ho, int Wo, long x_sb, long x_sc, long cols_sb, long cols_sk, void* stream);
int svb_col2im(const float* dcols, float* dx, int B, int C, int H, int W, int KH, int KW, int SH, int SW, int PH, int PW,
               int Ho, int Wo, long cols_sb, long cols_sk, void* stream);

/* Space-to-depth + zero border for the stride-2 3x3 Conv2d of the mel critic without im2col (see svb_conv1d_taps):
 * out[(ph*2+pw)*C + c][n][i+1][j+1] = x[n][c][2i+ph][2j+pw], row 0 / column 0 of each plane zero; x addressed with element
 * strides (n, c, h, w), out contiguous [4C][N][H/2+1][W/2+1]; H, W even.  _bwd: the inverse gather into a contiguous
 * dx [N][C][H][W].                                                                                                  */
int svb_s2d_pad(const float* x, float* out, int N, int C, int H, int W, long sn, long sc, long sh, long sw, void* stream);
int svb_s2d_pad_bwd(const float* dout, float* dx, int N, int C, int H, int W, void* stream);

/* Random-window crop of stacked critic calls straight into the first block's planes (reference multi_window_disc.py:131-152:
 * x[:, :, s:s+wl] per window length; the calls of a pass -- real / generated mels of each way -- stacked along the batch):
 * source k = xs[k], a [B][T][F] mel addressed with element strides strides[3k..3k+2] (batch, time, bin), window start
 * starts[k]; out = svb_s2d_pad's layout [4][n_src*B][wl/2+1][F/2+1] with C = 1 and clip b of source k at n = k*B + b.
 * xs / strides / starts (and dplanes / wls below) are HOST arrays.  n_src <= 8, wl and F even.
 * _bwd, all window lengths at once: dx [n_src][B][T][F] (contiguous) = sum over windows w with starts[w*n_src+k] <= t <
 * starts[..] + wls[w] of dplanes[w] at the matching plane element, 0 where no window covers t.  n_win <= 4.          */
int svb_win_s2d(const float* const* xs, const long* strides, const int* starts, float* out, int n_src, int B, int wl, int F,
                void* stream);
int svb_win_s2d_bwd(const float* const* dplanes, const int* wls, const int* starts, float* dx, int n_win, int n_src, int B, int T,
                    int F, void* stream);

/* ---- the rest of a critic block around that conv (reference multi_window_disc.py:14-31, :62-64), batch-stacked planes.
 * svb_s2_weight:      w [Cout][C][3][3] -> w4 [Cout][4C][4], the 2x2 stride-1 kernel over the space-to-depth planes
 *                     (channel (ph*2+pw)*C + c, tap (di+1)*2 + (dj+1); kernel row kh <-> (di,ph): 0:(-1,1) 1:(0,0) 2:(0,1)).
 * svb_s2_weight_bwd:  the inverse gather of the weight gradient; dwa / dwb = its tap pairs {0,1} / {2,3}, each [Cout][4C][2];
 *                     accumulate != 0 adds into dw (a parameter's .grad).
 * svb_crop_drop_inorm_fwd: y4 [C][N][Ho+1][Wo+1] (conv output, row 0 / column 0 junk) -> out [C][N][Ho][Wo]:
 *                     u = y4 * keep[n][c] (Dropout2d factor, keep may be null); gamma ? InstanceNorm2d(u) * gamma + beta : u
 *                     (biased variance per plane, eps inside the root); stats [C][N][2] = mean, rstd (gamma only).
 * svb_crop_drop_inorm_bwd: dout [N][C][Ho][Wo] by element strides -> dy4 (padded layout, zero border) and the per-plane
 *                     sums dgb [2][N][C] = (sum dout * xhat, sum dout) whose sums over N are dgamma / dbeta.
 *                     s2d != 0 (Ho, Wo even): `out` / `dout` are the NEXT block's conv input / input gradient in its
 *                     space-to-depth layout [4C][N][Ho/2+1][Wo/2+1] (svb_s2d_pad's), so no separate re-layout pass runs
 *                     between two blocks (the strides of dout are ignored then).
 * svb_plane_score_fwd/bwd: score[n] = bias + sum_{c,e} h[n][c][e] * w[c*HW+e] over contiguous (h,w) planes with element
 *                     strides sn / sc; backward reads ds[n * sds] (a column of the stacked window scores) and writes dh
 *                     with the same strides, dw [C*HW], db [1] (each optional).                                          */
int svb_s2_weight(const float* w, float* w4, int cout, int c, void* stream);
int svb_s2_weight_bwd(const float* dwa, const float* dwb, float* dw, int cout, int c, int accumulate, void* stream);
int svb_crop_drop_inorm_fwd(const float* y4, const float* keep, const float* gamma, const float* beta, float eps, float* out,
                            float* stats, int N, int C, int Ho, int Wo, int s2d, void* stream);
int svb_crop_drop_inorm_bwd(const float* dout, long sn, long sc, long sh, long sw, const float* y4, const float* keep,
                            const float* gamma, const float* stats, float* dy4, float* dgb, int N, int C, int Ho, int Wo,
                            int s2d, void* stream);
int svb_plane_score_fwd(const float* h, long sn, long sc, const float* w, const float* bias, float* score, int N, int C, int HW,
                        void* stream);
int svb_plane_score_bwd(const float* ds, long sds, const float* h, long sn, long sc, const float* w, float* dh, float* dw,
                        float* db, int N, int C, int HW, void* stream);

/* ---- SSIM map of two [B, T, F] mel images (+bias), 11x11 gaussian sigma 1.5, zero padding, C1=1e-4, C2=9e-4
 * (reference modules/commons/ssim.py:331-351 via tasks/tts/fs2.py:166-175).  Inputs are addressed with element
 * strides (batch, time, bin) so the decoder's [B,F,T] output is read in place.  out_map/dmap/dpred: contiguous
 * [B,T,F].  Only `pred` receives a gradient.  workspace: 3*B*T*F floats.                                      */
int svb_ssim_fwd(const float* pred, long psb, long pst, long psf, const float* tgt, long tsb, long tst, long tsf,
                 float* out_map, int B, int T, int F, float bias, void* stream);
int svb_ssim_bwd(const float* pred, long psb, long pst, long psf, const float* tgt, long tsb, long tst, long tsf,
                 const float* dmap, float* dpred, float* workspace, int B, int T, int F, float bias, void* stream);

/* ---- Fused mel loss of one way (reference tasks/tts/fs2.py:143-175 l1_loss / ssim_loss with weights_nonzero_speech):
 * w[b,t] = any_f(tgt[b,t,f] != 0);  out[0] = sum|pred-tgt|w / sum w;  out[1] = sum(1-ssim(pred+bias,tgt+bias))w / sum w;
 * out[2] = sum w (sums over all B*T*F pixels).  terms: bit 0 = L1, bit 1 = SSIM (an unselected term's out[] is 0).
 * part: 3*B*ceil(T/16) floats of scratch (per-tile partial sums, reduced in fixed order: deterministic).
 * Backward: gout [2+] = d/d out[0], d/d out[1] (device), sums = the forward's out; dpred contiguous [B,T,F];
 * workspace: 3*B*T*F floats (only with the SSIM term).  Only `pred` receives a gradient.                      */
int svb_mel_loss_fwd(const float* pred, long psb, long pst, long psf, const float* tgt, long tsb, long tst, long tsf,
                     float* out, float* part, int B, int T, int F, float bias, int terms, void* stream);
int svb_mel_loss_bwd(const float* pred, long psb, long pst, long psf, const float* tgt, long tsb, long tst, long tsf,
                     const float* gout, const float* sums, float* dpred, float* workspace, int B, int T, int F, float bias,
                     int terms, void* stream);

/* ---- Latent head of the global VAE encoder (reference modules/voice_conversion/vae_models.py:24-41,100-105):
 * xp [N][2L][Tp] pooled features -> m = mean_t xp[:, :L], lg = mean_t xp[:, L:];  z = m + eps exp(lg)  (eps [N][L]);
 * lq = lg where exp(lg) > 0 else 0 (the positivity guard);  kl[g] = sum_{n in g}(sum_l KL_nl)(sum_t mask[n]) /
 * sum_{n in g} sum_t mask[n] / L with KL_nl = 0.5 (exp(2 lq) + m^2 - 1) - lq, over `groups` equal slices of N (stacked
 * independent calls); mask [N][Tq].  z, mq, lq: [N][L]; kl [groups]; stat [N][2] scratch kept for the backward pass.
 * _bwd: cotangents gz / gm / glq [N][L] and gkl [groups] (each may be NULL = zero) -> dxp [N][2L][Tp].              */
int svb_vae_head_fwd(const float* xp, const float* eps, const float* mask, float* z, float* mq, float* lq, float* kl, float* stat,
                     int N, int groups, int L, int Tp, int Tq, void* stream);
int svb_vae_head_bwd(const float* xp, const float* eps, const float* stat, const float* gz, const float* gm, const float* glq,
                     const float* gkl, float* dxp, int N, int groups, int L, int Tp, void* stream);

/* ---- GroupNorm + ReLU (+ residual) over [B][C][T] (reference modules/commons/common_layers.py:739-773 ConvBlock with
 * norm 'gn', inside ConvStacks :688-707 with res=True):  y = (res ? res : 0) + relu(GroupNorm_G(h) * gamma + beta), biased
 * variance, two-pass statistics; stats [B*G][2] = (mean, rstd).  C/G <= 64.
 * _bwd: gy = d/dy -> dh; dgb [2][B][C] = per-clip partial sums (sum_t g xhat, sum_t g) of dgamma / dbeta, g = gy gated
 * by the ReLU (sum them over B on the host side of the ABI); the residual's gradient is gy itself.                  */
int svb_gn_relu_fwd(const float* h, const float* res, const float* gamma, const float* beta, float* y, float* stats, int B, int C,
                    int T, int G, float eps, void* stream);
int svb_gn_relu_bwd(const float* gy, const float* h, const float* gamma, const float* beta, const float* stats, float* dh,
                    float* dgb, int B, int C, int T, int G, void* stream);

/* ---- STFT magnitude + mel filterbank + log, one kernel (reference data_gen/tts/data_gen_utils.py:123-134 and
 * modules/hifigan/mel_utils.py:45-79).  wav: [B, N].  mode 0 = offline front-end: zero-pad n_fft/2 both sides
 * ("center", pad_mode constant), frames = 1 + N/hop, |X|, log10(max(eps, mel)), out [B, frames, n_mels];
 * mode 1 = in-graph: clamp to [-1,1], reflect-pad (n_fft-hop)/2, frames = N/hop, sqrt(re^2+im^2+1e-9),
 * ln(max(1e-5, mel)), out [B, n_mels, frames].  n_fft must be 512 or 1024 (radix-2 in LDS); window [n_fft];
 * mel_basis [n_mels][n_fft/2+1].                                                                              */
int svb_stft_mel(const float* wav, const float* window, const float* mel_basis, float* out, int B, int N, int n_fft,
                 int hop, int n_mels, int n_frames, int mode, float eps, void* stream);

/* ---- NSF harmonic source (reference modules/parallel_wavegan/models/source.py:44-137,385-398 and
 * modules/hifigan/hifigan.py:145-149): f0 [B, frames] Hz (0 = unvoiced) is held for `upp` samples (nearest upsample);
 * rand_ini [B, H] initial phases (column 0 must be 0); noise [B, frames*upp, H] standard normal;
 * sine_waves out [B, frames*upp, H] (may be NULL), merged out [B, frames*upp] = tanh(sum_h w[h]*sine_h + bias),
 * uv out [B, frames*upp] (may be NULL).  Phase is accumulated in fp32 with the reference's wrap rule.           */
int svb_nsf_source(const float* f0, const float* rand_ini, const float* noise, const float* lin_w, const float* lin_b,
                   float* sine_waves, float* merged, float* uv, int B, int frames, int upp, int H, float sample_rate,
                   float sine_amp, float noise_std, void* stream);

/* ---- pitch-bin quantisation, bit exact (reference utils/pitch_utils.py:130-146).  mode 0: numpy semantics
 * (f64 input, rint half-to-even) ; mode 1: torch semantics (f32 input, (x+0.5) truncation).                     */
int svb_f0_to_coarse_f64(const double* f0, int64_t* out, int64_t n, void* stream);
int svb_f0_to_coarse_f32(const float* f0, int64_t* out, int64_t n, void* stream);

/* ---- the data side of a batch on the GPU (reference utils/__init__.py:118-161 collate_1d/2d,
 * utils/pitch_utils.py:148-177 norm_interp_f0, tasks/tts/dataset_utils.py:133-205).  Ragged input = the rows of all clips
 * concatenated ([sum_len][W]); off[b] / len[b] (int32, device) = first row / row count of clip b.
 * svb_collate_pad_f32: out [B][Tmax][W] = rows of clip b, then `pad`.   svb_collate_pad_i64: the same for int64 scalars per
 * frame, optionally clamped to clip_max[b] (the a2p alignment, svb_vae_task.py:31-33).   svb_mel_energy: energy [B][Tmax] =
 * sqrt(sum_f exp(mel)^2) on real frames, 0 on padding.   svb_norm_interp_f0: f0 (Hz, fp64, 0 = unvoiced) -> normalised,
 * gap-interpolated fp32 track + unvoiced flags, both zero-padded; mode 0 none / 1 log2(f0 + 1e-8) / 2 (f0 - mean) / std;
 * fp64 arithmetic in numpy's order (np.interp's slope form, no fused multiply-add); Tmax <= 6144.                        */
int svb_collate_pad_f32(const float* src, const int* off, const int* len, float* out, int B, int Tmax, int W, float pad,
                        void* stream);
int svb_collate_pad_i64(const int64_t* src, const int* off, const int* len, const int* clip_max, int64_t* out, int B, int Tmax,
                        int64_t pad, void* stream);
int svb_mel_energy(const float* mels, const int* len, float* energy, int B, int Tmax, int W, void* stream);
int svb_norm_interp_f0(const double* src, const int* off, const int* len, float* f0_out, float* uv_out, int B, int Tmax, int mode,
                       double mean, double stdv, int use_uv, void* stream);

/* ---- offline F0 alignment of the binarizer (reference data_gen/singing/binarize_para.py:168-185 ->
 * modules/voice_conversion/dtw/enhance_sadtw.py:18-113 -> modules/voice_conversion/dtw/align.py:8-37), batched over P pairs
 * whose tracks are padded to L / La / Lb frames (len arrays: int32 on the device).
 * svb_f0_shape_hist: f0 [P][L] (Hz, fp64) -> hist [P][L][48] fp32, the normalised slope-class histograms of cal_hist_of_f0
 *                    (max_window 64; scale[p] = the track's scale_factor, 1 for the source, len_tgt/len_src for the target).
 * svb_hist_cost:     cost [P][Lb][La] = chi-square distance of target frame t and source frame s (cal_hist_dist, transposed
 *                    as align_from_distances receives it).
 * svb_dtw_align:     time_warp + back-tracking: dtw [P][Lb][La] (optional) the accumulated costs (bit-exact fp32), dir
 *                    [P][Lb][La] bytes (workspace: arg-min direction per cell), align [P][Lb] int64 = for every target
 *                    frame the matched source frame (0 where the path never passes).  Lb <= 4096.                         */
int svb_f0_shape_hist(const double* f0, const int* len, const double* scale, float* hist, int P, int L, void* stream);
int svb_hist_cost(const float* ha, const int* len_a, const float* hb, const int* len_b, float* cost, int P, int La, int Lb,
                  void* stream);
int svb_dtw_align(const float* cost, const int* len_b, const int* len_a, float* dtw, unsigned char* dir, int64_t* align, int P,
                  int La, int Lb, void* stream);

/* ---- pitch-bin embedding in the conv layout (reference modules/voice_conversion/svb_vae.py:66, nn.Embedding(300, H,
 * padding_idx=0) + transpose): out [B][H][T] = w[idx[b][t]][h]; _bwd: dw [V][H] = sum of dy [B][H][T] over the positions
 * that hold each row (row padding_idx stays zero; accumulate != 0 adds into dw), deterministic summation order;
 * part: workspace of B*V*H floats (per-clip partial sums).                                                              */
int svb_embed_nct_fwd(const int64_t* idx, const float* w, float* out, int B, int H, int T, int V, void* stream);
int svb_embed_nct_bwd(const int64_t* idx, const float* dy, float* part, float* dw, int B, int H, int T, int V, int padding_idx,
                      int accumulate, void* stream);

/* ---- `conv_precision: bf16` (ABI v8): the forward / transposed / tap-table convs of the *_bf16x3 entry points evaluate ONE bf16
 * product per operand pair (hi * hi, fp32 accumulation) instead of the three of the split; weight gradients keep the split.  A
 * process-wide mode (it mirrors an hparam), to be set before work is issued; default off.  BASELINE configs[1] names "bf16": this
 * is that arithmetic, reported as a secondary line only (it is narrower than the reference's fp32).                            */
void svb_conv_set_single_product(int on);
int svb_conv_get_single_product(void);
/* Arithmetic of svb_relpos_attn_fwd / svb_relpos_attn_pos_fwd (ABI v10).  on: operands split three ways into bf16 parts (24 mantissa
 * bits), six products per operand pair, ~2^-23 per product -- the `conv_precision: fp32` mode, whose reference evaluates this
 * attention in fp32 (modules/commons/espnet_transformer_attn.py:125-186); the position table then needs its third part (pt_lo2 of
 * svb_relpos_attn_pos_fwd; NULL: that call runs two-way).  off: hi + lo, three products (2^-16 per product), the bf16x3 / bf16 modes.
 * Process-wide, like the switch above; default ON (the host side's default precision is fp32).                                  */
void svb_attn_set_split3(int on);
int svb_attn_get_split3(void);

/* ---- nearest-neighbour upsampling along time, conv layout (reference modules/voice_conversion/svb_vae.py:39-45:
 * nn.Upsample(scale_factor=s, mode='nearest') of the content features; ABI v8).  adjoint == 0: x [rows][T] -> y [rows][T*scale],
 * y[r][t*scale + j] = x[r][t].  adjoint != 0: x = dy [rows][T*scale] -> y = dx [rows][T], the sum over each window in j order. */
int svb_upsample_nearest_nct(const float* x, float* y, long rows, int T, int scale, int adjoint, void* stream);

/* ---- Multi-period discriminator plumbing (reference modules/hifigan/hifigan.py:171-223: `x.view(b, c, t // period, period)` +
 * Conv2d((k,1), (stride,1)) layers).  Feature maps stay in the reference's [B][C][H][p] layout; a (k,1) conv is a 1-D conv with
 * dilation p over the flattened [H*p] axis, and a stride-s layer runs at stride 1 on the row space-to-depth image
 *   img[plane][r][row][w] = x[plane][s*(row - lead) + r][w]   (zero outside the plane; `lead` zero rows in front, R rows in all)
 * (inverse != 0: the gather back, x[plane][h][w] = img[plane][h % s][lead + h / s][w] -- its gradient).  planes = B*C.        */
int svb_period_s2d(const float* src, float* dst, long planes, int H, int p, int s, int lead, int R, int inverse, void* stream);
/* The weight of that stride-1 conv: w2[co][ci*s + r][q] = v[co][ci][s*q + r + front] (0 outside [0,k); front <= 0, s*taps + front >= k).
 * inverse = 0: w2 from v ([cout][cin][k] -> [cout][cin*s][taps]); inverse = 1: the gradient of v gathered from the gradient of w2,
 * added to dst when `accumulate`.                                                                                      */
int svb_period_weight(const float* src, float* dst, int cout, int cin, int k, int s, int taps, int front, int inverse,
                      int accumulate, void* stream);

/* ---- gradient clipping + AdamW of one optimizer over flat buffers (reference tasks/singing/svb_vae_task.py:84-118 torch.optim.AdamW,
 * :390-404 clip_grad_norm_): p, g, m, v are flat fp32 arrays of n elements (n % 4 == 0; parameters, gradients, first and second
 * moments at the same offsets).  coef = min(1, max_norm / (||g|| + 1e-6)) (max_norm <= 0: no clipping); g' = coef g;
 * p *= 1 - lr wd;  m = lerp(m, g', 1 - beta1);  v = beta2 v + (1 - beta2) g'^2;
 * p -= (lr / bias_correction1) m / (sqrt(v) / bias_correction2_sqrt + eps).  workspace: svb_adamw_flat_workspace_floats()
 * floats (needed when max_norm > 0 or norm_out != NULL); norm_out: optional, receives ||g||.  Two launches.                    */
int svb_adamw_flat_workspace_floats(void);
int svb_adamw_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float bias_correction1, float bias_correction2_sqrt, float max_norm, float* workspace,
                   float* norm_out, void* stream);

/* Copies `count` fp32 arrays (src[i], n[i] elements) to dst + dst_off[i]: the gradients autograd handed over as tensors of their
 * own (a parameter's `.grad` was None, reference semantics of optimizer.zero_grad()) into their slices of the flat gradient
 * buffer before svb_adamw_flat, one launch per 48 arrays.  src / dst_off / n are HOST arrays.                                   */
int svb_gather_segments(const void* const* src, const size_t* dst_off, const size_t* n, int count, float* dst, void* stream);

/* nn.BatchNorm1d on x [B,C,T] (reference: the `poolings` of modules/voice_conversion/vae_models.py -- train mode -- and the PPG
 * pre-net modules/voice_conversion/pe.py:23-41 -- eval mode, followed by `* nonpadding`).
 * training != 0: the batch is `groups` equal slices along dim 0, each normalised with its own biased batch statistics (what
 *   `groups` separate calls of the module compute); running_mean / running_var (NULL: not tracked) receive the groups' momentum
 *   updates in order (unbiased variance), *num_batches (NULL: none) += groups; save [2][groups][C] receives mean and rstd for
 *   the backward.  mask must be NULL.
 * training == 0: y = (x - running_mean) rsqrt(running_var + eps) gamma + beta, times mask[b,t] when mask != NULL; groups == 1.
 * gamma / beta NULL: affine=False.                                                                                             */
int svb_batchnorm_nct_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          long long* num_batches, const float* mask, float* y, float* save, int B, int C, int T, int groups,
                          int training, float momentum, float eps, void* stream);
/* Backward of the train-mode form: dx (NULL: skip), dgamma / dbeta [C] (written, summed over the groups; NULL: skip).           */
int svb_batchnorm_nct_bwd(const float* dy, const float* x, const float* gamma, const float* save, float* dx, float* dgamma,
                          float* dbeta, int B, int C, int T, int groups, void* stream);

/* torch.nn.utils.spectral_norm of a conv weight (reference modules/hifigan/hifigan.py:238-250, DiscriminatorS with
 * use_spectral_norm=True): w [R][C] = weight_orig.flatten(1), u [R], v [C] the module's buffers.
 * training != 0: one power iteration -- v = normalize(w^T u), u = normalize(w v), eps-clamped norms -- written to u / v in
 *   place; sigma = u . (w v); w_sn = w / sigma.  training == 0: sigma from the stored u, v; nothing written back.
 * u_save / v_save receive the vectors the graph treats as constants (the backward reads them), sigma_out [1] the scalar.
 * workspace: svb_spectral_norm_workspace_floats(R, C) floats.  3 launches (2 in eval mode).                                     */
size_t svb_spectral_norm_workspace_floats(int R, int C);
int svb_spectral_norm_fwd(const float* w, float* u, float* v, float* w_sn, float* u_save, float* v_save, float* sigma_out, int R,
                          int C, int training, float eps, float* workspace, void* stream);
/* dw = dw_sn / sigma - (sum dw_sn * w) / sigma^2 * u v^T  (u, v, sigma as saved by the forward).  2 launches.                  */
int svb_spectral_norm_bwd(const float* dw_sn, const float* w, const float* u, const float* v, const float* sigma, float* dw, int R,
                          int C, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVB_HIP_H */
