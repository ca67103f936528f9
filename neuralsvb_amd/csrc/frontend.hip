// frontend.hip -- HBM-bound front-end kernels of the NeuralSVB hot path (gfx950).
//   svb_stft_mel    : framing + window + radix-2 FFT in LDS + |X| + mel filterbank + log, one pass over the wave
//                     (reference data_gen/tts/data_gen_utils.py:123-134 offline; modules/hifigan/mel_utils.py:59-76 in-graph)
//   svb_nsf_source  : NSF harmonic excitation (reference modules/parallel_wavegan/models/source.py:44-137,385-398)
//   svb_f0_to_coarse: pitch-bin quantisation (reference utils/pitch_utils.py:130-146)
// Algorithmic bytes: STFT/mel 4 B/sample in + n_mels*4 B per hop out (6.5 B/sample at hop 128, 80 mels);
// NSF 4 B/frame in + 36 B/sample noise in + 4..44 B/sample out.  One 64-lane wave owns one frame: the FFT's
// butterflies, the magnitude and the (sparse, triangular) mel contraction never leave LDS.
#include "svb_common.h"
#include "../../include/svb_hip.h"

__device__ __forceinline__ unsigned svb_bitrev(unsigned v, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

template <int LOG2N>
__global__ __launch_bounds__(256) void svb_stft_mel_kernel(const float* wav, const float* window, const float* basis,
                                                           float* out, int B, int N, int hop, int n_mels, int n_frames,
                                                           int mode, float eps) {
    constexpr int NF = 1 << LOG2N, NB = NF / 2 + 1, PER = NF / 64;
    __shared__ float re[4][NF];
    __shared__ float im[4][NF];
    __shared__ float mag[4][NB + 3];
    __shared__ float tw_re[NF / 2];
    __shared__ float tw_im[NF / 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (int k = threadIdx.x; k < NF / 2; k += 256) {
        // exp(-2 pi i k / NF); evaluated in double so the table is correctly rounded fp32
        const double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)NF;
        tw_re[k] = (float)cos(ang);
        tw_im[k] = (float)sin(ang);
    }
    // non-zero support of the (triangular) mel filters owned by this lane: filters lane and lane+64
    int lo[2] = {0, 0}, hi[2] = {0, 0};
    for (int s = 0; s < 2; ++s) {
        const int m = lane + 64 * s;
        if (m < n_mels) {
            int first = NB, last = -1;
            for (int k = 0; k < NB; ++k)
                if (basis[(size_t)m * NB + k] != 0.f) { if (first == NB) first = k; last = k; }
            lo[s] = first; hi[s] = last + 1;
        }
    }
    const int pad = mode == 0 ? NF / 2 : (NF - hop) / 2;
    const float mag_eps = mode == 0 ? 0.f : 1e-9f;
    const long total = (long)B * n_frames;
    const long per_iter = (long)gridDim.x * 4;
    const long iters = (total + per_iter - 1) / per_iter;
    __syncthreads();

    for (long it = 0; it < iters; ++it) {
        const long fr = it * per_iter + (long)blockIdx.x * 4 + wave;
        const bool active = fr < total;
        const int b = active ? (int)(fr / n_frames) : 0;
        const int f = active ? (int)(fr - (long)b * n_frames) : 0;
        if (active) {
            const float* wv = wav + (size_t)b * N;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int n = lane + 64 * i;
                int s = f * hop + n - pad;
                float v;
                if (mode == 0) {
                    v = (s >= 0 && s < N) ? wv[s] : 0.f;
                } else {
                    if (s < 0) s = -s;
                    if (s >= N) s = 2 * (N - 1) - s;
                    v = fminf(1.f, fmaxf(-1.f, wv[s]));
                }
                const unsigned d = svb_bitrev((unsigned)n, LOG2N);
                re[wave][d] = v * window[n];
                im[wave][d] = 0.f;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int st = 1; st <= LOG2N; ++st) {
            const int half = 1 << (st - 1);
            const int tstep = NF >> st;
            if (active) {
#pragma unroll
                for (int i = 0; i < PER / 2; ++i) {
                    const int j = lane + 64 * i;
                    const int pos = j & (half - 1);
                    const int i0 = ((j >> (st - 1)) << st) + pos;
                    const int i1 = i0 + half;
                    const float wr = tw_re[pos * tstep], wi = tw_im[pos * tstep];
                    const float xr = re[wave][i1], xi = im[wave][i1];
                    const float tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
                    const float ar = re[wave][i0], ai = im[wave][i0];
                    re[wave][i0] = ar + tr; im[wave][i0] = ai + ti;
                    re[wave][i1] = ar - tr; im[wave][i1] = ai - ti;
                }
            }
            __syncthreads();
        }
        if (active) {
            for (int k = lane; k < NB; k += 64) {
                const float r = re[wave][k], q = im[wave][k];
                mag[wave][k] = sqrtf(r * r + q * q + mag_eps);
            }
        }
        __syncthreads();
        if (active) {
            for (int s = 0; s < 2; ++s) {
                const int m = lane + 64 * s;
                if (m < n_mels) {
                    float acc = 0.f;
                    for (int k = lo[s]; k < hi[s]; ++k) acc = fmaf(basis[(size_t)m * NB + k], mag[wave][k], acc);
                    if (mode == 0)
                        out[((size_t)b * n_frames + f) * n_mels + m] = log10f(fmaxf(eps, acc));
                    else
                        out[((size_t)b * n_mels + m) * n_frames + f] = logf(fmaxf(eps, acc));
                }
            }
        }
        __syncthreads();
    }
}

extern "C" int svb_stft_mel(const float* wav, const float* window, const float* mel_basis, float* out, int B, int N,
                            int n_fft, int hop, int n_mels, int n_frames, int mode, float eps, void* stream) {
    if (!wav || !window || !mel_basis || !out || B <= 0 || N <= 0 || hop <= 0 || n_mels <= 0 || n_mels > 128 ||
        n_frames <= 0 || (mode != 0 && mode != 1))
        return SVB_ERR_ARG;
    if (mode == 0 && n_frames != 1 + N / hop) return SVB_ERR_ARG;
    if (mode == 1 && (n_frames != N / hop || (n_fft - hop) / 2 >= N)) return SVB_ERR_ARG;
    const long total = (long)B * n_frames;
    long grid = (total + 3) / 4;
    if (grid > 2048) grid = 2048;
    if (n_fft == 512)
        hipLaunchKernelGGL(svb_stft_mel_kernel<9>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, wav, window,
                           mel_basis, out, B, N, hop, n_mels, n_frames, mode, eps);
    else if (n_fft == 1024)
        hipLaunchKernelGGL(svb_stft_mel_kernel<10>, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, wav, window,
                           mel_basis, out, B, N, hop, n_mels, n_frames, mode, eps);
    else
        return SVB_ERR_UNSUPPORTED;
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ------------------------------------------------------------------------------------------------------------
// NSF source.  f0 is piecewise constant per frame (nearest upsample by `upp`), so the running phase is
//   phase(frame f, sample n) = frac(P_f + (n+1) * rad_f),   P_f = frac(rand_ini + sum_{f'<f} upp * rad_f')
// with rad = (f0*(h+1)/sr) mod 1 rounded to fp32 exactly as the reference does.  Phases are carried in fp64, which
// equals the reference's wrap-corrected fp32 cumsum up to the reference's own rounding (integer wraps do not
// change sin(2 pi x)).
// ------------------------------------------------------------------------------------------------------------
#define SVB_NSF_FC 16
#define SVB_NSF_MAXH 16

__device__ __forceinline__ float svb_nsf_rad(float f0, int h, float sr) {
    const float fh = h == 0 ? f0 : f0 * (float)(h + 1);
    return fmodf(fh / sr, 1.0f);
}

__global__ __launch_bounds__(256) void svb_nsf_source_kernel(const float* f0, const float* rand_ini, const float* noise,
                                                             const float* lin_w, const float* lin_b, float* sine_waves,
                                                             float* merged, float* uv, int B, int frames, int upp, int H,
                                                             float sr, float sine_amp, float noise_std) {
    __shared__ double pstart[SVB_NSF_MAXH];
    __shared__ double pf[SVB_NSF_FC][SVB_NSF_MAXH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int fbeg = blockIdx.x * SVB_NSF_FC;
    const int fend = min(frames, fbeg + SVB_NSF_FC);
    const float* f0b = f0 + (size_t)b * frames;

    for (int h = wave; h < H; h += 4) {
        double acc = 0.0;
        for (int f = lane; f < fbeg; f += 64) {
            acc += (double)upp * (double)svb_nsf_rad(f0b[f], h, sr);
            acc -= floor(acc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        acc += (double)rand_ini[(size_t)b * H + h];
        acc -= floor(acc);
        if (lane == 0) pstart[h] = acc;
    }
    __syncthreads();
    if (threadIdx.x < H) {
        const int h = threadIdx.x;
        double p = pstart[h];
        for (int f = fbeg; f < fend; ++f) {
            pf[f - fbeg][h] = p;
            p += (double)upp * (double)svb_nsf_rad(f0b[f], h, sr);
            p -= floor(p);
        }
    }
    __syncthreads();
    const int nsamp = (fend - fbeg) * upp;
    const size_t L = (size_t)frames * upp;
    const float two_pi = 2.f * 3.14159274101257324f;
    for (int i = threadIdx.x; i < nsamp; i += 256) {
        const int fl = i / upp, n = i - fl * upp;
        const float f0v = f0b[fbeg + fl];
        const float uvv = f0v > 0.f ? 1.f : 0.f;
        const float namp = uvv * noise_std + (1.f - uvv) * sine_amp / 3.f;
        const size_t s = (size_t)b * L + (size_t)(fbeg + fl) * upp + n;
        float accm = lin_b ? lin_b[0] : 0.f;
        for (int h = 0; h < H; ++h) {
            const double rad = (double)svb_nsf_rad(f0v, h, sr);
            double ph = pf[fl][h] + (double)(n + 1) * rad;
            ph -= floor(ph);
            const float sine = sinf((float)ph * two_pi) * sine_amp;
            const float sw = sine * uvv + namp * noise[s * H + h];
            if (sine_waves) sine_waves[s * H + h] = sw;
            accm = fmaf(lin_w ? lin_w[h] : 0.f, sw, accm);
        }
        if (merged) merged[s] = tanhf(accm);
        if (uv) uv[s] = uvv;
    }
}

extern "C" int svb_nsf_source(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                              const float* lin_b, float* sine_waves, float* merged, float* uv, int B, int frames, int upp,
                              int H, float sample_rate, float sine_amp, float noise_std, void* stream) {
    if (!f0 || !rand_ini || !noise || B <= 0 || frames <= 0 || upp <= 0 || H <= 0 || H > SVB_NSF_MAXH || B > 65535 ||
        (!sine_waves && !merged && !uv))
        return SVB_ERR_ARG;
    if (merged && !lin_w) return SVB_ERR_ARG;
    dim3 grid(svb_cdiv(frames, SVB_NSF_FC), B);
    hipLaunchKernelGGL(svb_nsf_source_kernel, grid, dim3(256), 0, (hipStream_t)stream, f0, rand_ini, noise, lin_w, lin_b,
                       sine_waves, merged, uv, B, frames, upp, H, sample_rate, sine_amp, noise_std);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ------------------------------------------------------------------------------------------------------------
// pitch bins (1..255), reference utils/pitch_utils.py:130-146
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void svb_f0_to_coarse_f64_kernel(const double* f0, int64_t* out, int64_t n, double mel_min,
                                                                   double mel_max) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double m = 1127.0 * log(1.0 + f0[i] / 700.0);
        if (m > 0.0) m = (m - mel_min) * 254.0 / (mel_max - mel_min) + 1.0;
        if (m <= 1.0) m = 1.0;
        if (m > 255.0) m = 255.0;
        out[i] = (int64_t)rint(m);  // half-to-even == np.rint
    }
}
__global__ __launch_bounds__(256) void svb_f0_to_coarse_f32_kernel(const float* f0, int64_t* out, int64_t n, float mel_min,
                                                                   float scale_num, float mel_range) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // correctly rounded fp32 log (evaluated in double): independent of the device's logf, equals the CPU reference's
        float m = 1127.f * (float)log((double)(1.f + f0[i] / 700.f));
        if (m > 0.f) m = (m - mel_min) * scale_num / mel_range + 1.f;
        if (m <= 1.f) m = 1.f;
        if (m > 255.f) m = 255.f;
        out[i] = (int64_t)(m + 0.5f);  // torch: (f0_mel + 0.5).long()
    }
}

extern "C" int svb_f0_to_coarse_f64(const double* f0, int64_t* out, int64_t n, void* stream) {
    if (!f0 || !out || n <= 0) return SVB_ERR_ARG;
    const double mel_min = 1127.0 * log(1.0 + 50.0 / 700.0), mel_max = 1127.0 * log(1.0 + 1100.0 / 700.0);
    int64_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(svb_f0_to_coarse_f64_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, f0, out, n, mel_min,
                       mel_max);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}
extern "C" int svb_f0_to_coarse_f32(const float* f0, int64_t* out, int64_t n, void* stream) {
    if (!f0 || !out || n <= 0) return SVB_ERR_ARG;
    // the reference's module-level constants are float64 numpy scalars; torch computes
    // (f32 tensor - f64 scalar) * int / f64 scalar in fp32 with the scalars rounded to fp32.
    const double mel_min = 1127.0 * log(1.0 + 50.0 / 700.0), mel_max = 1127.0 * log(1.0 + 1100.0 / 700.0);
    int64_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(svb_f0_to_coarse_f32_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, f0, out, n,
                       (float)mel_min, 254.f, (float)(mel_max - mel_min));
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Pitch-bin embedding in the conv layout (reference modules/voice_conversion/svb_vae.py:66 `self.pitch_embed(pitch)`,
// nn.Embedding(300, H, padding_idx=0), followed there by a transpose for the Conv1d stack):
//   out[b][h][t] = W[idx[b][t]][h]                         one gather that writes [B,H,T] directly (lanes along t)
//   dW[v][h]     = sum over (b,t) with idx[b][t] == v of dy[b][h][t], v != padding_idx (row padding_idx stays 0)
// The gradient is DETERMINISTIC (torch sorts the indices for that; an atomic scatter would not be): one workgroup per
// (vocabulary row, clip) scans the clip's indices in t order -- a coalesced 64-index read + ballot per wave step -- and thread h
// adds dy[b][h][t] of every match to its register; a second kernel adds the per-clip partial sums [B][V][H] in clip order.
// ------------------------------------------------------------------------------------------------------------------
// grid (T / 256, channel slabs, B): a thread owns one position t of one clip, reads its index once and walks the slab's channels --
// out rows are written coalesced along t, the embedding row w[v][.] is read along h (L1 / L2 resident: V x H floats), and there is
// no index arithmetic per element (the flat form's two 64-bit divisions per element made this 37 MB write take 99 us, round 4).
#define SVB_EMBED_HSLAB 32
__global__ __launch_bounds__(256) void svb_embed_nct_fwd_kernel(const int64_t* idx, const float* w, float* out, int B, int H, int T,
                                                                int V) {
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z;
    if (t >= T) return;
    const int h0 = blockIdx.y * SVB_EMBED_HSLAB, h1 = h0 + SVB_EMBED_HSLAB < H ? h0 + SVB_EMBED_HSLAB : H;
    const int64_t v = idx[(long)b * T + t];
    const bool ok = v >= 0 && v < V;
    const float* wr = w + (ok ? v : 0) * H;
    float* o = out + ((long)b * H + h0) * T + t;
    for (int h = h0; h < h1; ++h, o += T) *o = ok ? wr[h] : 0.f;
}

// part[b][v][h] = sum over t with idx[b][t] == v of dy[b][h][t], in t order.  grid (V, B): a (row, clip) pair without a match
// costs one scan of the clip's indices.  The loads of up to four matches are in flight together (the loop is latency-bound:
// every match is one scattered 4-byte read per thread); they are added in t order, so the result does not depend on timing.
__global__ __launch_bounds__(256) void svb_embed_nct_bwd_part_kernel(const int64_t* idx, const float* dy, float* part, int H, int T,
                                                                     int V, int padding_idx) {
    const int v = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int64_t* ib = idx + (long)b * T;
    for (int h0 = 0; h0 < H; h0 += 256) {
        const int h = h0 + threadIdx.x;
        const float* dyb = dy + ((long)b * H + (h < H ? h : 0)) * T;
        float acc = 0.f;
        if (v != padding_idx) {
            for (int t0 = 0; t0 < T; t0 += 64) {
                const int t = t0 + lane;
                unsigned long long m = __ballot(t < T && ib[t] == (int64_t)v);
                while (m) {
                    float x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j = m ? __ffsll((long long)m) - 1 : -1;
                        m = m ? (m & (m - 1)) : 0ull;
                        x[q] = j >= 0 ? dyb[t0 + j] : 0.f;
                    }
                    acc += x[0];
                    acc += x[1];
                    acc += x[2];
                    acc += x[3];
                }
            }
        }
        if (h < H) part[((long)b * V + v) * H + h] = acc;
    }
}

__global__ __launch_bounds__(256) void svb_embed_nct_bwd_sum_kernel(const float* part, float* dw, int B, long VH, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= VH) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += part[(long)b * VH + i];
    dw[i] = accumulate ? dw[i] + acc : acc;
}

extern "C" int svb_embed_nct_fwd(const int64_t* idx, const float* w, float* out, int B, int H, int T, int V, void* stream) {
    if (!idx || !w || !out || B <= 0 || H <= 0 || T <= 0 || V <= 0) return SVB_ERR_ARG;
    if (B > 65535 || (H + SVB_EMBED_HSLAB - 1) / SVB_EMBED_HSLAB > 65535) return SVB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(svb_embed_nct_fwd_kernel, dim3((T + 255) / 256, (H + SVB_EMBED_HSLAB - 1) / SVB_EMBED_HSLAB, B), dim3(256), 0,
                       (hipStream_t)stream, idx, w, out, B, H, T, V);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

extern "C" int svb_embed_nct_bwd(const int64_t* idx, const float* dy, float* part, float* dw, int B, int H, int T, int V,
                                 int padding_idx, int accumulate, void* stream) {
    if (!idx || !dy || !part || !dw || B <= 0 || B > 65535 || H <= 0 || T <= 0 || V <= 0) return SVB_ERR_ARG;
    hipLaunchKernelGGL(svb_embed_nct_bwd_part_kernel, dim3(V, B), dim3(256), 0, (hipStream_t)stream, idx, dy, part, H, T, V,
                       padding_idx);
    SVB_CHECK_LAUNCH();
    const long VH = (long)V * H;
    hipLaunchKernelGGL(svb_embed_nct_bwd_sum_kernel, dim3((unsigned)((VH + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                       dw, B, VH, accumulate);
    SVB_CHECK_LAUNCH();
    return SVB_OK;
}

// ---- nearest-neighbour upsampling along time in the conv layout (reference modules/voice_conversion/svb_vae.py:39-45:
// nn.Upsample(scale_factor=s, mode='nearest') in front of the content features' smoothing conv) ---------------------------------
//   y[r][t*s + j] = x[r][t]   (rows = B*C, 0 <= j < s);   adjoint: dx[r][t] = sum_j dy[r][t*s + j], added in j order.
// HBM-bound: 4 B read + 4*s B written per input element.  grid (ceil(T*s / 1024), rows): a thread writes one float4 of the
// output row (coalesced 16-byte stores along t); the row base is per block, so there is no 64-bit division per element.
__global__ __launch_bounds__(256) void svb_upsample_nearest_kernel(const float* x, float* y, int T, int s) {
    const long row = blockIdx.y;
    const int To = T * s;
    const int q = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (q >= To) return;
    const float* xr = x + row * T;
    float* yr = y + row * To;
    if (q + 3 < To && (To & 3) == 0) {
        float4 o;
        o.x = xr[q / s]; o.y = xr[(q + 1) / s]; o.z = xr[(q + 2) / s]; o.w = xr[(q + 3) / s];
        *reinterpret_cast<float4*>(yr + q) = o;
    } else {
        for (int e = q; e < To && e < q + 4; ++e) yr[e] = xr[e / s];
    }
}

__global__ __launch_bounds__(256) void svb_upsample_nearest_bwd_kernel(const float* dy, float* dx, int T, int s) {
    const long row = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const float* g = dy + row * (long)T * s + (long)t * s;
    float acc = 0.f;
    for (int j = 0; j < s; ++j) acc += g[j];
    dx[row * T + t] = acc;
}

extern "C" int svb_upsample_nearest_nct(const float* x, float* y, long rows, int T, int scale, int adjoint, void* stream) {
    if (!x || !y || rows <= 0 || T <= 0 || scale <= 0) return SVB_ERR_ARG;
    if (rows > 65535L * 32768L || (long)T * scale > 0x7fffffffL) return SVB_ERR_UNSUPPORTED;
    // rows ride on gridDim.y (<= 65535): larger row counts go out in slabs
    for (long r0 = 0; r0 < rows; r0 += 65535) {
        const unsigned nr = (unsigned)(rows - r0 < 65535 ? rows - r0 : 65535);
        if (!adjoint) {
            const int To = T * scale;
            hipLaunchKernelGGL(svb_upsample_nearest_kernel, dim3((To + 1023) / 1024, nr), dim3(256), 0, (hipStream_t)stream,
                               x + r0 * T, y + r0 * To, T, scale);
        } else {
            // x = dy [rows][T*scale], y = dx [rows][T]
            hipLaunchKernelGGL(svb_upsample_nearest_bwd_kernel, dim3((T + 255) / 256, nr), dim3(256), 0, (hipStream_t)stream,
                               x + r0 * (long)T * scale, y + r0 * T, T, scale);
        }
        SVB_CHECK_LAUNCH();
    }
    return SVB_OK;
}

