// Host-glue arithmetic of the reference moved onto the device: PCM_16 round trip, chunk slice/pad,
// Hann WOLA stitch (as a deterministic gather) and the STFT-magnitude used by the LSD yardstick.
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "egr_common.h"
#include "egr_fft_device.h"
#include "egr_plan.h"

namespace egr {

__global__ __launch_bounds__(256) void k_pcm16(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                float ws, float rd) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        long long q = (long long)rintf(__fmul_rn(x[i], ws));
        q = ((q + 32768) & 65535) - 32768;
        y[i] = __fdiv_rn((float)q, rd);
    }
}

__global__ __launch_bounds__(256) void k_chunk_gather(const float* __restrict__ x, int C, long long total,
                                                       long long win, long long hop, int chunk_begin,
                                                       float* __restrict__ chunks) {
    const int k = blockIdx.y / C, c = blockIdx.y % C;
    const long long start = (long long)(chunk_begin + k) * hop;
    const float* xc = x + (size_t)c * total;
    float* dst = chunks + ((size_t)k * C + c) * win;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < win;
         j += (long long)gridDim.x * blockDim.x) {
        const long long i = start + j;
        dst[j] = (i < total) ? xc[i] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_wola(const float* __restrict__ preds, int nchunks, int C, long long lp,
                                               long long total, long long win, long long hop,
                                               const float* __restrict__ window, float* __restrict__ out) {
    const int c = blockIdx.y;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long k_lo = (i < win) ? 0 : (i - win) / hop + 1;
        long long k_hi = i / hop;
        if (k_hi > nchunks - 1) k_hi = nchunks - 1;
        float acc = 0.f, ws = 0.f;
        for (long long k = k_lo; k <= k_hi; ++k) {
            const long long start = k * hop;
            long long L = total - start;
            if (L > win) L = win;
            if (L > lp) L = lp;
            const long long j = i - start;
            if (j < L) {
                const float w = window[j];
                const float y = preds[((size_t)k * C + c) * lp + j];
                acc = __fadd_rn(acc, __fmul_rn(y, w));
                ws = __fadd_rn(ws, w);
            }
        }
        if (ws == 0.f) ws = 1.f;
        out[(size_t)c * total + i] = __fdiv_rn(acc, ws);
    }
}

// Rational-rate polyphase FIR, same definition and float32 accumulation order as scipy.signal.resample_poly
// (upfirdn with zero extension): y[m] = sum_k h[m*down - k*up + half] * x[k], k ascending, unfused mul/add.
__global__ __launch_bounds__(256) void k_resample_poly(const float* __restrict__ x, long long n_in, long long n_out,
                                                        int up, int down, const float* __restrict__ h, int half,
                                                        float* __restrict__ y) {
    const int c = blockIdx.y;
    const float* xc = x + (size_t)c * n_in;
    float* yc = y + (size_t)c * n_out;
    const long long hl = 2LL * half;
    for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < n_out;
         m += (long long)gridDim.x * blockDim.x) {
        const long long t = m * down + half;
        long long k_hi = t / up;
        if (k_hi > n_in - 1) k_hi = n_in - 1;
        long long k_lo = t - hl <= 0 ? 0 : (t - hl + up - 1) / up;
        float acc = 0.f;
        for (long long k = k_lo; k <= k_hi; ++k) acc = __fadd_rn(acc, __fmul_rn(xc[k], h[t - k * up]));
        yc[m] = acc;
    }
}

// One workgroup per frame: mono downmix, window, half-length complex FFT in LDS, real split, |X|.
__global__ __launch_bounds__(256) void k_stft_mag(const float* __restrict__ x, int C, long long n, int n_fft, int hop,
                                                   const float* __restrict__ window, FftDesc fd,
                                                   const cplx* __restrict__ tw, const cplx* __restrict__ wsplit,
                                                   float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Mh = n_fft / 2;
    cplx* cur = (cplx*)smem;
    cplx* alt = cur + Mh;
    const long long f = blockIdx.x;
    const long long s0 = f * hop;
    const float invC = (float)C;
    for (int e = threadIdx.x; e < Mh; e += blockDim.x) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long t = s0 + 2 * e + h;
            float m = 0.f;
            if (t < n) {
                m = x[t];
                if (C > 1) {
                    for (int c = 1; c < C; ++c) m = __fadd_rn(m, x[(size_t)c * n + t]);
                    m = __fdiv_rn(m, invC);
                }
            }
            v[h] = __fmul_rn(m, window[2 * e + h]);
        }
        cur[e] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    lds_fft<false>(cur, alt, fd, tw, 1, 0, 1, Mh, false);
    float* o = out + (size_t)f * (Mh + 1);
    for (int k = threadIdx.x; k <= Mh; k += blockDim.x) {
        const cplx Za = cur[k == Mh ? 0 : k];
        const cplx Zb = cur[(k == 0 || k == Mh) ? 0 : Mh - k];
        const cplx E = make_float2(0.5f * (Za.x + Zb.x), 0.5f * (Za.y - Zb.y));
        const cplx O = make_float2(0.5f * (Za.y + Zb.y), -0.5f * (Za.x - Zb.x));
        const cplx X = cadd(E, cmul(wsplit[k], O));
        o[k] = sqrtf(X.x * X.x + X.y * X.y);
    }
}

struct StftTables {
    FftDesc fd;
    cplx *tw, *wsplit;
};
static std::mutex g_stft_mu;
static std::map<std::pair<int, int>, StftTables> g_stft;   // (device, n_fft)

static int stft_tables(int n_fft, StftTables* out) {
    int dev = 0;
    EGR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_stft_mu);
    auto it = g_stft.find({dev, n_fft});
    if (it != g_stft.end()) { *out = it->second; return EGR_OK; }
    StftTables t;
    const int Mh = n_fft / 2;
    EGR_CHECK(make_schedule(Mh, &t.fd, 127), EGR_ERR_UNSUPPORTED, "n_fft=%d: n_fft/2 has a prime factor above 127", n_fft);
    std::vector<float2> h;
    make_twiddles(h, Mh, 1, Mh);
    EGR_HIP(hipMalloc((void**)&t.tw, h.size() * sizeof(float2)));
    EGR_HIP(hipMemcpy(t.tw, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    make_twiddles(h, Mh + 1, 1, n_fft);
    EGR_HIP(hipMalloc((void**)&t.wsplit, h.size() * sizeof(float2)));
    EGR_HIP(hipMemcpy(t.wsplit, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    g_stft[{dev, n_fft}] = t;
    *out = t;
    return EGR_OK;
}

// ---------------------------------------------------------------- evaluation metrics (egregora_audio_eval_pack.py:405-429)
// per[f] = sqrt(mean_k (20 log10(A[f][k] + eps) - 20 log10(B[f][k] + eps))^2 + 1e-12), float32 terms like the reference,
// the bin sum in double.  One workgroup per frame.
__global__ __launch_bounds__(256) void k_lsd_frames(const float* __restrict__ SA, const float* __restrict__ SB, int nb,
                                                     float* __restrict__ per) {
    __shared__ double red[256];
    const size_t f = blockIdx.x;
    const float* a = SA + f * nb;
    const float* b = SB + f * nb;
    double acc = 0.0;
    for (int k = threadIdx.x; k < nb; k += 256) {
        const float la = 20.0f * log10f(a[k] + 1e-12f), lb = 20.0f * log10f(b[k] + 1e-12f);
        const float d = la - lb;
        acc += (double)(d * d);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) per[f] = sqrtf((float)(red[0] / nb) + 1e-12f);
}

__global__ __launch_bounds__(256) void k_sum_f64(const float* __restrict__ v, long long n, double* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += (double)v[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, red[0]);
}

// k-th smallest (0-based) of v[0..n) for two ranks, by 4-pass radix selection on order-preserving keys.  ONE workgroup.
__device__ __forceinline__ unsigned f2key(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__global__ __launch_bounds__(1024) void k_order_stats2(const float* __restrict__ v, long long n, long long k0, long long k1,
                                                        float* __restrict__ out2) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix;
    __shared__ long long s_k;
    for (int which = 0; which < 2; ++which) {
        if (threadIdx.x == 0) { s_prefix = 0u; s_k = which ? k1 : k0; }
        __syncthreads();
        for (int pass = 3; pass >= 0; --pass) {
            if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
            __syncthreads();
            const unsigned prefix = s_prefix;
            const unsigned himask = pass == 3 ? 0u : (0xffffffffu << (8 * (pass + 1)));
            for (long long i = threadIdx.x; i < n; i += 1024) {
                const unsigned key = f2key(v[i]);
                if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                long long k = s_k;
                int bkt = 0;
                for (; bkt < 255; ++bkt) {
                    if (k < (long long)hist[bkt]) break;
                    k -= hist[bkt];
                }
                s_k = k;
                s_prefix = prefix | ((unsigned)bkt << (8 * pass));
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) out2[which] = key2f(s_prefix);
        __syncthreads();
    }
}

// SI-SDR terms on the mono downmixes (mean over channels in float32, like the reference's .mean(axis=0); sums in double):
//   mode 0: out[0] += <s_hat, s>, out[1] += <s, s>
//   mode 1: alpha = out[0] / (out[1] + 1e-20); out[2] += |alpha s|^2, out[3] += |s_hat - alpha s|^2
__global__ __launch_bounds__(256) void k_sisdr(const float* __restrict__ s, int cs, long long stride_s,
                                                const float* __restrict__ sh, int csh, long long stride_sh, long long n,
                                                int mode, double* __restrict__ out) {
    __shared__ double r0[256], r1[256];
    const double alpha = mode ? out[0] / (out[1] + 1e-20) : 0.0;
    double a0 = 0.0, a1 = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float ms = 0.f, mh = 0.f;
        for (int c = 0; c < cs; ++c) ms += s[c * stride_s + i];
        for (int c = 0; c < csh; ++c) mh += sh[c * stride_sh + i];
        const double x = (double)(ms / (float)cs), y = (double)(mh / (float)csh);
        if (mode == 0) {
            a0 += y * x;
            a1 += x * x;
        } else {
            const double t = alpha * x, e = y - t;
            a0 += t * t;
            a1 += e * e;
        }
    }
    r0[threadIdx.x] = a0;
    r1[threadIdx.x] = a1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(out + 2 * mode, r0[0]);
        atomicAdd(out + 2 * mode + 1, r1[0]);
    }
}

// np.interp(linspace(0,1,n_out,False), linspace(0,1,n_in,False), x) per row (Resample Audio (HQ) "linear" mode,
// egregora_audio_eval_pack.py:515-519): position j*n_in/n_out in input samples, clamped to the last sample, double math.
__global__ __launch_bounds__(256) void k_resample_linear(const float* __restrict__ x, long long n_in, long long n_out,
                                                          float* __restrict__ y) {
    const float* xr = x + (size_t)blockIdx.y * n_in;
    float* yr = y + (size_t)blockIdx.y * n_out;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n_out; j += (long long)gridDim.x * 256) {
        const double t = (double)j / (double)n_out;              // linspace(0, 1, n_out, endpoint=False)[j]
        long long i = (long long)(t * (double)n_in);
        if ((double)i / (double)n_in > t) --i;                    // largest i with t_old[i] <= t
        if (i >= n_in - 1) { yr[j] = xr[n_in - 1]; continue; }
        const double t0 = (double)i / (double)n_in, t1 = (double)(i + 1) / (double)n_in;
        const double y0 = xr[i], y1 = xr[i + 1];
        yr[j] = (float)(y0 + (t - t0) * ((y1 - y0) / (t1 - t0)));
    }
}

// ---------------------------------------------------------------- DeepFilterNet stage glue (egregora_audio_enhance_extras.py:548-704)
// rms[c][f] = sqrt(mean(x[c][480 f .. 480 f + 480)^2)) (last frame short), one wave per frame, sum in double.
__global__ __launch_bounds__(64) void k_frame_rms(const float* __restrict__ x, long long T, long long n_frames,
                                                   float* __restrict__ rms) {
    const long long f = blockIdx.x;
    const float* xc = x + (size_t)blockIdx.y * T;
    const long long a = f * 480, b = a + 480 < T ? a + 480 : T;
    double acc = 0.0;
    for (long long i = a + threadIdx.x; i < b; i += 64) acc += (double)xc[i] * (double)xc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) rms[(size_t)blockIdx.y * n_frames + f] = (float)sqrt(acc / (double)(b - a));
}

// Per channel (one thread: the smoothing is a recurrence): probs = clip(rms / p95, 0, 1); one-pole smoothing; adaptive strength;
// wet/dry gains.  Every step rounds to float32 exactly where numpy does (no fused multiply-adds), see oracle/dfn_mix.py.
struct DfnP {
    float alpha, oma;      // exp(-10 ms / tau) and 1 - alpha (both already rounded to float32); smooth = 0: no smoothing
    int smooth, mode;      // mode 0 off, 1 more_on_noise, 2 more_on_speech, 3 gate_on_noise
    float s0, a, one_m_s0, thr, s_noise, s_speech;
    int linear;            // 0: equal power (sin / cos of pi/2 * s), 1: linear
    double pos_frac;       // fractional part of 0.95 * (n - 1): numpy's linear percentile between the two order statistics
};
__global__ void k_dfn_gains(const float* __restrict__ rms, const float* __restrict__ ostat2, long long n, DfnP q,
                            float* __restrict__ g_dry, float* __restrict__ g_wet) {
    if (threadIdx.x != 0) return;
    const int c = blockIdx.x;
    const float* r = rms + (size_t)c * n;
    const double lo = ostat2[2 * c], hi = ostat2[2 * c + 1], t = q.pos_frac;
    // numpy _lerp: a + (b - a) t, or b - (b - a)(1 - t) for t >= 0.5 (float32 order statistics, float64 arithmetic)
    double p95 = t >= 0.5 ? hi - (hi - lo) * (1.0 - t) : lo + (hi - lo) * t;
    p95 = (double)(float)p95;                       // np.percentile of a float32 array returns float32
    if (p95 == 0.0) p95 = 1e-6;
    const float p95f = (float)p95;
    float acc = 0.f;
    for (long long i = 0; i < n; ++i) {
        float v = __fdiv_rn(r[i], p95f);
        v = fminf(fmaxf(v, 0.f), 1.f);
        if (q.smooth) {
            if (i == 0) acc = v;
            acc = __fadd_rn(__fmul_rn(q.alpha, acc), __fmul_rn(q.oma, v));
            v = fminf(fmaxf(acc, 0.f), 1.f);
        }
        float s;
        if (q.mode == 1) s = __fadd_rn(q.s0, __fmul_rn(__fmul_rn(q.a, __fsub_rn(1.0f, v)), q.one_m_s0));
        else if (q.mode == 2) s = __fadd_rn(q.s0, __fmul_rn(__fmul_rn(q.a, v), q.one_m_s0));
        else if (q.mode == 3) s = v < q.thr ? q.s_noise : q.s_speech;
        else s = q.s0;
        s = fminf(fmaxf(s, 0.f), 1.f);
        float gd, gw;
        if (q.linear) { gw = s; gd = __fsub_rn(1.0f, s); }
        else {
            const float ang = __fmul_rn(1.57079637f, s);       // float32(0.5 * pi) * s
            gw = sinf(ang);
            gd = cosf(ang);
        }
        g_dry[(size_t)c * n + i] = gd;
        g_wet[(size_t)c * n + i] = gw;
    }
}

// y = clip(g_dry[f] * dry + g_wet[f] * wet, -1, 1) * gain, frame f = i / hop (the last gain repeats); running max |y|
__global__ __launch_bounds__(256) void k_dfn_mix(const float* __restrict__ dry, const float* __restrict__ wet,
                                                  const float* __restrict__ g_dry, const float* __restrict__ g_wet, long long T,
                                                  long long n, int hop, float gain, int use_gain, float* __restrict__ y,
                                                  unsigned* __restrict__ peak) {
    __shared__ float red[4];
    const size_t co = (size_t)blockIdx.y * T, go = (size_t)blockIdx.y * n;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < T; i += (long long)gridDim.x * 256) {
        long long f = i / hop;
        if (f > n - 1) f = n - 1;
        float v = __fadd_rn(__fmul_rn(g_dry[go + f], dry[co + i]), __fmul_rn(g_wet[go + f], wet[co + i]));
        v = fminf(fmaxf(v, -1.f), 1.f);
        if (use_gain) v = __fmul_rn(v, gain);
        y[co + i] = v;
        mx = fmaxf(mx, fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(peak, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// limiter: if (limit && peak > ceiling && peak > 0) y *= float32(ceiling / peak); then clamp to [-1, 1]
__global__ __launch_bounds__(256) void k_dfn_limit(float* __restrict__ y, long long n, const unsigned* __restrict__ peak,
                                                    int limit, double ceiling) {
    const float pk = __uint_as_float(*peak);
    const bool scale = limit && (double)pk > ceiling && pk > 0.f;
    const float sc = scale ? (float)(ceiling / (double)pk) : 1.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = y[i];
        if (scale) v = __fmul_rn(v, sc);
        y[i] = fminf(fmaxf(v, -1.f), 1.f);
    }
}

// Delay compensation of the null-test suite (_apply_frac_delay_CN + _pad_or_crop_CN, egregora_null_test_suite.py:203-266):
// s[i] = x[i - shift] inside [0, n_in) else 0; y[i] = sum_k h[k] s[i + (taps-1)/2 - k]  (np.convolve(s, h, "same"); taps = 0:
// y = s); output length n_out (zero beyond n_in).  float32 products summed in order of k.
__global__ __launch_bounds__(256) void k_shift_fir(const float* __restrict__ x, long long n_in, long long shift,
                                                    const float* __restrict__ h, int taps, long long n_out,
                                                    float* __restrict__ y) {
    const float* xr = x + (size_t)blockIdx.y * n_in;
    float* yr = y + (size_t)blockIdx.y * n_out;
    const int off = (taps - 1) / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (long long)gridDim.x * 256) {
        float acc = 0.f;
        if (i < n_in) {
            if (taps <= 0) {
                const long long j = i - shift;
                acc = (j >= 0 && j < n_in) ? xr[j] : 0.f;
            } else {
                for (int k = 0; k < taps; ++k) {
                    const long long si = i + off - k;          // index into the shifted signal
                    const long long j = si - shift;
                    if (si >= 0 && si < n_in && j >= 0 && j < n_in) acc += h[k] * xr[j];
                }
            }
        }
        yr[i] = acc;
    }
}

// ---------------------------------------------------------------- null-test suite levels and sums (egregora_null_test_suite.py:119-165, 420-470)
// K-weighting approximation of the suite's loudness: z = (1-k) x + k z, y = x - z (each product and sum rounded to float32 as numpy's
// scalar loop does), then y[n] += 0.02 (y[n] - y[n-1]).  The recurrence contracts by k per sample, so a thread restarts it W samples
// ahead of its L-sample chunk (k^W < 1e-11: below float32 resolution of the state).
__global__ __launch_bounds__(64) void k_kweight(const float* __restrict__ x, long long n, float a1, float kf, int L, int W,
                                                float* __restrict__ y) {
    const long long s = ((long long)blockIdx.x * 64 + threadIdx.x) * L;
    if (s >= n) return;
    const float* xc = x + (size_t)blockIdx.y * n;
    float* yc = y + (size_t)blockIdx.y * n;
    long long i = s - W - 1;
    if (i < 0) i = 0;
    float z = 0.f, yp = 0.f;
    for (; i < s; ++i) {
        const float xv = xc[i];
        z = __fadd_rn(__fmul_rn(a1, xv), __fmul_rn(kf, z));
        yp = __fsub_rn(xv, z);
    }
    const long long e = s + L < n ? s + L : n;
    for (; i < e; ++i) {
        const float xv = xc[i];
        z = __fadd_rn(__fmul_rn(a1, xv), __fmul_rn(kf, z));
        const float yv = __fsub_rn(xv, z);
        yc[i] = i > 0 ? __fadd_rn(yv, __fmul_rn(0.02f, __fsub_rn(yv, yp))) : yv;
        yp = yv;
    }
}

// mean over channels as numpy's float32 .mean(axis=0): rows added in order, one division
__device__ __forceinline__ float mono_mean(const float* __restrict__ x, int C, long long ld, long long i) {
    float m = x[i];
    for (int c = 1; c < C; ++c) m = __fadd_rn(m, x[(size_t)c * ld + i]);
    return C > 1 ? __fdiv_rn(m, (float)C) : m;
}

__global__ __launch_bounds__(256) void k_mono_mean(const float* __restrict__ x, int C, long long ld, long long n, float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = mono_mean(x, C, ld, i);
}

// out[f] = mean(mono[f hop .. f hop + blk)^2) in double (the block is cut at n), one workgroup per block
__global__ __launch_bounds__(256) void k_frame_meansq(const float* __restrict__ x, int C, long long n, long long blk, long long hop,
                                                      double* __restrict__ out) {
    __shared__ double red[256];
    const long long a = (long long)blockIdx.x * hop, b = a + blk < n ? a + blk : n;
    double acc = 0.0;
    for (long long i = a + threadIdx.x; i < b; i += 256) {
        const double m = (double)mono_mean(x, C, n, i);
        acc += m * m;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] / (double)(b - a);
}

template <int K>
__device__ __forceinline__ void block_add_f64(double (&v)[K], double* __restrict__ out) {
    __shared__ double red[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < K) atomicAdd(out + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// sums over the mono downmixes a_m, b_m (b optionally scaled by kf in float32 first): out += {sum a, sum b, sum ab, sum aa, sum bb}
__global__ __launch_bounds__(256) void k_pair_stats(const float* __restrict__ a, int ca, long long lda, const float* __restrict__ b,
                                                    int cb, long long ldb, long long n, float kf, int use_k, double* __restrict__ out) {
    double v[5] = {0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double x = (double)mono_mean(a, ca, lda, i);
        float m = use_k ? __fmul_rn(b[i], kf) : b[i];
        for (int c = 1; c < cb; ++c) m = __fadd_rn(m, use_k ? __fmul_rn(b[(size_t)c * ldb + i], kf) : b[(size_t)c * ldb + i]);
        const double yv = (double)(cb > 1 ? __fdiv_rn(m, (float)cb) : m);
        v[0] += x; v[1] += yv; v[2] += x * yv; v[3] += x * x; v[4] += yv * yv;
    }
    block_add_f64<5>(v, out);
}

// null[c][i] = a[c][i] + sgn * (b[c][i] * kf); out += {sum null_m^2, count(|null| > 1)}
__global__ __launch_bounds__(256) void k_null_mix(const float* __restrict__ a, long long lda, const float* __restrict__ b, long long ldb,
                                                  int C, long long n, float kf, int use_k, float sgn, float* __restrict__ nul,
                                                  double* __restrict__ out) {
    double v[2] = {0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float m = 0.f;
        for (int c = 0; c < C; ++c) {
            float bv = b[(size_t)c * ldb + i];
            if (use_k) bv = __fmul_rn(bv, kf);
            const float d = __fadd_rn(a[(size_t)c * lda + i], sgn * bv);
            nul[(size_t)c * n + i] = d;
            if (fabsf(d) > 1.0f) v[1] += 1.0;
            m = c ? __fadd_rn(m, d) : d;
        }
        if (C > 1) m = __fdiv_rn(m, (float)C);
        v[0] += (double)m * (double)m;
    }
    block_add_f64<2>(v, out);
}

// out += {sum x^2, sum x, sum (-1)^i x, sum y^2, sum y, sum (-1)^i y}: the time-domain side of Parseval for one-sided band energies
__global__ __launch_bounds__(256) void k_band_sums(const float* __restrict__ x, const float* __restrict__ y, long long n,
                                                   double* __restrict__ out) {
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double a = x[i], b = y[i], s = (i & 1) ? -1.0 : 1.0;
        v[0] += a * a; v[1] += a; v[2] += s * a; v[3] += b * b; v[4] += b; v[5] += s * b;
    }
    block_add_f64<6>(v, out);
}

static inline int grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace egr

using namespace egr;

extern "C" int egr_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount -> %s", hipGetErrorString(e));
        return -EGR_ERR_HIP;
    }
    return n;
}

extern "C" int egr_device_arch(int dev, char* buf, size_t buflen) {
    EGR_CHECK(buf && buflen > 0, EGR_ERR_ARG, "null buffer");
    hipDeviceProp_t prop;
    EGR_HIP(hipGetDeviceProperties(&prop, dev));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return EGR_OK;
}

extern "C" int egr_pcm16_roundtrip(const float* x, float* y, int64_t n, float write_scale, float read_div,
                                   void* stream) {
    EGR_CHECK(x && y && n >= 0 && read_div != 0.f, EGR_ERR_ARG, "bad argument");
    if (n == 0) return EGR_OK;
    hipLaunchKernelGGL(k_pcm16, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n, write_scale,
                       read_div);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_chunk_gather(const float* x, int channels, int64_t total, int64_t win, int64_t hop,
                                int chunk_begin, int n_chunks, float* chunks, void* stream) {
    EGR_CHECK(x && chunks && channels >= 1 && total >= 0 && win >= 1 && hop >= 1 && chunk_begin >= 0 && n_chunks >= 0,
              EGR_ERR_ARG, "bad argument");
    if (n_chunks == 0) return EGR_OK;
    EGR_CHECK((int64_t)n_chunks * channels <= 65535, EGR_ERR_ARG, "too many chunk rows for one launch");
    int gx = grid_for(win);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_chunk_gather, dim3(gx, n_chunks * channels), dim3(256), 0, (hipStream_t)stream, x, channels,
                       (long long)total, (long long)win, (long long)hop, chunk_begin, chunks);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_wola_stitch(const float* preds, int n_chunks, int channels, int64_t lp, int64_t total, int64_t win,
                               int64_t hop, const float* window, float* out, void* stream) {
    EGR_CHECK(preds && window && out && n_chunks >= 1 && channels >= 1 && lp >= 1 && total >= 1 && win >= 1 && hop >= 1,
              EGR_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_wola, dim3(grid_for(total), channels), dim3(256), 0, (hipStream_t)stream, preds, n_chunks,
                       channels, (long long)lp, (long long)total, (long long)win, (long long)hop, window, out);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_resample_poly(const float* x, int channels, int64_t n_in, int up, int down, const float* h, int half,
                                 float* y, int64_t n_out, void* stream) {
    EGR_CHECK(x && h && y && channels >= 1 && channels <= 65535 && n_in >= 1 && up >= 1 && down >= 1 && half >= 0,
              EGR_ERR_ARG, "bad argument");
    EGR_CHECK(n_out == (n_in * up + down - 1) / down, EGR_ERR_ARG, "n_out must be ceil(n_in*up/down)");
    hipLaunchKernelGGL(k_resample_poly, dim3(grid_for(n_out), channels), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)n_in, (long long)n_out, up, down, h, half, y);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_stft_mag(const float* x, int channels, int64_t n, int n_fft, int hop, const float* window,
                            float* out, void* stream) {
    EGR_CHECK(x && window && out && channels >= 1 && n >= 0 && hop >= 1, EGR_ERR_ARG, "bad argument");
    EGR_CHECK(n_fft >= 2 && (n_fft % 2) == 0 && n_fft <= 8192, EGR_ERR_UNSUPPORTED, "n_fft=%d must be even and <= 8192",
              n_fft);
    StftTables t;
    int rc = stft_tables(n_fft, &t);
    if (rc) return rc;
    const int64_t frames = 1 + ((n - n_fft) > 0 ? (n - n_fft) / hop : 0);
    const size_t lds = (size_t)2 * (n_fft / 2) * sizeof(float2);
    hipLaunchKernelGGL(k_stft_mag, dim3((unsigned)frames), dim3(256), lds, (hipStream_t)stream, x, channels,
                       (long long)n, n_fft, hop, window, t.fd, t.tw, t.wsplit, out);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_lsd_frames(const float* SA, const float* SB, int64_t frames, int nb, float* per, void* stream) {
    EGR_CHECK(SA && SB && per && frames >= 1 && frames < (1LL << 31) && nb >= 1, EGR_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_lsd_frames, dim3((unsigned)frames), dim3(256), 0, (hipStream_t)stream, SA, SB, nb, per);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_sum_f64(const float* v, int64_t n, double* out, void* stream) {
    EGR_CHECK(v && out && n >= 1, EGR_ERR_ARG, "bad argument");
    EGR_HIP(hipMemsetAsync(out, 0, sizeof(double), (hipStream_t)stream));
    long long nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_sum_f64, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, v, (long long)n, out);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_order_stats2(const float* v, int64_t n, int64_t k_lo, int64_t k_hi, float* out2, void* stream) {
    EGR_CHECK(v && out2 && n >= 1 && k_lo >= 0 && k_lo < n && k_hi >= 0 && k_hi < n, EGR_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_order_stats2, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, (long long)n, (long long)k_lo,
                       (long long)k_hi, out2);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_si_sdr_terms(const float* s, int cs, int64_t stride_s, const float* s_hat, int csh, int64_t stride_sh,
                                int64_t n, double* out4, void* stream) {
    EGR_CHECK(s && s_hat && out4 && cs >= 1 && csh >= 1 && n >= 1 && stride_s >= n && stride_sh >= n, EGR_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    EGR_HIP(hipMemsetAsync(out4, 0, 4 * sizeof(double), st));
    long long nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    for (int mode = 0; mode < 2; ++mode)
        hipLaunchKernelGGL(k_sisdr, dim3((unsigned)nb), dim3(256), 0, st, s, cs, (long long)stride_s, s_hat, csh,
                           (long long)stride_sh, (long long)n, mode, out4);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_resample_linear(const float* x, int channels, int64_t n_in, float* y, int64_t n_out, void* stream) {
    EGR_CHECK(x && y && channels >= 1 && channels <= 65535 && n_in >= 1 && n_out >= 1, EGR_ERR_ARG, "bad argument");
    long long nb = (n_out + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_resample_linear, dim3((unsigned)nb, channels), dim3(256), 0, (hipStream_t)stream, x, (long long)n_in,
                       (long long)n_out, y);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" size_t egr_dfn_workspace_bytes(int channels, int64_t n48) {
    const int64_t n = (n48 + 479) / 480;
    return (size_t)channels * n * sizeof(float) + (size_t)channels * 2 * sizeof(float) + 64;
}

extern "C" int egr_dfn_vad_gains(const float* dry48, int channels, int64_t n48, double smooth_ms, int mode, double strength,
                                 double amount, double vad_threshold, int linear_curve, void* workspace, float* g_dry,
                                 float* g_wet, void* stream) {
    EGR_CHECK(dry48 && workspace && g_dry && g_wet && channels >= 1 && channels <= 65535 && n48 >= 1, EGR_ERR_ARG, "bad argument");
    EGR_CHECK(mode >= 0 && mode <= 3 && strength >= 0.0 && strength <= 1.0, EGR_ERR_ARG, "bad mode / strength");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (n48 + 479) / 480;
    float* rms = (float*)workspace;
    float* ost = rms + (size_t)channels * n;
    hipLaunchKernelGGL(k_frame_rms, dim3((unsigned)n, channels), dim3(64), 0, st, dry48, (long long)n48, (long long)n, rms);
    const double pos = 0.95 * (double)(n - 1);
    const int64_t lo = (int64_t)pos, hi = lo + 1 < n ? lo + 1 : n - 1;
    for (int c = 0; c < channels; ++c)
        hipLaunchKernelGGL(k_order_stats2, dim3(1), dim3(1024), 0, st, rms + (size_t)c * n, (long long)n, (long long)lo,
                           (long long)hi, ost + 2 * c);
    DfnP q;
    const double s0 = strength, a = amount;          // Python floats in the reference: constants are formed in double
    const double alpha = smooth_ms > 0.0 ? exp(-10.0 / fmax(1e-3, smooth_ms)) : 0.0;
    q.alpha = (float)alpha; q.oma = (float)(1.0 - alpha); q.smooth = smooth_ms > 0.0 ? 1 : 0; q.mode = mode;
    q.s0 = (float)s0; q.a = (float)a; q.one_m_s0 = (float)(1.0 - s0); q.thr = (float)vad_threshold;
    q.s_noise = (float)fmin(fmax(s0 + a * (1.0 - s0), 0.0), 1.0); q.s_speech = (float)fmin(fmax(s0 * (1.0 - a), 0.0), 1.0);
    q.linear = linear_curve ? 1 : 0; q.pos_frac = pos - (double)lo;
    hipLaunchKernelGGL(k_dfn_gains, dim3(channels), dim3(64), 0, st, rms, ost, (long long)n, q, g_dry, g_wet);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_dfn_mix(const float* dry, const float* wet, const float* g_dry, const float* g_wet, int channels, int64_t n,
                           int64_t n_frames, int hop, float post_gain, int use_gain, int limit, double ceiling, float* y,
                           void* peak_ws, void* stream) {
    EGR_CHECK(dry && wet && g_dry && g_wet && y && peak_ws && channels >= 1 && channels <= 65535 && n >= 1 && n_frames >= 1 &&
                  hop >= 1, EGR_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    EGR_HIP(hipMemsetAsync(peak_ws, 0, sizeof(unsigned), st));
    long long nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_dfn_mix, dim3((unsigned)nb, channels), dim3(256), 0, st, dry, wet, g_dry, g_wet, (long long)n,
                       (long long)n_frames, hop, post_gain, use_gain, y, (unsigned*)peak_ws);
    const long long tot = (long long)n * channels;
    long long nb2 = (tot + 255) / 256;
    if (nb2 > 4096) nb2 = 4096;
    hipLaunchKernelGGL(k_dfn_limit, dim3((unsigned)nb2), dim3(256), 0, st, y, tot, (const unsigned*)peak_ws, limit, ceiling);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_shift_fir(const float* x, int channels, int64_t n_in, int64_t shift, const float* h, int taps, float* y,
                             int64_t n_out, void* stream) {
    EGR_CHECK(x && y && channels >= 1 && channels <= 65535 && n_in >= 1 && n_out >= 1 && taps >= 0 && (taps == 0 || h),
              EGR_ERR_ARG, "bad argument");
    long long nb = (n_out + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_shift_fir, dim3((unsigned)nb, channels), dim3(256), 0, (hipStream_t)stream, x, (long long)n_in,
                       (long long)shift, h, taps, (long long)n_out, y);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_kweight(const float* x, int channels, int64_t n, float one_minus_k, float k, float* y, void* stream) {
    EGR_CHECK(x && y && channels >= 1 && channels <= 65535 && n >= 1 && k > 0.f && k < 1.f, EGR_ERR_ARG, "bad argument");
    int W = (int)ceil(26.0 / -log((double)k));       // k^W < 6e-12
    if (W < 64) W = 64;
    const int L = (W + 2) / 3;
    const long long chunks = (n + L - 1) / L;
    hipLaunchKernelGGL(k_kweight, dim3((unsigned)((chunks + 63) / 64), channels), dim3(64), 0, (hipStream_t)stream, x, (long long)n,
                       one_minus_k, k, L, W, y);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_mono_mean(const float* x, int channels, int64_t stride, int64_t n, float* y, void* stream) {
    EGR_CHECK(x && y && channels >= 1 && n >= 1 && stride >= n, EGR_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(k_mono_mean, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, channels, (long long)stride, (long long)n, y);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_frame_meansq(const float* x, int channels, int64_t n, int64_t block, int64_t hop, int64_t frames, double* out,
                                void* stream) {
    EGR_CHECK(x && out && channels >= 1 && n >= 1 && block >= 1 && hop >= 1 && frames >= 1 && (frames - 1) * hop < n, EGR_ERR_ARG,
              "bad argument");
    hipLaunchKernelGGL(k_frame_meansq, dim3((unsigned)frames), dim3(256), 0, (hipStream_t)stream, x, channels, (long long)n,
                       (long long)block, (long long)hop, out);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_pair_stats(const float* a, int ca, int64_t stride_a, const float* b, int cb, int64_t stride_b, int64_t n, float k,
                              int use_k, double* out5, void* stream) {
    EGR_CHECK(a && b && out5 && ca >= 1 && cb >= 1 && n >= 1, EGR_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    EGR_HIP(hipMemsetAsync(out5, 0, 5 * sizeof(double), st));
    hipLaunchKernelGGL(k_pair_stats, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, a, ca, (long long)stride_a, b, cb,
                       (long long)stride_b, (long long)n, k, use_k, out5);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_null_mix(const float* a, int64_t stride_a, const float* b, int64_t stride_b, int channels, int64_t n, float k,
                            int use_k, int invert_b, float* null_out, double* out2, void* stream) {
    EGR_CHECK(a && b && null_out && out2 && channels >= 1 && n >= 1, EGR_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    EGR_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(double), st));
    hipLaunchKernelGGL(k_null_mix, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, a, (long long)stride_a, b,
                       (long long)stride_b, channels, (long long)n, k, use_k, invert_b ? -1.0f : 1.0f, null_out, out2);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_band_sums(const float* x, const float* y, int64_t n, double* out6, void* stream) {
    EGR_CHECK(x && y && out6 && n >= 1, EGR_ERR_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    EGR_HIP(hipMemsetAsync(out6, 0, 6 * sizeof(double), st));
    hipLaunchKernelGGL(k_band_sums, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, x, y, (long long)n, out6);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}
