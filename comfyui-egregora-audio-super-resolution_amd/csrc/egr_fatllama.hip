// Fat-Llama iterative spectral enhancer on MI355X (gfx950).
//
// Replaces upstream fat_llama.audio_fattener.feed.upscale as called from the reference at
// egregora_fat_llama_gpu.py:213-224 / egregora_fat_llama_cpu.py:126-134 (see include/egregora_amd.h).
//
// Data layout in HBM (per plan; C channels, N = n_in*factor real samples, M = N/2 = M1*M2):
//   out  [C][N]  float   holds y (the up-rated signal) during the loop, the final result afterwards
//   work [C][M]  float2  the loop state, updated IN PLACE by both loop kernels:
//                          after k_col : A[k1][n2]  (column FFT over n1 done, four-step twiddle applied)
//                          after k_row : B[k1][n2]  (half-spectrum thresholded, row IFFT over k2 done)
// One loop iteration = k_row (FFT_M2 . real-split . threshold . un-split . IFFT_M2 on the row pair
// (k1, M1-k1), all in LDS) followed by k_col (twiddle^-1 . IFFT_M1 . [time domain] . FFT_M1 . twiddle on a
// tile of TC adjacent columns, all in LDS).  Each launch reads 4N and writes 4N bytes per channel; the
// spectrum is never materialised in natural order and nothing is transposed.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "egr_common.h"
#include "egr_fft_device.h"
#include "egr_plan.h"

namespace egr {

struct FlParams {
    FftDesc f1, f2;
    int M1, M2;
    long long M, N;
    int TC, TClog2, ntiles, tiles_per_xcd;
    const cplx* tw1;   // W_M1^q   (stage table of f1; also the high part of the four-step twiddle)
    const cplx* tw2;   // W_M2^j   (stage table of f2)
    const cplx* T2;    // W_M^s,  s < M2   (low part of the four-step twiddle)
    const cplx* T3;    // W_N^k1, k1 < M1  (real-split twiddle, low part)
    const cplx* T4;    // W_N^(M1*k2), k2 < M2
    unsigned long long magic_m2;   // ceil(2^44 / M2): r / M2 for r < 2^23
    float thr, thr2, inv_M;
};

__device__ __forceinline__ cplx four_step_tw(const FlParams& p, int n2, int k1) {
    const unsigned r = (unsigned)n2 * (unsigned)k1;            // < M <= 2^22
    const unsigned q = (unsigned)(((unsigned long long)r * p.magic_m2) >> 44);
    const unsigned s = r - q * (unsigned)p.M2;
    return cmul(p.tw1[q], p.T2[s]);
}

__device__ __forceinline__ void atomic_max_abs(unsigned* slot, float v) {
    // non-negative IEEE floats order like unsigned ints
    atomicMax(slot, __float_as_uint(v));
}

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, red[i]);
    __syncthreads();
    return r;
}

// MODE 0: first  (y -> time threshold -> FFT_M1 -> twiddle -> A)
// MODE 1: middle (B -> twiddle^-1 -> IFFT_M1 -> FFT_M1 -> twiddle -> A)
// MODE 2: last   (B -> twiddle^-1 -> IFFT_M1 -> out = y + d, per-channel max|out|)
template <int MODE>
__global__ __launch_bounds__(256) void k_col(FlParams p, cplx* __restrict__ work, float* __restrict__ out,
                                              unsigned* __restrict__ peak_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[8];
    // XCD-aware tile order: the dispatcher places block b on XCD b%8; give every XCD a contiguous run
    // of column tiles so the two tiles sharing a 128-byte line hit the same L2.
    const int tile = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int ch = blockIdx.y;
    const int TC = p.TC, lg = p.TClog2, M1 = p.M1, M2 = p.M2;
    const int c0 = tile * TC;
    cplx* cur = (cplx*)smem;
    cplx* alt = cur + (size_t)M1 * TC;
    cplx* W = work + (size_t)ch * p.M;
    float2* Y = (float2*)(out + (size_t)ch * p.N);
    const int nel = M1 * TC;

    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
        cplx v = make_float2(0.f, 0.f);
        if (col < M2) {
            const size_t g = (size_t)i * M2 + col;
            if (MODE == 0) {
                const float2 y = Y[g];
                v.x = fabsf(y.x) > p.thr ? y.x : 0.f;
                v.y = fabsf(y.y) > p.thr ? y.y : 0.f;
            } else {
                v = cmulc(W[g], four_step_tw(p, col, i));
            }
        }
        cur[e] = v;
    }
    __syncthreads();
    if (MODE != 0) lds_fft<true>(cur, alt, p.f1, p.tw1, TC, lg, TC, 1, true);
    if (MODE == 2) {
        float mx = 0.f;
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            if (col < M2) {
                const size_t g = (size_t)i * M2 + col;
                const float2 y = Y[g];
                const cplx d = cur[e];
                const float2 o = make_float2(__fadd_rn(y.x, d.x), __fadd_rn(y.y, d.y));
                Y[g] = o;
                mx = fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y)));
            }
        }
        mx = block_max(mx, red);
        if (threadIdx.x == 0) atomic_max_abs(peak_out + ch, mx);
        return;
    }
    lds_fft<true>(cur, alt, p.f1, p.tw1, TC, lg, TC, 1, false);
    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
        if (col < M2) W[(size_t)i * M2 + col] = cmul(cur[e], four_step_tw(p, col, i));
    }
}

// Row-pair kernel: rows ka = pair, kb = M1 - pair of the in-place state.
__global__ __launch_bounds__(256) void k_row(FlParams p, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M1 = p.M1, M2 = p.M2;
    const int ka = blockIdx.x;
    const int kb = (M1 - ka) % M1;
    const bool self = (ka == kb);
    const int nrows = self ? 1 : 2;
    const int ch = blockIdx.y;
    cplx* cur = (cplx*)smem;
    cplx* alt = cur + 2 * (size_t)M2;
    cplx* W = work + (size_t)ch * p.M;
    cplx* ga = W + (size_t)ka * M2;
    cplx* gb = W + (size_t)kb * M2;

    for (int e = threadIdx.x; e < M2; e += blockDim.x) {
        cur[e] = ga[e];
        if (!self) cur[M2 + e] = gb[e];
    }
    __syncthreads();
    lds_fft<false>(cur, alt, p.f2, p.tw2, nrows, 0, 1, M2, false);

    // real-split, threshold, un-split on the (k, M-k) pairs; Z[k1 + M1*k2] sits at row(k1)[k2]
    const cplx wa = p.T3[ka];
    int cnt, boff;          // partner of row-a element k2 is row-b element (boff - k2) mod M2
    cplx* rb;
    if (!self) { cnt = M2; boff = M2 - 1; rb = cur + M2; }
    else if (ka == 0) { cnt = M2 / 2 + 1; boff = M2; rb = cur; }
    else { cnt = (M2 + 1) / 2; boff = M2 - 1; rb = cur; }
    const float thr2 = p.thr2, sc = p.inv_M;
    for (int k2 = threadIdx.x; k2 < cnt; k2 += blockDim.x) {
        int pb = boff - k2;
        if (pb >= M2) pb -= M2;
        const bool same = self && (pb == k2);
        const cplx Za = cur[k2];
        const cplx Zb = rb[pb];
        const cplx Wk = cmul(wa, p.T4[k2]);
        // E = (Za + conj Zb)/2 ; O = (Za - conj Zb)/(2i)
        const cplx E = make_float2(0.5f * (Za.x + Zb.x), 0.5f * (Za.y - Zb.y));
        const cplx O = make_float2(0.5f * (Za.y + Zb.y), -0.5f * (Za.x - Zb.x));
        const cplx WO = cmul(Wk, O);
        cplx Xk = cadd(E, WO);      // X[k]
        cplx Xm = csub(E, WO);      // conj X[M-k]
        if (!(Xk.x * Xk.x + Xk.y * Xk.y > thr2)) Xk = make_float2(0.f, 0.f);
        if (!(Xm.x * Xm.x + Xm.y * Xm.y > thr2)) Xm = make_float2(0.f, 0.f);
        const cplx E2 = make_float2(0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y + Xm.y));
        const cplx H = make_float2(0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y - Xm.y));
        const cplx O2 = cmulc(H, Wk);
        // Za' = E2 + i*O2 ; Zb' = conj(E2 - i*O2)
        cur[k2] = make_float2(sc * (E2.x - O2.y), sc * (E2.y + O2.x));
        if (!same) rb[pb] = make_float2(sc * (E2.x + O2.y), -sc * (E2.y - O2.x));
    }
    __syncthreads();
    lds_fft<false>(cur, alt, p.f2, p.tw2, nrows, 0, 1, M2, true);
    for (int e = threadIdx.x; e < M2; e += blockDim.x) {
        ga[e] = cur[e];
        if (!self) gb[e] = cur[M2 + e];
    }
}

// x (optionally PCM_16-quantised) -> y = linear up-rate by f, per-channel max|x_q|.
__global__ __launch_bounds__(256) void k_prepare(const float* __restrict__ x, float* __restrict__ y, long long n_in,
                                                  int f, int pcm_in, unsigned* __restrict__ peak_in) {
    __shared__ float red[8];
    const int ch = blockIdx.y;
    const float* xc = x + (size_t)ch * n_in;
    float* yc = y + (size_t)ch * n_in * f;
    const float ff = (float)f;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_in;
         i += (long long)gridDim.x * blockDim.x) {
        float a = xc[i];
        float b = (i + 1 < n_in) ? xc[i + 1] : 0.f;
        if (pcm_in) {
            long long qa = (long long)rintf(__fmul_rn(a, 32767.0f));
            long long qb = (long long)rintf(__fmul_rn(b, 32767.0f));
            qa = ((qa + 32768) & 65535) - 32768;
            qb = ((qb + 32768) & 65535) - 32768;
            a = (float)qa;
            b = (float)qb;
        }
        mx = fmaxf(mx, fabsf(a));
        if (i + 1 < n_in) {
            for (int j = 0; j < f; ++j) {
                const float t = __fdiv_rn((float)j, ff);
                const float u = __fsub_rn(1.0f, t);
                yc[i * f + j] = __fadd_rn(__fmul_rn(u, a), __fmul_rn(t, b));
            }
        } else {
            for (int j = 0; j < f; ++j) yc[i * f + j] = 0.f;   // upstream leaves the last sample's slots zero
        }
    }
    mx = block_max(mx, red);
    if (threadIdx.x == 0) atomic_max_abs(peak_in + ch, mx);
}

// max_iter == 0 path: out = y + (|y|>thr ? y : 0), peaks
__global__ __launch_bounds__(256) void k_noiter(float* __restrict__ y, long long N, float thr,
                                                 unsigned* __restrict__ peak_out) {
    __shared__ float red[8];
    const int ch = blockIdx.y;
    float* yc = y + (size_t)ch * N;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N;
         i += (long long)gridDim.x * blockDim.x) {
        const float v = yc[i];
        const float o = __fadd_rn(v, fabsf(v) > thr ? v : 0.f);
        yc[i] = o;
        mx = fmaxf(mx, fabsf(o));
    }
    mx = block_max(mx, red);
    if (threadIdx.x == 0) atomic_max_abs(peak_out + ch, mx);
}

// autoscale / normalise / write patch / PCM_16 round trip, all driven by the 2*C peak scalars.
__global__ __launch_bounds__(256) void k_finalize(float* __restrict__ out, long long N, int C, unsigned flags,
                                                   const unsigned* __restrict__ peak_in,
                                                   const unsigned* __restrict__ peak_out) {
    const int ch = blockIdx.y;
    float s_auto = 1.f;
    float joint = 0.f;
    for (int c = 0; c < C; ++c) {
        const float pi = __uint_as_float(peak_in[c]), po = __uint_as_float(peak_out[c]);
        float s = 1.f, m = po;
        if ((flags & EGR_FL_AUTOSCALE) && po > 0.f) {
            s = (float)((double)pi / (double)po);
            m = __fmul_rn(po, s);
        }
        if (c == ch) s_auto = s;
        joint = fmaxf(joint, m);
    }
    const bool do_auto = (flags & EGR_FL_AUTOSCALE) != 0;
    const bool do_norm = (flags & EGR_FL_NORMALIZE) && joint > 0.f;
    float peak_final = do_norm ? 1.0f : joint;
    const bool patch = (flags & EGR_FL_NODE_POST) && peak_final > 1.0f;
    float* oc = out + (size_t)ch * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N;
         i += (long long)gridDim.x * blockDim.x) {
        float v = oc[i];
        if (do_auto) v = __fmul_rn(v, s_auto);
        if (do_norm) v = __fdiv_rn(v, joint);
        if (flags & EGR_FL_NODE_POST) {
            if (patch) v = __fmul_rn(v, 1.0f / 32768.0f);
            long long q = (long long)rintf(__fmul_rn(v, 32767.0f));
            q = ((q + 32768) & 65535) - 32768;
            v = __fmul_rn((float)q, 1.0f / 32768.0f);
        }
        oc[i] = v;
    }
}

}  // namespace egr

using namespace egr;

struct egr_fatllama_plan {
    int64_t n_in;
    int C, factor, device;
    FlSplit sp;
    FlParams prm;
    cplx *d_tw1, *d_tw2, *d_T2, *d_T3, *d_T4, *d_work;
    unsigned* d_peaks;   // [2*C]: peak_in[C], peak_out[C]
    bool profiling;
    std::vector<hipEvent_t> ev;   // pairs (start, stop) tagged by kind
    std::vector<int> ev_kind;     // 0 = row, 1 = col
};

static int upload(const std::vector<float2>& h, cplx** d) {
    EGR_HIP(hipMalloc((void**)d, h.size() * sizeof(float2)));
    EGR_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    return EGR_OK;
}

extern "C" int egr_fatllama_plan_query(int64_t n_in, int factor, int m1_hint, int64_t info[EGR_FL_INFO_LEN]) {
    EGR_CHECK(info != nullptr, EGR_ERR_ARG, "info is null");
    memset(info, 0, sizeof(int64_t) * EGR_FL_INFO_LEN);
    EGR_CHECK(n_in >= 1 && factor >= 1, EGR_ERR_ARG, "n_in=%lld factor=%d out of range", (long long)n_in, factor);
    const int64_t N = n_in * factor;
    FlSplit sp = plan_split(N, m1_hint, 0);
    info[1] = N;
    info[2] = N / 2;
    if (!sp.ok) {
        set_error("length %lld unsupported: needs even N with N/2 = M1*M2, M1<=1024, M2<=4096, primes<=13", (long long)N);
        return EGR_ERR_UNSUPPORTED;
    }
    info[0] = 1;
    info[3] = sp.M1; info[4] = sp.M2; info[5] = sp.TC; info[6] = sp.f1.nst; info[7] = sp.f2.nst;
    for (int i = 0; i < EGR_MAX_STAGES; ++i) { info[8 + i] = sp.f1.radix[i]; info[22 + i] = sp.f2.radix[i]; }
    info[36] = (int64_t)sp.lds_col;
    info[37] = (int64_t)sp.lds_row;
    return EGR_OK;
}

extern "C" int egr_fatllama_plan_destroy(egr_fatllama_plan* p) {
    if (!p) return EGR_OK;
    hipFree(p->d_tw1); hipFree(p->d_tw2); hipFree(p->d_T2); hipFree(p->d_T3); hipFree(p->d_T4);
    hipFree(p->d_work); hipFree(p->d_peaks);
    for (auto e : p->ev) hipEventDestroy(e);
    delete p;
    return EGR_OK;
}

extern "C" int egr_fatllama_plan_create(egr_fatllama_plan** out, int64_t n_in, int channels, int factor,
                                        int m1_hint, int tc_hint) {
    EGR_CHECK(out != nullptr, EGR_ERR_ARG, "out is null");
    *out = nullptr;
    EGR_CHECK(n_in >= 1 && factor >= 1 && channels >= 1 && channels <= 64, EGR_ERR_ARG,
              "n_in=%lld channels=%d factor=%d out of range", (long long)n_in, channels, factor);
    const int64_t N = n_in * factor;
    FlSplit sp = plan_split(N, m1_hint, tc_hint);
    if (!sp.ok) {
        set_error("length %lld unsupported: needs even N with N/2 = M1*M2, M1<=1024, M2<=4096, primes<=13", (long long)N);
        return EGR_ERR_UNSUPPORTED;
    }
    egr_fatllama_plan* p = new egr_fatllama_plan();
    p->n_in = n_in; p->C = channels; p->factor = factor; p->sp = sp; p->profiling = false;
    p->d_tw1 = p->d_tw2 = p->d_T2 = p->d_T3 = p->d_T4 = p->d_work = nullptr;
    p->d_peaks = nullptr;
    hipGetDevice(&p->device);
    std::vector<float2> h;
    int rc = EGR_OK;
    const int64_t M = sp.M;
    make_twiddles(h, sp.M1, 1, sp.M1); if ((rc = upload(h, &p->d_tw1))) { egr_fatllama_plan_destroy(p); return rc; }
    make_twiddles(h, sp.M2, 1, sp.M2); if ((rc = upload(h, &p->d_tw2))) { egr_fatllama_plan_destroy(p); return rc; }
    make_twiddles(h, sp.M2, 1, M);     if ((rc = upload(h, &p->d_T2))) { egr_fatllama_plan_destroy(p); return rc; }
    make_twiddles(h, sp.M1, 1, N);     if ((rc = upload(h, &p->d_T3))) { egr_fatllama_plan_destroy(p); return rc; }
    make_twiddles(h, sp.M2, 1, 2 * (int64_t)sp.M2); if ((rc = upload(h, &p->d_T4))) { egr_fatllama_plan_destroy(p); return rc; }
    if (hipMalloc((void**)&p->d_work, (size_t)channels * M * sizeof(float2)) != hipSuccess ||
        hipMalloc((void**)&p->d_peaks, 2 * channels * sizeof(unsigned)) != hipSuccess) {
        set_error("hipMalloc of the %lld-byte loop state failed", (long long)(channels * M * 8));
        egr_fatllama_plan_destroy(p);
        return EGR_ERR_ALLOC;
    }
    FlParams& q = p->prm;
    memset(&q, 0, sizeof(q));
    q.f1 = sp.f1; q.f2 = sp.f2; q.M1 = sp.M1; q.M2 = sp.M2; q.M = M; q.N = N;
    q.TC = sp.TC; q.TClog2 = sp.TClog2;
    q.ntiles = ceil_div(sp.M2, sp.TC);
    q.tiles_per_xcd = ceil_div(q.ntiles, 8);
    q.tw1 = p->d_tw1; q.tw2 = p->d_tw2; q.T2 = p->d_T2; q.T3 = p->d_T3; q.T4 = p->d_T4;
    q.magic_m2 = ((1ULL << 44) + (unsigned long long)sp.M2 - 1) / (unsigned long long)sp.M2;
    q.inv_M = (float)(1.0 / (double)M);
    // dynamic LDS above the 64 KiB default needs an explicit opt-in per kernel
    hipError_t e = hipSuccess;
    e = hipFuncSetAttribute((const void*)k_col<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds_col);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds_col);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds_col);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_row, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds_row);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) -> %s", hipGetErrorString(e));
        egr_fatllama_plan_destroy(p);
        return EGR_ERR_HIP;
    }
    *out = p;
    return EGR_OK;
}

extern "C" int egr_fatllama_set_profiling(egr_fatllama_plan* p, int enable) {
    EGR_CHECK(p != nullptr, EGR_ERR_ARG, "plan is null");
    p->profiling = enable != 0;
    return EGR_OK;
}

static inline void prof_begin(egr_fatllama_plan* p, int kind, hipStream_t st, size_t* slot) {
    if (!p->profiling) return;
    if (*slot + 2 > p->ev.size()) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        p->ev.push_back(a); p->ev.push_back(b);
        p->ev_kind.push_back(kind);
    } else {
        p->ev_kind[*slot / 2] = kind;
    }
    hipEventRecord(p->ev[*slot], st);
}
static inline void prof_end(egr_fatllama_plan* p, hipStream_t st, size_t* slot) {
    if (!p->profiling) return;
    hipEventRecord(p->ev[*slot + 1], st);
    *slot += 2;
}

extern "C" int egr_fatllama_enhance(egr_fatllama_plan* p, const float* x, float* out, int max_iter, float thr,
                                    unsigned flags, void* stream) {
    EGR_CHECK(p && x && out, EGR_ERR_ARG, "null plan/x/out");
    EGR_CHECK(max_iter >= 0, EGR_ERR_ARG, "max_iter=%d < 0", max_iter);
    hipStream_t st = (hipStream_t)stream;
    FlParams q = p->prm;
    q.thr = thr;
    q.thr2 = thr * thr;
    const int C = p->C;
    unsigned* peak_in = p->d_peaks;
    unsigned* peak_out = p->d_peaks + C;
    EGR_HIP(hipMemsetAsync(p->d_peaks, 0, 2 * C * sizeof(unsigned), st));
    {
        const int nb = (int)((p->n_in + 255) / 256 < 2048 ? (p->n_in + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_prepare, dim3(nb, C), dim3(256), 0, st, x, out, (long long)p->n_in, p->factor,
                           (flags & EGR_FL_PCM_IN) ? 1 : 0, peak_in);
    }
    const dim3 gcol(8 * q.tiles_per_xcd, C), grow(q.M1 / 2 + 1, C), blk(256);
    const size_t lc = p->sp.lds_col, lr = p->sp.lds_row;
    size_t slot = 0;
    if (max_iter == 0) {
        const int nb = (int)((q.N + 255) / 256 < 2048 ? (q.N + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_noiter, dim3(nb, C), blk, 0, st, out, q.N, thr, peak_out);
    } else {
        hipLaunchKernelGGL(k_col<0>, gcol, blk, lc, st, q, p->d_work, out, peak_out);
        for (int it = 0; it < max_iter; ++it) {
            prof_begin(p, 0, st, &slot);
            hipLaunchKernelGGL(k_row, grow, blk, lr, st, q, p->d_work);
            prof_end(p, st, &slot);
            if (it + 1 < max_iter) {
                prof_begin(p, 1, st, &slot);
                hipLaunchKernelGGL(k_col<1>, gcol, blk, lc, st, q, p->d_work, out, peak_out);
                prof_end(p, st, &slot);
            }
        }
        hipLaunchKernelGGL(k_col<2>, gcol, blk, lc, st, q, p->d_work, out, peak_out);
    }
    if (flags & (EGR_FL_NORMALIZE | EGR_FL_AUTOSCALE | EGR_FL_NODE_POST)) {
        const int nb = (int)((q.N + 255) / 256 < 2048 ? (q.N + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_finalize, dim3(nb, C), blk, 0, st, out, q.N, C, flags, peak_in, peak_out);
    }
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_fatllama_last_peaks(egr_fatllama_plan* p, float* host_pin, float* host_pout, void* stream) {
    EGR_CHECK(p && host_pin && host_pout, EGR_ERR_ARG, "null argument");
    std::vector<unsigned> h(2 * p->C);
    EGR_HIP(hipStreamSynchronize((hipStream_t)stream));
    EGR_HIP(hipMemcpy(h.data(), p->d_peaks, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    for (int c = 0; c < p->C; ++c) {
        memcpy(&host_pin[c], &h[c], 4);
        memcpy(&host_pout[c], &h[p->C + c], 4);
    }
    return EGR_OK;
}

extern "C" int egr_fatllama_kernel_times(egr_fatllama_plan* p, double* row_ms_avg, double* col_ms_avg,
                                         int64_t* row_launches, int64_t* col_launches) {
    EGR_CHECK(p != nullptr, EGR_ERR_ARG, "plan is null");
    double sum[2] = {0, 0};
    int64_t cnt[2] = {0, 0};
    EGR_HIP(hipDeviceSynchronize());
    for (size_t i = 0; i + 1 < p->ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]) != hipSuccess) continue;
        const int k = p->ev_kind[i / 2];
        sum[k] += ms;
        cnt[k] += 1;
    }
    if (row_ms_avg) *row_ms_avg = cnt[0] ? sum[0] / cnt[0] : 0.0;
    if (col_ms_avg) *col_ms_avg = cnt[1] ? sum[1] / cnt[1] : 0.0;
    if (row_launches) *row_launches = cnt[0];
    if (col_launches) *col_launches = cnt[1];
    return EGR_OK;
}
