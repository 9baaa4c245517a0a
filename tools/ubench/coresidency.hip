// dev: standalone bisect of the co-residency erratum: k_stft_frames returns wrong bins when its workgroups share
// compute units with k_conv_s3 workgroups of another stream.  No torch: victims (variants of the STFT-frame kernel) run on one
// stream while an aggressor runs on another; every victim output is compared with the output of a solo run.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../comfyui-egregora-audio-super-resolution_amd/csrc coresidency.hip \
//         -L../../comfyui-egregora-audio-super-resolution_amd -legregora_amd -o coresidency
//   LD_LIBRARY_PATH=../../comfyui-egregora-audio-super-resolution_amd ./coresidency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "egr_plan.h"      // FftDesc, make_schedule, make_twiddles, lds_fft (egr_fft_device.h)

using namespace egr;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- victims
// V 0: the shipped kernel.  1: no transform, no table read (window product -> LDS -> magnitude of the packed pair).
// 2: transform, but |cur[k]| only (no partner read, no split-table read).  3: shipped arithmetic, results staged in registers and
// written after a barrier.  4: shipped, with a full s_waitcnt + barrier before the output loop.  5: shipped arithmetic with every
// intermediate pinned in its own register (asm barriers) so the compiler cannot form packed-fp32 (v_pk_*_f32) instructions in
// the output loop.  6: packed arithmetic, but the split twiddle is a constant instead of the global table read.  7: like 5 with
// the split twiddle constant as well (neither packed math nor the table read).
template <int V>
__global__ __launch_bounds__(256) void k_victim(const float* __restrict__ x, int L, int n_fft, int hop, int rpad, int T, int ldm,
                                                 const float* __restrict__ window, FftDesc fd, const cplx* __restrict__ tw,
                                                 const cplx* __restrict__ wsplit, float* __restrict__ mag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Mh = n_fft / 2;
    cplx* cur = (cplx*)smem;
    cplx* alt = cur + Mh;
    const int t = blockIdx.x, b = blockIdx.y;
    float* o = mag + ((size_t)b * T + t) * ldm;
    const float* xb = x + (size_t)b * L;
    const int s0 = t * hop - rpad;
    for (int e = threadIdx.x; e < Mh; e += blockDim.x) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int i = s0 + 2 * e + h;
            if (i < 0) i = -i;
            if (i > L - 1) i = 2 * (L - 1) - i;
            i = i < 0 ? 0 : (i > L - 1 ? L - 1 : i);
            v[h] = xb[i] * window[2 * e + h];
        }
        cur[e] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    if (V != 1) lds_fft<false>(cur, alt, fd, tw, 1, 0, 1, Mh, false);
    if (V == 4) { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }
    float stage[8];
    int ns = 0;
    for (int k = threadIdx.x; k < ldm; k += blockDim.x) {
        float r = 0.f;
        if (k <= Mh) {
            if (V == 1 || V == 2) {
                const cplx Z = cur[k == Mh ? 0 : k];
                r = sqrtf(Z.x * Z.x + Z.y * Z.y);
            } else if (V == 5 || V == 7) {
#define PIN(v) asm volatile("" : "+v"(v))
                const cplx Za = cur[k == Mh ? 0 : k];
                const cplx Zb = cur[(k == 0 || k == Mh) ? 0 : Mh - k];
                const cplx w = (V == 7) ? make_float2(0.6f, -0.8f) : wsplit[k];
                float ax = Za.x, ay = Za.y, bx = Zb.x, by = Zb.y, wx = w.x, wy = w.y;
                PIN(ax); PIN(ay); PIN(bx); PIN(by); PIN(wx); PIN(wy);
                float ex = ax + bx; PIN(ex); ex *= 0.5f; PIN(ex);
                float ey = ay - by; PIN(ey); ey *= 0.5f; PIN(ey);
                float ox = ay + by; PIN(ox); ox *= 0.5f; PIN(ox);
                float oy = ax - bx; PIN(oy); oy *= -0.5f; PIN(oy);
                float px = wx * ox; PIN(px); px = fmaf(-wy, oy, px); PIN(px);
                float py = wx * oy; PIN(py); py = fmaf(wy, ox, py); PIN(py);
                float xr = ex + px; PIN(xr);
                float xi = ey + py; PIN(xi);
                float m = xr * xr; PIN(m); m = fmaf(xi, xi, m); PIN(m);
                r = sqrtf(m);
#undef PIN
            } else {
                const cplx Za = cur[k == Mh ? 0 : k];
                const cplx Zb = cur[(k == 0 || k == Mh) ? 0 : Mh - k];
                if (V == 6) {
                    const cplx E6 = make_float2(0.5f * (Za.x + Zb.x), 0.5f * (Za.y - Zb.y));
                    const cplx O6 = make_float2(0.5f * (Za.y + Zb.y), -0.5f * (Za.x - Zb.x));
                    const cplx X6 = cadd(E6, cmul(make_float2(0.6f, -0.8f), O6));
                    r = sqrtf(X6.x * X6.x + X6.y * X6.y);
                    if (V == 3) stage[ns++] = r; else o[k] = r;
                    continue;
                }
                const cplx E = make_float2(0.5f * (Za.x + Zb.x), 0.5f * (Za.y - Zb.y));
                const cplx O = make_float2(0.5f * (Za.y + Zb.y), -0.5f * (Za.x - Zb.x));
                const cplx X = cadd(E, cmul(wsplit[k], O));
                r = sqrtf(X.x * X.x + X.y * X.y);
            }
        }
        if (V == 3) stage[ns++] = r;
        else o[k] = r;
    }
    if (V == 3) {
        __syncthreads();
        ns = 0;
        for (int k = threadIdx.x; k < ldm; k += blockDim.x) o[k] = stage[ns++];
    }
}

// ---------------------------------------------------------------- aggressors
// 1: bf16 MFMA only (registers).  2: LDS ds_read_b128 traffic + barriers only.  3: f32 MFMA only.  4: bf16 MFMA + LDS reads + barrier.
template <int A>
__global__ __launch_bounds__(256, 2) void k_aggr(float* sink, int iters) {
    __shared__ uint4 lds[3072];
    for (int i = threadIdx.x; i < 3072; i += 256) lds[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    uint4 a = lds[threadIdx.x], b = lds[threadIdx.x + 256];
    float facc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (A == 2 || A == 4) {
            a = lds[(threadIdx.x * 7 + it) % 3072];
            b = lds[(threadIdx.x * 5 + it * 3) % 3072];
        }
        if (A == 1 || A == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
        }
        if (A == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc[j], 0, 0, 0);
        }
        if (A == 2) facc += __uint_as_float(a.x) + __uint_as_float(b.y);
        if (A == 2 || A == 4) __syncthreads();
    }
    float s = facc;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) sink[0] = s;
}

// ---------------------------------------------------------------- micro victims: which instruction class goes wrong?
// M 0: packed-fp32 FMA chain, all lanes active.  1: the same under a partial EXEC mask (16 of 64 lanes).  2: packed ops with
// component swizzles (op_sel forms: complex multiply).  3: plain scalar fp32 FMA chain (control).  4: registers only -- 48 values
// are held across a long scalar spin and written back unchanged (is the register file itself disturbed?).  5: v_sqrt chain.
// 6: v_pk_add_f32 without swizzle, with and without neg modifiers (clean).  7: the same add with op_sel swapping the second operand's
// halves (20/20 runs wrong).  8: v_pk_mov_b32 with swapped halves -- a pure MOVE (15/20 runs wrong): the fault sits in the VOP3P
// op_sel routing of a high source dword into the low lane, not in the arithmetic.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int Mv>
__global__ __launch_bounds__(256) void k_micro(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    f32x2 x = {in[2 * gid], in[2 * gid + 1]};
    const f32x2 w = {0.999f + 1e-4f * (float)(lane & 7), -0.03f + 1e-3f * (float)(lane & 3)};
    f32x2 acc = {0.25f, -0.125f};
    if (Mv == 0 || Mv == 1) {
        if (Mv == 0 || lane < 16) {
            for (int i = 0; i < iters; ++i) { acc = acc * w + x; x = x * w - acc * 0.5f; }
        }
    } else if (Mv == 2) {
        for (int i = 0; i < iters; ++i) {                     // complex multiply-accumulate: swizzled operands
            const f32x2 xs = __builtin_shufflevector(x, x, 1, 0);
            const f32x2 wn = {-w.y, w.y};
            acc = acc + x * w.x + xs * wn;
            x = __builtin_shufflevector(acc, acc, 1, 0) * 0.5f + x * 0.5f;
        }
    } else if (Mv == 3) {
        float a0 = acc.x, a1 = acc.y, x0 = x.x, x1 = x.y;
        for (int i = 0; i < iters; ++i) {
            a0 = fmaf(a0, w.x, x0); asm volatile("" : "+v"(a0));
            a1 = fmaf(a1, w.y, x1); asm volatile("" : "+v"(a1));
            x0 = fmaf(x0, w.x, -0.5f * a0); asm volatile("" : "+v"(x0));
            x1 = fmaf(x1, w.y, -0.5f * a1); asm volatile("" : "+v"(x1));
        }
        acc.x = a0; acc.y = a1; x.x = x0; x.y = x1;
    } else if (Mv == 4) {
        float r[48];
#pragma unroll
        for (int j = 0; j < 48; ++j) { r[j] = x.x * (float)(j + 1) + x.y; asm volatile("" : "+v"(r[j])); }
        int spin = 0;
        for (int i = 0; i < iters * 8; ++i) { spin += i ^ (spin >> 3); asm volatile("" : "+s"(spin)); }
        float sacc = (float)(spin & 1) * 0.f;
#pragma unroll
        for (int j = 0; j < 48; ++j) { asm volatile("" : "+v"(r[j])); sacc += r[j] * (float)(1 + (j & 3)); }
        acc.x = sacc; acc.y = r[17];
    } else if (Mv == 6) {                                         // unswizzled packed add / subtract (neg modifiers), as csrc cadd / csub
        for (int i = 0; i < iters; ++i) {
            f32x2 t;
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(acc), "v"(x));
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(acc), "v"(x));
            acc.x = t.x * 0.5f; acc.y = t.y * 0.25f; x.x *= 0.75f; x.y *= 0.5f;
            asm volatile("" : "+v"(acc), "+v"(x));
        }
    } else if (Mv == 7) {                                         // the same add with the second operand's halves swapped (op_sel)
        for (int i = 0; i < iters; ++i) {
            f32x2 t;
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(acc), "v"(x));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x) : "v"(acc), "v"(x));
            acc.x = t.x * 0.5f; acc.y = t.y * 0.25f; x.x *= 0.75f; x.y *= 0.5f;
            asm volatile("" : "+v"(acc), "+v"(x));
        }
    } else if (Mv == 8) {                                         // packed MOVE with swapped halves between scalar arithmetic
        for (int i = 0; i < iters; ++i) {
            f32x2 t;
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(acc), "v"(x));
            acc.x = fmaf(t.x, 0.5f, x.y); acc.y = fmaf(t.y, 0.25f, x.x);
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(x) : "v"(x), "v"(x));
            x.x *= 0.75f; x.y *= 0.5f;
            asm volatile("" : "+v"(acc), "+v"(x));
        }
    } else {
        float a0 = fabsf(x.x) + 1.f, a1 = fabsf(x.y) + 2.f;
        for (int i = 0; i < iters; ++i) { a0 = sqrtf(a0 * 1.5f + 1.f); asm volatile("" : "+v"(a0)); a1 = sqrtf(a1 + a0); asm volatile("" : "+v"(a1)); }
        acc.x = a0; acc.y = a1;
    }
    out[2 * gid] = acc.x + x.x;
    out[2 * gid + 1] = acc.y - x.y;
}

extern "C" {
int egr_conv_s3(const float* x, const void* w3, const float* bias, const float* bias_b, const float* res, float* y, int B, int H, int W, int Cin,
                int OH, int OW, int Cout, int KH, int KW, int stride, int dil, int pad_t, int pad_l, int up2, int act, float act_param, int osy,
                int osx, int ooy, int oox, int OHF, int OWF, int nz, int64_t zx, int64_t zw3, int64_t zy, void* stream);
int egr_split3_pack(const float* w_packed, void* w3, int64_t nslabs, int Cout, void* stream);
typedef struct egr_fatllama_plan egr_fatllama_plan;
int egr_fatllama_plan_create(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, int m1_hint, int tc_hint);
int egr_fatllama_enhance(egr_fatllama_plan* plan, const float* x, float* out, int max_iter, float threshold, unsigned flags, void* stream);
int egr_fatllama_set_graph(egr_fatllama_plan* plan, int enable);
const char* egr_last_error(void);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int V>
static void launch_victim(hipStream_t st, const float* x, int B, int L, int n_fft, int hop, int T, int ldm, const float* win, const FftDesc& fd,
                          const cplx* tw, const cplx* ws, float* mag) {
    const int rpad = (n_fft - hop) / 2;
    hipLaunchKernelGGL(k_victim<V>, dim3(T, B), dim3(256), (size_t)2 * (n_fft / 2) * sizeof(float2), st, x, L, n_fft, hop, rpad, T, ldm, win, fd,
                       tw, ws, mag);
}

int main() {
    const int B = 9, L = 245760, n_fft = 2048, hop = 480, T = 512, ldm = 1040, Cout = 256;
    FftDesc fd;
    if (!make_schedule(n_fft / 2, &fd, 127)) { printf("schedule failed\n"); return 1; }
    std::vector<float2> h;
    cplx *tw, *ws;
    make_twiddles(h, n_fft / 2, 1, n_fft / 2);
    CK(hipMalloc(&tw, h.size() * 8)); CK(hipMemcpy(tw, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    make_twiddles(h, n_fft / 2 + 1, 1, n_fft);
    CK(hipMalloc(&ws, h.size() * 8)); CK(hipMemcpy(ws, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<float> hx((size_t)B * L), hw(n_fft);
    srand(1);
    for (auto& v : hx) v = 0.2f * ((rand() % 2001) / 1000.0f - 1.0f);
    for (int i = 0; i < n_fft; ++i) hw[i] = 0.5f - 0.5f * cosf(6.283185307f * i / n_fft);
    float *x, *win, *mag[3], *ref, *sink, *cx, *cy, *wp;
    void* w3;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&win, n_fft * 4)); CK(hipMemcpy(win, hw.data(), n_fft * 4, hipMemcpyHostToDevice));
    const size_t nmag = (size_t)B * T * ldm;
    for (int i = 0; i < 3; ++i) CK(hipMalloc(&mag[i], nmag * 4));
    CK(hipMalloc(&ref, nmag * 4)); CK(hipMalloc(&sink, 64));
    // aggressor 0: the library's k_conv_s3 on the mel-GEMM shape (M = B*T rows, K = ldm, N = 256)
    const int slabs = ldm / 16;
    std::vector<float> hwp((size_t)slabs * Cout * 16);
    for (auto& v : hwp) v = ((rand() % 2001) / 1000.0f - 1.0f) * 0.03f;
    CK(hipMalloc(&wp, hwp.size() * 4)); CK(hipMemcpy(wp, hwp.data(), hwp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&w3, hwp.size() * 6));
    if (egr_split3_pack(wp, w3, slabs, Cout, nullptr)) { printf("split3: %s\n", egr_last_error()); return 1; }
    CK(hipMalloc(&cx, nmag * 4)); CK(hipMemset(cx, 0, nmag * 4));
    CK(hipMalloc(&cy, (size_t)B * T * Cout * 4));
    hipStream_t s[4];
    for (auto& q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    std::vector<float> href(nmag), hgot(nmag);
    const char* an[5] = {"k_conv_s3 (library)", "bf16 MFMA only", "LDS reads + barriers only", "f32 MFMA only", "bf16 MFMA + LDS reads + barrier"};
    const char* vn[8] = {"shipped kernel", "no FFT, no table read", "FFT, |cur[k]| only", "register-staged stores", "waitcnt+barrier before output",
                         "no packed-fp32 in the output loop", "constant split twiddle (no table read), packed math", "no packed-fp32, no table read"};
    auto victim = [&](int V, hipStream_t st, float* out) {
        switch (V) {
            case 0: launch_victim<0>(st, x, B, L, n_fft, hop, T, ldm, win, fd, tw, ws, out); break;
            case 1: launch_victim<1>(st, x, B, L, n_fft, hop, T, ldm, win, fd, tw, ws, out); break;
            case 2: launch_victim<2>(st, x, B, L, n_fft, hop, T, ldm, win, fd, tw, ws, out); break;
            case 3: launch_victim<3>(st, x, B, L, n_fft, hop, T, ldm, win, fd, tw, ws, out); break;
            case 4: launch_victim<4>(st, x, B, L, n_fft, hop, T, ldm, win, fd, tw, ws, out); break;
            case 5: launch_victim<5>(st, x, B, L, n_fft, hop, T, ldm, win, fd, tw, ws, out); break;
            case 6: launch_victim<6>(st, x, B, L, n_fft, hop, T, ldm, win, fd, tw, ws, out); break;
            default: launch_victim<7>(st, x, B, L, n_fft, hop, T, ldm, win, fd, tw, ws, out); break;
        }
    };
    auto aggressor = [&](int A, hipStream_t st) {
        if (A == 0) {
            if (egr_conv_s3(cx, w3, nullptr, nullptr, nullptr, cy, B * T, 1, 1, ldm, 1, 1, Cout, 1, 1, 1, 1, 0, 0, 0, 0, 0.f, 1, 1, 0, 0, 1, 1, 1, 0, 0, 0, st))
                printf("conv: %s\n", egr_last_error());
        } else if (A == 1) hipLaunchKernelGGL(k_aggr<1>, dim3(1024), dim3(256), 0, st, sink, 2000);
        else if (A == 2) hipLaunchKernelGGL(k_aggr<2>, dim3(1024), dim3(256), 0, st, sink, 2000);
        else if (A == 3) hipLaunchKernelGGL(k_aggr<3>, dim3(1024), dim3(256), 0, st, sink, 1000);
        else hipLaunchKernelGGL(k_aggr<4>, dim3(1024), dim3(256), 0, st, sink, 2000);
    };
    for (int V = 0; V < 8; ++V) {
        victim(V, s[1], ref);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(href.data(), ref, nmag * 4, hipMemcpyDeviceToHost));
        for (int A = 0; A < 5; ++A) {
            if (V > 0 && A > 0 && A != 4) continue;             // the full aggressor matrix only for the shipped victim
            if (V == 1 || V == 3 || V == 4 || V == 7) continue;
            int bad_runs = 0;
            long long bad_vals = 0, first_row = -1, first_col = -1;
            for (int rep = 0; rep < 30; ++rep) {
                for (int i = 0; i < 6; ++i) aggressor(A, s[0]);
                for (int i = 0; i < 3; ++i) { victim(V, s[1], mag[0]); victim(V, s[2], mag[1]); }
                CK(hipDeviceSynchronize());
                bool bad = false;
                for (int m = 0; m < 2; ++m) {
                    CK(hipMemcpy(hgot.data(), mag[m], nmag * 4, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < nmag; ++i)
                        if (hgot[i] != href[i]) { bad = true; ++bad_vals; if (first_row < 0) { first_row = (long long)(i / ldm); first_col = (long long)(i % ldm); } }
                }
                bad_runs += bad;
            }
            printf("victim %d (%s) next to aggressor %d (%s): bad runs %d / 30, wrong values %lld%s", V, vn[V], A, an[A], bad_runs, bad_vals,
                   first_row >= 0 ? "" : "\n");
            if (first_row >= 0) printf(" (first at row %lld col %lld)\n", first_row, first_col);
            fflush(stdout);
        }
    }
    // ---- the Fat-Llama loop (complex arithmetic = component-swapped packed fp32 all over) next to the aggressors
    if (!getenv("CR_SKIP_FL")) {
        const int n = 480000;
        egr_fatllama_plan* plan = nullptr;
        if (egr_fatllama_plan_create(&plan, n, 1, 1, 0, 0)) { printf("plan: %s\n", egr_last_error()); return 1; }
        egr_fatllama_set_graph(plan, 0);
        float *fx, *fo, *fr;
        CK(hipMalloc(&fx, n * 4)); CK(hipMalloc(&fo, n * 4)); CK(hipMalloc(&fr, n * 4));
        std::vector<float> hfx(n), hfr(n), hfo(n);
        for (auto& v : hfx) v = (float)((rand() % 16001) - 8000);
        CK(hipMemcpy(fx, hfx.data(), n * 4, hipMemcpyHostToDevice));
        egr_fatllama_enhance(plan, fx, fr, 20, 0.6f, 0, s[1]);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hfr.data(), fr, n * 4, hipMemcpyDeviceToHost));
        for (int A : {4, 1, 0, 3}) {
            int bad_runs = 0; long long bad_vals = 0;
            for (int rep = 0; rep < 20; ++rep) {
                for (int i = 0; i < 12; ++i) aggressor(A, s[0]);
                egr_fatllama_enhance(plan, fx, fo, 20, 0.6f, 0, s[1]);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hfo.data(), fo, n * 4, hipMemcpyDeviceToHost));
                bool bad = false;
                for (int i = 0; i < n; ++i) if (memcmp(&hfo[i], &hfr[i], 4)) { bad = true; ++bad_vals; }
                bad_runs += bad;
            }
            printf("fat-llama loop (N = %d, 20 iterations) next to aggressor %d (%s): bad runs %d / 20, wrong values %lld\n", n, A, an[A], bad_runs, bad_vals);
            fflush(stdout);
        }
    }
    // ---- micro victims
    {
        const int nwg = 4096, n = nwg * 256 * 2;
        float *min_, *mout[2], *mref;
        CK(hipMalloc(&min_, n * 4)); CK(hipMalloc(&mout[0], n * 4)); CK(hipMalloc(&mout[1], n * 4)); CK(hipMalloc(&mref, n * 4));
        std::vector<float> hi(n), hr(n), hg(n);
        for (auto& v : hi) v = ((rand() % 2001) / 1000.0f - 1.0f);
        CK(hipMemcpy(min_, hi.data(), n * 4, hipMemcpyHostToDevice));
        const char* mn[9] = {"packed-fp32 FMA chain, full EXEC", "packed-fp32 FMA chain, 16 of 64 lanes", "packed-fp32 with swizzled operands (op_sel)",
                             "scalar fp32 FMA chain", "48 registers held across a scalar spin", "v_sqrt_f32 chain",
                             "v_pk_add_f32 unswizzled, with and without neg", "v_pk_add_f32 with swapped halves (op_sel)",
                             "v_pk_mov_b32 with swapped halves"};
        auto micro = [&](int Mv, hipStream_t st, float* o) {
            switch (Mv) {
                case 0: hipLaunchKernelGGL(k_micro<0>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
                case 1: hipLaunchKernelGGL(k_micro<1>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
                case 2: hipLaunchKernelGGL(k_micro<2>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
                case 3: hipLaunchKernelGGL(k_micro<3>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
                case 4: hipLaunchKernelGGL(k_micro<4>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
                case 6: hipLaunchKernelGGL(k_micro<6>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
                case 7: hipLaunchKernelGGL(k_micro<7>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
                case 8: hipLaunchKernelGGL(k_micro<8>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
                default: hipLaunchKernelGGL(k_micro<5>, dim3(nwg), dim3(256), 0, st, min_, o, 200); break;
            }
        };
        for (int Mv = 0; Mv < 9; ++Mv) {
            micro(Mv, s[1], mref);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hr.data(), mref, n * 4, hipMemcpyDeviceToHost));
            for (int A : {1, 4, 3}) {
                int bad_runs = 0; long long bad_vals = 0; long long first = -1;
                for (int rep = 0; rep < 20; ++rep) {
                    for (int i = 0; i < 6; ++i) aggressor(A, s[0]);
                    for (int i = 0; i < 4; ++i) { micro(Mv, s[1], mout[0]); micro(Mv, s[2], mout[1]); }
                    CK(hipDeviceSynchronize());
                    bool bad = false;
                    for (int m = 0; m < 2; ++m) {
                        CK(hipMemcpy(hg.data(), mout[m], n * 4, hipMemcpyDeviceToHost));
                        for (int i = 0; i < n; ++i) if (memcmp(&hg[i], &hr[i], 4)) { bad = true; ++bad_vals; if (first < 0) first = i; }
                    }
                    bad_runs += bad;
                }
                printf("micro %d (%s) next to aggressor %d (%s): bad runs %d / 20, wrong values %lld", Mv, mn[Mv], A, an[A], bad_runs, bad_vals);
                if (first >= 0) printf(" (first: element %lld = lane %lld of wave %lld)", first, (first / 2) % 64, (first / 2) / 64);
                printf("\n");
                fflush(stdout);
            }
        }
    }
    return 0;
}
