// Dense contractions of the FlashSR graph on the gfx950 matrix cores, exact fp32:
//   k_conv_igemm : implicit-GEMM convolution over channels-last activations (2-D 3x3 / 1x1 / strided /
//                  nearest-2x-upsampled input, 1-D dilated), fused bias / per-row channel bias / residual /
//                  activation epilogue.  Also serves every Linear layer (1x1, H=W=1).
//   k_bgemm      : strided batched GEMM (QK^T with B transposed, PV) for the attention blocks.
// Both use v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bitwise an fmaf chain, MI355X_MICROARCH.md
// "Matrix cores"), a 128 x BN x 16 block tile staged through double-buffered LDS in k-major order so an MFMA
// operand fetch is two conflict-free 32-lane rows, 4 waves per workgroup each owning TM x TN 32x32 accumulators.
// At 64 cycles per MFMA the pipe, not LDS or HBM, is the bound for Cin*Cout >= 128*128; the roofline for these
// kernels is the 157.3 TFLOP/s f32 matrix peak.
#include <string.h>

#include "egr_common.h"

namespace egr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_TANH = 2, ACT_LEAKY01 = 3, ACT_LOGCLAMP = 4 };

struct ConvP {
    const float* x;        // [B][H][W][Cin]   (physical; logical input is 2H x 2W when up2 != 0)
    const float* w;        // [KH][KW][Cin][Cout]
    const float* bias;     // [Cout] or null
    const float* bias_b;   // [B][Cout] or null (time-embedding bias)
    const float* res;      // [M][Cout] or null
    float* y;              // [M][Cout]
    int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, dil, pad_t, pad_l, up2, act;
    int M, K;
    float act_param;
};

#define BM 128
#define BK 16
#define LPAD 4

__device__ __forceinline__ float apply_act(float v, int act, float prm) {
    switch (act) {
        case ACT_SILU: return v / (1.0f + __expf(-v));
        case ACT_TANH: return tanhf(v);
        case ACT_LEAKY01: return v > 0.f ? v : 0.1f * v;
        case ACT_LOGCLAMP: return __logf(fmaxf(v, prm));
        default: return v;
    }
}

// acc[tm][tn] += A(32 rows x 2 k) * B(2 k x 32 cols) over one BK slab held in LDS
template <int TM, int TN, int BN>
__device__ __forceinline__ void mma_slab(const float (*As)[BM + LPAD], const float (*Bs)[BN + LPAD], int wm0, int wn0,
                                         f32x16 (&acc)[TM][TN]) {
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[kk + lk][wm0 + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[kk + lk][wn0 + j * 32 + li];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// BN = 128: waves 2x2, each 64x64 (TM=2,TN=2); BN = 64: waves 2x2, each 64x32 (2,1); BN = 32: waves 4x1, each 32x32.
template <int BN> struct TileCfg;
template <> struct TileCfg<128> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 2; };
template <> struct TileCfg<64> { static constexpr int WM = 2, WN = 2, TM = 2, TN = 1; };
template <> struct TileCfg<32> { static constexpr int WM = 4, WN = 1, TM = 1, TN = 1; };

template <int BN, bool VEC>
__global__ __launch_bounds__(256) void k_conv_igemm(ConvP p) {
    typedef TileCfg<BN> TC;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + LPAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + LPAD];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int wm0 = (wave / TC::WN) * (BM / TC::WM), wn0 = (wave % TC::WN) * (BN / TC::WN);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int LH = p.up2 ? 2 * p.H : p.H, LW = p.up2 ? 2 * p.W : p.W;   // logical input extent

    f32x16 acc[TC::TM][TC::TN];
#pragma unroll
    for (int i = 0; i < TC::TM; ++i)
#pragma unroll
        for (int j = 0; j < TC::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- per-thread A-row bookkeeping ----
    // VEC: thread owns rows (tid>>2) and (tid>>2)+64, channel quad (tid&3).  Generic: 8 scalars, row = e%128.
    int rb[2], roy[2], rox[2];
    bool rvalid[2];
    if (VEC) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + (tid >> 2) + 64 * h;
            rvalid[h] = m < p.M;
            const int mm = rvalid[h] ? m : 0;
            rox[h] = mm % p.OW;
            const int t = mm / p.OW;
            roy[h] = t % p.OH;
            rb[h] = t / p.OH;
        }
    }
    const int ktiles = (p.K + BK - 1) / BK;
    const int tiles_per_tap = VEC ? p.Cin / BK : 1;

    float4 ra[2];
    float rs[8];
    float4 rbv[2];
    float rbs[8];

    auto load_tile = [&](int kt) {
        if (VEC) {
            const int tap = kt / tiles_per_tap, c0 = (kt - tap * tiles_per_tap) * BK + (tid & 3) * 4;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int iy = roy[h] * p.stride + ky - p.pad_t;
                const int ix = rox[h] * p.stride + kx * p.dil - p.pad_l;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rvalid[h] && iy >= 0 && iy < LH && ix >= 0 && ix < LW) {
                    const int py = p.up2 ? iy >> 1 : iy, px = p.up2 ? ix >> 1 : ix;
                    v = *(const float4*)(p.x + (((size_t)rb[h] * p.H + py) * p.W + px) * p.Cin + c0);
                }
                ra[h] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i, ml = e & (BM - 1), kk = e >> 7;
                const int kg = kt * BK + kk, m = m0 + ml;
                float v = 0.f;
                if (kg < p.K && m < p.M) {
                    const int tap = kg / p.Cin, ci = kg - tap * p.Cin;
                    const int ky = tap / p.KW, kx = tap - ky * p.KW;
                    const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
                    const int iy = oy * p.stride + ky - p.pad_t, ix = ox * p.stride + kx * p.dil - p.pad_l;
                    if (iy >= 0 && iy < LH && ix >= 0 && ix < LW) {
                        const int py = p.up2 ? iy >> 1 : iy, px = p.up2 ? ix >> 1 : ix;
                        v = p.x[(((size_t)b * p.H + py) * p.W + px) * p.Cin + ci];
                    }
                }
                rs[i] = v;
            }
        }
        // weights: rows kt*BK .. +BK of [K][Cout]
        if ((p.Cout & 3) == 0) {
            constexpr int TPR = BN / 4;            // threads per k-row
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = tid + 256 * h;
                const int kk = idx / TPR, nq = idx - kk * TPR;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < BK) {
                    const int kg = kt * BK + kk, n = n0 + nq * 4;
                    if (kg < p.K && n < p.Cout) v = *(const float4*)(p.w + (size_t)kg * p.Cout + n);
                }
                rbv[h] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;
                float v = 0.f;
                if (e < BK * BN) {
                    const int kk = e / BN, nl = e - kk * BN;
                    const int kg = kt * BK + kk, n = n0 + nl;
                    if (kg < p.K && n < p.Cout) v = p.w[(size_t)kg * p.Cout + n];
                }
                rbs[i] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        if (VEC) {
            const int kq = (tid & 3) * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ml = (tid >> 2) + 64 * h;
                As[buf][kq + 0][ml] = ra[h].x;
                As[buf][kq + 1][ml] = ra[h].y;
                As[buf][kq + 2][ml] = ra[h].z;
                As[buf][kq + 3][ml] = ra[h].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;
                As[buf][e >> 7][e & (BM - 1)] = rs[i];
            }
        }
        if ((p.Cout & 3) == 0) {
            constexpr int TPR = BN / 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = tid + 256 * h;
                const int kk = idx / TPR, nq = idx - kk * TPR;
                if (kk < BK) *(float4*)&Bs[buf][kk][nq * 4] = rbv[h];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * i;
                if (e < BK * BN) Bs[buf][e / BN][e % BN] = rbs[i];
            }
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) load_tile(kt + 1);
        mma_slab<TC::TM, TC::TN, BN>(As[cur], Bs[cur], wm0, wn0, acc);
        if (kt + 1 < ktiles) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    const int lane = tid & 63, col = lane & 31, rhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TC::TM; ++i)
#pragma unroll
        for (int j = 0; j < TC::TN; ++j) {
            const int n = n0 + wn0 + j * 32 + col;
            if (n >= p.Cout) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.bias_b) v += p.bias_b[(size_t)(m / (p.OH * p.OW)) * p.Cout + n];
                if (p.res) v += p.res[(size_t)m * p.Cout + n];
                p.y[(size_t)m * p.Cout + n] = apply_act(v, p.act, p.act_param);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// strided batched GEMM: C[b] = alpha * A[b] (MxK, row-major, lda) * op(B[b]); op = B (KxN, ldb) or B^T (B is NxK)
// batch index b = (b1, b2) with separate strides so heads can live inside a [B][T][C] tensor.
// ------------------------------------------------------------------------------------------------
struct GemmP {
    const float* a; const float* b; float* c;
    int M, N, K, lda, ldb, ldc, nb2, transB;
    long long sa1, sa2, sb1, sb2, sc1, sc2;
    float alpha;
};

template <int BN>
__global__ __launch_bounds__(256) void k_bgemm(GemmP p) {
    typedef TileCfg<BN> TC;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + LPAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + LPAD];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int wm0 = (wave / TC::WN) * (BM / TC::WM), wn0 = (wave % TC::WN) * (BN / TC::WN);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z - b1 * p.nb2;
    const float* A = p.a + b1 * p.sa1 + b2 * p.sa2;
    const float* Bm = p.b + b1 * p.sb1 + b2 * p.sb2;
    float* Cm = p.c + b1 * p.sc1 + b2 * p.sc2;

    f32x16 acc[TC::TM][TC::TN];
#pragma unroll
    for (int i = 0; i < TC::TM; ++i)
#pragma unroll
        for (int j = 0; j < TC::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ktiles = (p.K + BK - 1) / BK;
    float ra[8], rbq[8];
    // element-wise loaders with bounds checks; k-contiguous operands are read 4 at a time along k by 4 lanes
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {                    // A tile: 128 rows x 16 k ; k fastest across lanes
            const int e = tid + 256 * i, kk = e & (BK - 1), ml = e >> 4;
            const int kg = kt * BK + kk, m = m0 + ml;
            ra[i] = (kg < p.K && m < p.M) ? A[(size_t)m * p.lda + kg] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            float v = 0.f;
            if (e < BK * BN) {
                if (p.transB) {                          // B^T stored [N][K]: k fastest
                    const int kk = e & (BK - 1), nl = e >> 4;
                    const int kg = kt * BK + kk, n = n0 + nl;
                    if (kg < p.K && n < p.N) v = Bm[(size_t)n * p.ldb + kg];
                } else {                                  // B stored [K][N]: n fastest
                    const int nl = e % BN, kk = e / BN;
                    const int kg = kt * BK + kk, n = n0 + nl;
                    if (kg < p.K && n < p.N) v = Bm[(size_t)kg * p.ldb + n];
                }
            }
            rbq[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            As[buf][e & (BK - 1)][e >> 4] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            if (e < BK * BN) {
                if (p.transB) Bs[buf][e & (BK - 1)][e >> 4] = rbq[i];
                else Bs[buf][e / BN][e % BN] = rbq[i];
            }
        }
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) load_tile(kt + 1);
        mma_slab<TC::TM, TC::TN, BN>(As[cur], Bs[cur], wm0, wn0, acc);
        if (kt + 1 < ktiles) store_tile(cur ^ 1);
        __syncthreads();
    }
    const int lane = tid & 63, col = lane & 31, rhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TC::TM; ++i)
#pragma unroll
        for (int j = 0; j < TC::TN; ++j) {
            const int n = n0 + wn0 + j * 32 + col;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                if (m < p.M) Cm[(size_t)m * p.ldc + n] = p.alpha * acc[i][j][r];
            }
        }
}

}  // namespace egr

using namespace egr;

extern "C" int egr_conv_nhwc(const float* x, const float* w, const float* bias, const float* bias_b, const float* res,
                             float* y, int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                             int stride, int dil, int pad_t, int pad_l, int up2, int act, float act_param,
                             void* stream) {
    EGR_CHECK(x && w && y, EGR_ERR_ARG, "null x/w/y");
    EGR_CHECK(B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && OH >= 1 && OW >= 1 && Cout >= 1 && KH >= 1 && KW >= 1 &&
                  stride >= 1 && dil >= 1,
              EGR_ERR_ARG, "bad conv geometry");
    const long long M = (long long)B * OH * OW;
    EGR_CHECK(M < (1LL << 31) && (long long)KH * KW * Cin < (1LL << 31), EGR_ERR_ARG, "conv too large for 32-bit indexing");
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.bias = bias; p.bias_b = bias_b; p.res = res; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW;
    p.stride = stride; p.dil = dil; p.pad_t = pad_t; p.pad_l = pad_l; p.up2 = up2; p.act = act; p.act_param = act_param;
    p.M = (int)M; p.K = KH * KW * Cin;
    const bool vec = (Cin % BK) == 0 && (((uintptr_t)x) & 15) == 0;
    const int bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((Cout + bn - 1) / bn));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(BN_, V_) hipLaunchKernelGGL((k_conv_igemm<BN_, V_>), grid, dim3(256), 0, st, p)
    if (bn == 128) { if (vec) LAUNCH(128, true); else LAUNCH(128, false); }
    else if (bn == 64) { if (vec) LAUNCH(64, true); else LAUNCH(64, false); }
    else { if (vec) LAUNCH(32, true); else LAUNCH(32, false); }
#undef LAUNCH
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_bgemm(const float* a, const float* b, float* c, int nb1, int nb2, int M, int N, int K, int lda,
                         int ldb, int ldc, int64_t sa1, int64_t sa2, int64_t sb1, int64_t sb2, int64_t sc1,
                         int64_t sc2, int transB, float alpha, void* stream) {
    EGR_CHECK(a && b && c && nb1 >= 1 && nb2 >= 1 && M >= 1 && N >= 1 && K >= 1, EGR_ERR_ARG, "bad gemm argument");
    EGR_CHECK((long long)nb1 * nb2 <= 65535, EGR_ERR_ARG, "too many batches");
    GemmP p;
    p.a = a; p.b = b; p.c = c; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.nb2 = nb2;
    p.transB = transB; p.sa1 = sa1; p.sa2 = sa2; p.sb1 = sb1; p.sb2 = sb2; p.sc1 = sc1; p.sc2 = sc2; p.alpha = alpha;
    const int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    dim3 grid((M + BM - 1) / BM, (N + bn - 1) / bn, nb1 * nb2);
    hipStream_t st = (hipStream_t)stream;
    if (bn == 128) hipLaunchKernelGGL((k_bgemm<128>), grid, dim3(256), 0, st, p);
    else if (bn == 64) hipLaunchKernelGGL((k_bgemm<64>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_bgemm<32>), grid, dim3(256), 0, st, p);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}
