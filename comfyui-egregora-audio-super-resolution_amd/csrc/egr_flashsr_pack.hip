// Weight repacking for the FlashSR engine (gfx950): torch-layout fp32 tensors -> the layouts the contraction kernels read.
// Done once per model build, on the device.  Shared by the C-ABI model handle (egr_flashsr.cpp) and the Python graph driver
// (flashsr_engine.py) so that both executors hold bit-identical operands.
#include "egr_common.h"

namespace egr {

// dst [ceil(K/16)][N][16]: dst[s][n][j] = W2[16 s + j][n] (zero beyond K), W2 = the [K][N] GEMM view of src:
//   layout 0  conv / linear  src [N = Co][Ci][KH][KW]      k = (ky KW + kx) Ci + ci
//   layout 1  ConvTranspose1d src [K = Ci][Co][KT]          n = kk Co + co            (N = KT Co)
//   layout 2  per-tap products src [Co][K = Ci][KH][KW]     n = (ky KW + kx) Co + co  (N = KH KW Co)
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ src, float* __restrict__ dst, int layout, int K, int N,
                                                      int Ci, int Co, int KH, int KW) {
    const long long total = (long long)((K + 15) / 16) * N * 16;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = (int)(e & 15);
        const long long t = e >> 4;
        const int n = (int)(t % N);
        const int k = (int)(t / N) * 16 + j;
        float v = 0.f;
        if (k < K) {
            if (layout == 0) {
                const int ci = k % Ci, tap = k / Ci, ky = tap / KW, kx = tap % KW;
                v = src[(((size_t)n * Ci + ci) * KH + ky) * KW + kx];
            } else if (layout == 1) {
                const int co = n % Co, kk = n / Co;
                v = src[((size_t)k * Co + co) * KW + kk];
            } else {
                const int co = n % Co, tap = n / Co, ky = tap / KW, kx = tap % KW;
                v = src[(((size_t)co * Ci + k) * KH + ky) * KW + kx];
            }
        }
        dst[e] = v;
    }
}

// nearest-2x upsample followed by a 3x3 convolution == four 2x2 convolutions on the low-resolution input, one per output phase
// (a, b): the taps that read the same source pixel are pre-summed (float32, ky outer / kx inner, as the first engine did).
// src [Co][Ci][3][3] -> dst [4 = 2a+b][Co][Ci][2][2]
__global__ __launch_bounds__(256) void k_phase_weights(const float* __restrict__ src, float* __restrict__ dst, long long pairs) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (long long)gridDim.x * 256) {
        float g[3][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = src[e * 9 + i];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // rows of the 3x3 kernel that collapse onto source row i of phase a: a = 0: {0}, {1,2}; a = 1: {0,1}, {2}
                        const int ky0 = (a == 0) ? (i == 0 ? 0 : 1) : (i == 0 ? 0 : 2), ky1 = (a == 0) ? (i == 0 ? 0 : 2) : (i == 0 ? 1 : 2);
                        const int kx0 = (b == 0) ? (j == 0 ? 0 : 1) : (j == 0 ? 0 : 2), kx1 = (b == 0) ? (j == 0 ? 0 : 2) : (j == 0 ? 1 : 2);
                        float w = 0.f;
                        for (int ky = ky0; ky <= ky1; ++ky)
                            for (int kx = kx0; kx <= kx1; ++kx) w = __fadd_rn(w, g[ky][kx]);
                        dst[(((size_t)(2 * a + b) * pairs + e) * 2 + i) * 2 + j] = w;
                    }
    }
}

// U = G g G^T per (co, ci) pair in double, rounded once; written as n*n matrices [Ci][Co], each in the slab-major pack:
// dst[(i n + j) * zfloats + (ci / 16) * Co * 16 + co * 16 + ci % 16].  dst must be zeroed first (K padding).
template <int NP>
__global__ __launch_bounds__(256) void k_winograd_u(const float* __restrict__ src, float* __restrict__ dst, const double* __restrict__ G,
                                                     int Co, int Ci, long long zfloats) {
    const long long pairs = (long long)Co * Ci;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (long long)gridDim.x * 256) {
        const int co = (int)(e / Ci), ci = (int)(e % Ci);
        double g[3][3], t[NP][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = (double)src[e * 9 + i];
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int l = 0; l < 3; ++l) t[i][l] = G[i * 3 + 0] * g[0][l] + G[i * 3 + 1] * g[1][l] + G[i * 3 + 2] * g[2][l];
        const size_t off = (size_t)(ci / 16) * Co * 16 + (size_t)co * 16 + (ci % 16);
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double u = t[i][0] * G[j * 3 + 0] + t[i][1] * G[j * 3 + 1] + t[i][2] * G[j * 3 + 2];
                dst[(size_t)(i * NP + j) * zfloats + off] = (float)u;
            }
    }
}

}  // namespace egr

using namespace egr;

static inline unsigned grid_for(long long total) {
    long long nb = (total + 255) / 256;
    return (unsigned)(nb < 1 ? 1 : (nb > 65535 ? 65535 : nb));
}

extern "C" int egr_pack_weight(const float* src, float* dst, int layout, int K, int N, int Ci, int Co, int KH, int KW, void* stream) {
    EGR_CHECK(src && dst && layout >= 0 && layout <= 2 && K >= 1 && N >= 1, EGR_ERR_ARG, "egr_pack_weight: bad argument");
    const long long total = (long long)((K + 15) / 16) * N * 16;
    hipLaunchKernelGGL(k_pack_weight, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, layout, K, N, Ci, Co, KH, KW);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_phase_weights(const float* w_oihw, float* dst4, int Co, int Ci, void* stream) {
    EGR_CHECK(w_oihw && dst4 && Co >= 1 && Ci >= 1, EGR_ERR_ARG, "egr_phase_weights: bad argument");
    const long long pairs = (long long)Co * Ci;
    hipLaunchKernelGGL(k_phase_weights, dim3(grid_for(pairs)), dim3(256), 0, (hipStream_t)stream, w_oihw, dst4, pairs);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// F(2x2,3x3) (np = 4) or F(4x4,3x3) (np = 6): G given by the caller as np x 3 doubles on the DEVICE.
extern "C" int egr_winograd_pack_u(const float* w_oihw, float* dst, const double* G_dev, int np, int Co, int Ci, void* stream) {
    EGR_CHECK(w_oihw && dst && G_dev && (np == 4 || np == 6) && Co >= 1 && Ci >= 1, EGR_ERR_ARG, "egr_winograd_pack_u: bad argument");
    const long long zfloats = (long long)((Ci + 15) / 16) * Co * 16;
    EGR_HIP(hipMemsetAsync(dst, 0, (size_t)np * np * zfloats * sizeof(float), (hipStream_t)stream));
    const long long pairs = (long long)Co * Ci;
    if (np == 4)
        hipLaunchKernelGGL(k_winograd_u<4>, dim3(grid_for(pairs)), dim3(256), 0, (hipStream_t)stream, w_oihw, dst, G_dev, Co, Ci, zfloats);
    else
        hipLaunchKernelGGL(k_winograd_u<6>, dim3(grid_for(pairs)), dim3(256), 0, (hipStream_t)stream, w_oihw, dst, G_dev, Co, Ci, zfloats);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}
