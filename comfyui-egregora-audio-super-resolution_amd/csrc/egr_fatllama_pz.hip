// Paired chirp-z: the Fat-Llama loop for every length the packed real transform cannot take (odd N, N/2 with a prime
// factor above 13 -- most real files; reference: whole-file transform, no length restriction,
// egregora_fat_llama_gpu.py:272-288).
//
// The exact length-N real DFT is reduced to ONE complex DFT of length D per state before the chirp:
//   kind 1 (N even)  z[m] = d[2m] + i d[2m+1], D = N/2, one state per channel           (even/odd packing, as the packed plans)
//   kind 2 (N odd)   z[m] = d_a[m] + i d_b[m], D = N,   one state per channel PAIR      (two real signals in one complex one)
// and the length-D DFT is Bluestein's cyclic convolution of length P >= 2D - 1 (P = L x nc, smooth):
//   Z[k] = w[k] (a (*) b)[k],  a[n] = z[n] w[n],  b[j] = conj(w[j]) (|j| < D),  w[n] = exp(-i pi n^2 / D) = W_2D^(n^2).
// Half the points of the full-complex form (k_colz in egr_fatllama.hip: P >= 2N - 1 per channel).  Both kinds need the
// spectrum in PAIRS (k, D - k): X[k] = E[k] + W_N^k O[k] with E, O from Z[k] and conj Z[D - k] (kind 1) or A[k] = E[k], B[k] = O[k]
// (kind 2).  The convolution output sits in natural order only inside the strided column pass, where index p = i nc + c lives
// in column c: the partner D - p lives in the mirrored column, so a workgroup of the spectrum pass owns a tile of TC columns
// AND its mirror tile (k_pzpair).  Index 0 is stored at position s (a shift of the whole convolution, s < TC) chosen so that
// the tiles of a pair are whole aligned tiles when D is odd.
//
// The inverse DFT uses the conjugate chirps, z[m] = conj(w[m]) / D sum_k (Z[k] conj(w[k])) w[m - k]: convolution with conj(b),
// i.e. a multiplication by conj(Bhat) (b is even), and the next forward input is a[m] = z[m] w[m] = (a' (*) conj b)[m] / D --
// the time side of an iteration is a CROP of the convolution output (positions outside [s, s + D) zeroed), no chirp at all.
// One iteration = 4 launches over the P-point state:
//   k_pz_rowconv<false>   row FFT . x Bhat . row IFFT
//   k_pzpair              twiddle^-1 . column IFFT . [Z = w c ; pair hook ; a' = Z' conj(w) / D] . column FFT . twiddle
//   k_pz_rowconv<true>    row FFT . x conj(Bhat) . row IFFT
//   k_pzcol<1>            twiddle^-1 . column IFFT . crop . column FFT . twiddle
// The pair hook runs in double precision (the chirp values are double table products, so |w|^2 = 1 to 1e-16 and nothing
// compounds over the iterations); Bhat = FFT_P(b) / P is computed once per plan by a double-precision transform on the device
// and rounded to float once.
#ifndef EGR_BFLY_HILO
#define EGR_BFLY_HILO 15          // the register butterflies' cos / sin constants as two floats (egr_fft_device.h), as in egr_fatllama.hip
#endif
#include "egr_fatllama_int.h"

namespace egr {

struct PzP {
    FftDesc f;                  // schedule of the L-point column transform
    int L, nc;                  // state [L][nc] complex, row-major; position p = i nc + c
    int TC, TClog2;             // columns per tile
    int ntiles, tiles_per_xcd;  // single tiles (k_pzcol)
    int G, g_per_xcd;           // tile pairs (k_pzpair)
    const cplx* tw;
    const dcplx* twd;           // W_L stage tables
    const cplx* stw;            // butterfly-ordered stage tables of a compile-time column schedule (PzSched)
    Tw2 big;                    // W_P^r (four-step twiddle)
    Tw2 w;                      // W_(2D)^r (chirp and real-split twiddle)
    unsigned long long D;       // transform length
    double inv_2D, inv_D;
    int s;                      // position of index 0
    int kind;                   // 1: even/odd packing, 2: channel pair
    int odd;                    // D odd
    int c0;                     // first right column (a multiple of TC)
    int iDr, cDr;               // Dr = D + 2 s = iDr nc + cDr: the partner of position p is Dr - p
    long long P;
    long long N;                // real samples per channel
    int C;                      // channels
};

struct PzHook {
    float thr;
    int soft;                   // soft shrink instead of the hard threshold
    int band;                   // keep bins >= band_lo instead of thresholding (egr_band_filter)
    unsigned long long band_lo;
    const unsigned* max2;       // relative threshold: max |X|^2 of this iteration per channel (slots of fl_max2_read)
    unsigned* max2_out;         // MAXONLY: where those maxima go
    unsigned* max2_next;        // carried maximum: max |S(X)|^2 of this iteration = what the next iteration's spectrum holds (RowP)
    unsigned* max2_zero;        // ring slot two iterations ahead, cleared by workgroup 0 of every state
};

__device__ __forceinline__ dcplx pz_chirp(const PzP& p, unsigned long long k) {       // w[k] = W_(2D)^(k^2 mod 2D)
    const unsigned long long m = 2ULL * p.D, k2 = k * k;
    unsigned long long r;
    if (k2 < (1ULL << 53)) {          // exact in double: quotient estimate off by at most one (see chirp() in egr_fatllama_int.h)
        const unsigned long long q = (unsigned long long)((double)k2 * p.inv_2D);
        long long d = (long long)(k2 - q * m);
        if (d < 0) d += (long long)m;
        if (d >= (long long)m) d -= (long long)m;
        r = (unsigned long long)d;
    } else {
        r = k2 % m;
    }
    return tw2d(p.w, (unsigned)r);
}

// dev ablations (wrong results): EGR_PZ_ABL_NOTW drops the four-step twiddles, _NOFFT the in-LDS transforms, _NOHOOK the pair hook
// Four-step twiddles of one thread's tile elements.  A thread's elements e = tid + k T (k < NE) sit in ONE column (T is a multiple of
// the tile width) at rows i0 + k di, so their twiddles W_P^(col (i0 + k di)) are a geometric run: two double table products per
// thread (first value, ratio) and one double multiply per element instead of a table product per element, rounded to float once
// each; the eight values stay in registers from the tile's load to its store.
template <int NE> struct PzTwRun {
    cplx w[NE];
    __device__ __forceinline__ void init(const PzP& p, int col, int i0, int di) {
#ifdef EGR_PZ_ABL_NOTW
#pragma unroll
        for (int k = 0; k < NE; ++k) w[k] = make_float2(1.f, 0.f);
#else
        if (i0 >= p.L) i0 = 0;                       // a thread past the tile has no elements: keep its table index in range
        dcplx cur = tw2d(p.big, (unsigned)col * (unsigned)i0);
        const dcplx st = tw2d(p.big, (unsigned)(((unsigned long long)col * (unsigned long long)di) % (unsigned long long)p.P));
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            w[k] = make_float2((float)cur.x, (float)cur.y);
            cur = dcmul(cur, st);
        }
#endif
    }
};
__device__ __forceinline__ dcplx dmulf(dcplx w, cplx c) {          // w * c
    return make_double2(w.x * (double)c.x - w.y * (double)c.y, w.x * (double)c.y + w.y * (double)c.x);
}
__device__ __forceinline__ cplx dmulc_f(dcplx a, dcplx w, double sc) {    // a * conj(w) * sc, rounded once
    return make_float2((float)(sc * (a.x * w.x + a.y * w.y)), (float)(sc * (a.y * w.x - a.x * w.y)));
}

// Round 6: the four-step twiddle products of the thread-per-(row class, column) kernels in double (the run `cur` is double anyway): a
// float-rounded twiddle is a gain |W|^2 - 1 ~ 6e-8 applied in the same direction every iteration (egr_fatllama_wl.h EGR_WL_HILO), and the
// register butterflies' constants as two floats (EGR_BFLY_HILO below) -- N = 2 880 002, 800 iterations against the float64 loop:
// max 2.91 -> 1.00, rms 0.635 -> 0.193 (the float32 Bluestein oracle: 2.70 / 0.543), plain LSD 6.8e-3 -> 4.2e-3 dB, 103.7 -> 108.4 ms per stereo stage; the
// odd length (channel pairs) rms 0.490 -> 0.156, 150.3 -> 155.9 ms (profiles/r06/chirpz_precision.txt: the butterfly constants are nine tenths of it)
#ifndef EGR_PZ_TWD
#define EGR_PZ_TWD 1
#endif
__device__ __forceinline__ cplx pz_tw_mul(cplx v, dcplx w) {         // v * w
    if (EGR_PZ_TWD) { const double x = (double)v.x, y = (double)v.y; return make_float2((float)(x * w.x - y * w.y), (float)(x * w.y + y * w.x)); }
    return cmul(v, make_float2((float)w.x, (float)w.y));
}
__device__ __forceinline__ cplx pz_tw_mulc(cplx v, dcplx w) {        // v * conj(w)
    if (EGR_PZ_TWD) { const double x = (double)v.x, y = (double)v.y; return make_float2((float)(x * w.x + y * w.y), (float)(y * w.x - x * w.y)); }
    return cmulc(v, make_float2((float)w.x, (float)w.y));
}
#ifndef EGR_PZ_TWPOW
#define EGR_PZ_TWPOW true
#endif
// Transform policies of the column kernels: run() transforms `ncols` interleaved columns in place (element i of column t at
// i * ncols + t).  PzRt: the run-time radix schedule of the plan (any length; FUSE = all columns in one call, else the two halves
// one after the other -- tile pairs above 8 elements per thread).  PzSched: ONE compile-time schedule R0 R1 R2 (R3) for NCOLS
// columns (egr_fft_device.h lds_fft_sched_inplace: butterfly-ordered float stage tables, no radix dispatch, registers sized for
// this schedule only); the workgroup size is part of the type.
template <bool FUSE> struct PzRt {
    static constexpr int MAXT = 1024;
    static constexpr int NE1 = 8, NE2 = FUSE ? 8 : 16;       // elements per thread at most: one tile, a tile pair
    static constexpr int MINW1 = 1, MINW2 = 1;
    static __device__ __forceinline__ void run(cplx* cur, const PzP& p, int ncols, int lg, bool inverse) {
#ifdef EGR_PZ_ABL_NOFFT
        return;
#endif
        if (FUSE) {
            lds_fft_ip<true, 0, EGR_PZ_TWPOW>(cur, p.f, p.tw, ncols, lg, ncols, 1, inverse, p.twd);
        } else {
            lds_fft_ip<true, 0, EGR_PZ_TWPOW>(cur, p.f, p.tw, ncols / 2, lg - 1, ncols, 1, inverse, p.twd);
            lds_fft_ip<true, 0, EGR_PZ_TWPOW>(cur + ncols / 2, p.f, p.tw, ncols / 2, lg - 1, ncols, 1, inverse, p.twd);
        }
    }
};
constexpr int pz_min(int a, int b) { return a < b ? a : b; }
constexpr int pz_sched_threads(int ltot, int rmin) {     // one butterfly of the smallest radix per thread, two above 1024 threads ...
    const int bf = ltot / rmin;
    const int nbt = (bf + 1023) / 1024;
    return ((bf + nbt - 1) / nbt + 63) / 64 * 64;
}
template <int NCOLS, int R0, int R1, int R2, int R3 = 1> struct PzSched {
    static constexpr int L = R0 * R1 * R2 * R3;
    static constexpr int LTOT = L * NCOLS;
    static constexpr int RMIN = pz_min(pz_min(R0, R1), pz_min(R2, R3 > 1 ? R3 : R2));
    static constexpr int THREADS = pz_sched_threads(LTOT, RMIN);
    static constexpr int MAXT = THREADS;
    static constexpr int NE1 = (LTOT + THREADS - 1) / THREADS, NE2 = NE1;
    // register budget (waves per SIMD the kernel must leave room for): four single-tile workgroups or two tile-pair workgroups per CU
    static constexpr int WAVES = THREADS / 64;
    static constexpr int MINW1 = WAVES < 6 ? WAVES : 6, MINW2 = WAVES <= 12 ? (WAVES + 1) / 2 : 4;
    static __device__ __forceinline__ void run(cplx* cur, const PzP& p, int ncols, int lg, bool inverse) {
#ifdef EGR_PZ_ABL_NOFFT
        return;
#endif
        lds_fft_sched_inplace<true, 0, LTOT, THREADS, R0, R1, R2, R3>(cur, L, p.stw, ncols, lg, ncols, 1, inverse);
    }
};

// Single-tile column pass.
// MODE 0: first (y -> time threshold -> z -> a = z w -> FFT -> twiddle)
// MODE 1: crop  (twiddle^-1 -> IFFT -> positions outside [s, s + D) zeroed -> FFT -> twiddle)
// MODE 2: last  (twiddle^-1 -> IFFT -> z = conj(w) c -> out = y + d, per-channel peak)
// thr_rel (MODE 0, optional): per-channel max|y| as float bits; the level becomes thr * max|y|.
template <int MODE, class F>
__global__ __launch_bounds__(F::MAXT, F::MINW1) void k_pzcol(PzP p, float thr, cplx* __restrict__ work, float* __restrict__ out,
                                                unsigned* __restrict__ peak_out, const unsigned* __restrict__ thr_rel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    __shared__ float red[16];
    const int tile = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int st = blockIdx.y;
    const int TC = p.TC, lg = p.TClog2, L = p.L, nc = p.nc;
    const int c0 = tile * TC;
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* W = work + (size_t)st * p.P;
    const int nel = L * TC;
    const unsigned long long D = p.D;
    const long long s = p.s;
    // channels of this state
    const int cha = p.kind == 1 ? st : 2 * st, chb = p.kind == 1 ? st : 2 * st + 1;
    const bool hasb = chb < p.C;
    float* Ya = out + (size_t)cha * p.N;
    float* Yb = out + (size_t)(hasb ? chb : cha) * p.N;
    // this thread's column and its twiddle run (the workgroup size is a multiple of TC, at most 8 elements per thread)
    const int tcol = c0 + ((int)threadIdx.x & (TC - 1));
    PzTwRun<F::NE1> twr;
    twr.init(p, tcol < nc ? tcol : 0, (int)threadIdx.x >> lg, (int)blockDim.x >> lg);

    if (MODE == 0) {
        float ta = thr, tb = thr;
        if (thr_rel) { ta = thr * __uint_as_float(thr_rel[cha]); tb = thr * __uint_as_float(thr_rel[hasb ? chb : cha]); }
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            const long long m = (long long)i * nc + col - s;
            cplx v = make_float2(0.f, 0.f);
            if (col < nc && m >= 0 && (unsigned long long)m < D) {
                float2 y;
                if (p.kind == 1) y = ((const float2*)Ya)[m];
                else y = make_float2(Ya[m], hasb ? Yb[m] : 0.f);
                y.x = fabsf(y.x) > ta ? y.x : 0.f;
                y.y = fabsf(y.y) > tb ? y.y : 0.f;
                const dcplx a = dmulf(pz_chirp(p, (unsigned long long)m), y);
                v = make_float2((float)a.x, (float)a.y);
            }
            cur[e] = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < F::NE1; ++k) {
            const int e = threadIdx.x + k * blockDim.x;
            if (e < nel) {
                const int i = e >> lg;
                cur[e] = tcol < nc ? cmulc(W[(size_t)i * nc + tcol], twr.w[k]) : make_float2(0.f, 0.f);
            }
        }
    }
    __syncthreads();
    if (MODE != 0) F::run(cur, p, TC, lg, true);
    if (MODE == 2) {
        float mxa = 0.f, mxb = 0.f;
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            const long long m = (long long)i * nc + col - s;
            if (col < nc && m >= 0 && (unsigned long long)m < D) {
                const dcplx w = pz_chirp(p, (unsigned long long)m);
                const cplx cc = cur[e];
                // z = conj(w) c (the 1 / D went into the spectrum hook)
                const float zx = (float)(w.x * (double)cc.x + w.y * (double)cc.y);
                const float zy = (float)(w.x * (double)cc.y - w.y * (double)cc.x);
                if (p.kind == 1) {
                    float2 y = ((const float2*)Ya)[m];
                    y.x = __fadd_rn(y.x, zx);
                    y.y = __fadd_rn(y.y, zy);
                    ((float2*)Ya)[m] = y;
                    mxa = fmaxf(mxa, fmaxf(fabsf(y.x), fabsf(y.y)));
                } else {
                    const float oa = __fadd_rn(Ya[m], zx);
                    Ya[m] = oa;
                    mxa = fmaxf(mxa, fabsf(oa));
                    if (hasb) {
                        const float ob = __fadd_rn(Yb[m], zy);
                        Yb[m] = ob;
                        mxb = fmaxf(mxb, fabsf(ob));
                    }
                }
            }
        }
        mxa = block_max(mxa, red);
        if (p.kind == 2 && hasb) mxb = block_max(mxb, red);
        if (threadIdx.x == 0) {
            atomic_max_abs(peak_out + cha, mxa);
            if (p.kind == 2 && hasb) atomic_max_abs(peak_out + chb, mxb);
        }
        return;
    }
    if (MODE == 1) {
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            const long long m = (long long)i * nc + col - s;
            if (m < 0 || (unsigned long long)m >= D) cur[e] = make_float2(0.f, 0.f);
        }
        __syncthreads();
    }
    F::run(cur, p, TC, lg, false);
#pragma unroll
    for (int k = 0; k < F::NE1; ++k) {
        const int e = threadIdx.x + k * blockDim.x;
        if (e < nel && tcol < nc) W[(size_t)(e >> lg) * nc + tcol] = cmul(cur[e], twr.w[k]);
    }
}


// The crop pass (k_pzcol<1>) with two workgroup barriers -- the scheme of k_col_wl (egr_fatllama_wl.h) for columns of L = LA x LB
// points, tiles of 8 columns, one thread per (row class, column):
//   thread (b < LB, col) loads its LA elements i = LB a + b straight from global memory (64-byte row segments per 8 lanes),
//   x conj W_P^(col i) (a geometric run in double), inverse radix-LA over a, LDS transpose (barrier), thread (c < LA, col):
//   x conj W_L^(b c), inverse radix-LB over b -> the time-domain points n = c + LA d, crop (positions outside [s, s + D) zeroed) in
//   registers -- and those are the inputs n = LA a' + b' (b' = c) of its forward radix-LB over a': no exchange between the inverse
//   and the forward transform.  x W_L^(b' c''), written into the LDS row it has just read, barrier, thread (c'' < LB, col): forward
//   radix-LA over b', x W_P^(col i), store to the addresses it loaded from.  LDS rows are padded to an odd length (conflict-free
//   strided accesses); 4 LDS passes per element instead of 14, 2 barriers instead of 13.
template <int LA, int LB> struct PzColWl {
    static constexpr int T = LA > LB ? LA : LB;
    static constexpr int TC = 8;
    static constexpr int THREADS = ((T * TC + 63) / 64) * 64;
    static constexpr int LBP = LB | 1;
    static constexpr int LDS = LA * LBP * TC * 8;
};
template <int LA, int LB>
__global__ __launch_bounds__((PzColWl<LA, LB>::THREADS)) void k_pzcol_wl(PzP p, const cplx* __restrict__ tab, int ntiles, int tiles_per_xcd, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    using G = PzColWl<LA, LB>;
    constexpr int TC = G::TC, LBP = G::LBP, L = LA * LB;
    __shared__ cplx ts[L];
    const int tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int st = blockIdx.y;
    const int nc = p.nc;
    cplx* lds = (cplx*)EGR_LDS_BASE(smem);
    const int tid = threadIdx.x;
    const int b = tid >> 3, cl = tid & 7, col = tile * TC + cl;
    cplx* W = work + (size_t)st * p.P + col;
    for (int e = tid; e < L; e += G::THREADS) ts[e] = tab[e];
    const unsigned long long D = p.D;
    const long long s = p.s;
    // four-step twiddles W_P^(col (LB a + b)) of this thread's rows (threads b < LB own rows; the same run serves load and store)
    dcplx w0 = make_double2(1.0, 0.0), wst = w0;
    cplx v[LA];
    if (b < LB) {
#pragma unroll
        for (int a = 0; a < LA; ++a) v[a] = W[(size_t)(LB * a + b) * nc];
        w0 = tw2d(p.big, (unsigned)col * (unsigned)b);
        wst = tw2d(p.big, (unsigned)col * (unsigned)LB);
        dcplx cur = w0;
#pragma unroll
        for (int a = 0; a < LA; ++a) {
            v[a] = pz_tw_mulc(v[a], cur);
            cur = dcmul(cur, wst);
        }
        wl_bfly_inv<LA>(v);                              // v[c] = Z[c][b]
#pragma unroll
        for (int c = 0; c < LA; ++c) lds[(c * LBP + b) * TC + cl] = v[c];
    }
    __syncthreads();
    if (b < LA) {
        const int c = b;
        cplx u[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const cplx z = lds[(c * LBP + j) * TC + cl];
            u[j] = (j && c) ? cmulc(z, ts[j * LA + c]) : z;
        }
        wl_bfly_inv<LB>(u);                              // u[d] = t[c + LA d]
#pragma unroll
        for (int d = 0; d < LB; ++d) {
            const long long m = (long long)(c + LA * d) * nc + col - s;
            if (m < 0 || (unsigned long long)m >= D) u[d] = make_float2(0.f, 0.f);
        }
        Bfly<LB>::run(u);                                // forward over a' = d: u[c''] = Z2[c''][b' = c]
#pragma unroll
        for (int j = 0; j < LB; ++j) lds[(c * LBP + j) * TC + cl] = (j && c) ? cmul(u[j], ts[j * LA + c]) : u[j];
    }
    __syncthreads();
    if (b < LB) {
#pragma unroll
        for (int j = 0; j < LA; ++j) v[j] = lds[(j * LBP + b) * TC + cl];          // Z2[c'' = b][b' = j]
        Bfly<LA>::run(v);                                // v[d''] = X[c'' + LB d'']
        dcplx cur = w0;
#pragma unroll
        for (int a = 0; a < LA; ++a) {
            W[(size_t)(LB * a + b) * nc] = pz_tw_mul(v[a], cur);
            cur = dcmul(cur, wst);
        }
    }
}
typedef void (*PzColWlFn)(PzP, const cplx*, int, int, cplx*);
struct PzColWlEntry { int L, la, lb, threads, lds; PzColWlFn fn; };
#define PZ_COLWL_LIST(X) X(512, 16, 32) X(560, 20, 28) X(600, 24, 25) X(640, 20, 32) X(672, 24, 28) X(720, 24, 30) X(768, 24, 32) X(800, 25, 32) \
    X(840, 28, 30) X(900, 30, 30) X(960, 30, 32) X(1024, 32, 32)
#define PZ_COLWL_ENTRY(LEN, A, B) {LEN, A, B, PzColWl<A, B>::THREADS, PzColWl<A, B>::LDS, k_pzcol_wl<A, B>},
static const PzColWlEntry kPzColWl[] = {PZ_COLWL_LIST(PZ_COLWL_ENTRY)};

// Carried maximum of the relative threshold (PzHook::max2_next): per-wave maxima to LDS in front of a barrier the kernel has anyway,
// ONE commit per workgroup and channel behind it; workgroup g = 0 of a state clears the ring slot two iterations ahead.
// the carried maxima of this state's channel(s), requested at the kernel's top (an L2 miss -- the slots were written by memory-side atomics --
// that the inverse transform in front of the hook covers)
__device__ __forceinline__ void pz_max_prefetch(const PzP& p, const PzHook& h, int st, float& m2a, float& m2b) {
    m2a = m2b = 0.f;
    if (!h.max2) return;
    const int cha = p.kind == 1 ? st : 2 * st, chb = p.kind == 1 ? st : 2 * st + 1;
    m2a = fl_max2_read(h.max2, cha);
    m2b = p.kind == 1 ? m2a : fl_max2_read(h.max2, chb < p.C ? chb : cha);
}
__device__ __forceinline__ void pz_max_stage(const PzHook& h, float mxa, float mxb, float* red) {
    if (!h.max2_next) return;
    mxa = wave_max(mxa); mxb = wave_max(mxb);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mxa; red[16 + (threadIdx.x >> 6)] = mxb; }
}
__device__ __forceinline__ void pz_max_commit(const PzP& p, const PzHook& h, int st, int g, const float* red) {
    const int cha = p.kind == 1 ? st : 2 * st, chb = p.kind == 1 ? st : 2 * st + 1;
    const bool hasb = p.kind == 2 && chb < p.C;
    if (h.max2_zero && g == 0 && threadIdx.x < EGR_FL_MAX_SUB) {
        fl_max2_clear(h.max2_zero, cha, threadIdx.x);
        if (hasb) fl_max2_clear(h.max2_zero, chb, threadIdx.x);
    }
    if (!h.max2_next || threadIdx.x != 0) return;
    float ra = red[0], rb = red[16];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) { ra = fmaxf(ra, red[i]); rb = fmaxf(rb, red[16 + i]); }
    fl_max2_commit(h.max2_next, cha, ra);
    if (hasb) fl_max2_commit(h.max2_next, chb, rb);
}

// The pair hook of the spectrum pass on natural-order positions p = i nc + c of a tile pair held in LDS; index k = p - s is valid
// for 0 <= k < D.  at(i, slot): element i of right column t at slot t, of left column t at slot TC + t.  Shared by k_pzpair (stages
// in place, interleaved tile) and k_pzpair_wl (rows of the thread-per-(row class, column) layout).
// m2a / m2b: max |X|^2 of this iteration's spectrum for the state's channel(s) (relative threshold; pz_max_prefetch at the kernel's top)
template <bool MAXONLY, class AT>
__device__ __forceinline__ void pz_pair_hook(const PzP& p, const PzHook& h, AT at, int st, int rs, unsigned rmask, unsigned selfmask,
                                             float& mxa, float& mxb, float m2a = 0.f, float m2b = 0.f) {
    const int TC = p.TC, lg = p.TClog2, L = p.L, nc = p.nc;
    const unsigned long long D = p.D;
    const long long s = p.s;
    // ---- pair hook on natural-order positions p = i nc + c; index k = p - s is valid for 0 <= k < D ----
    double ta = (double)h.thr, tb = (double)h.thr;
    if (!MAXONLY && h.max2) {
        ta = (double)(h.thr * sqrtf(m2a));
        tb = p.kind == 1 ? ta : (double)(h.thr * sqrtf(m2b));
    }
    const double ta2 = ta * ta, tb2 = tb * tb;
    const double sgn = p.odd ? -1.0 : 1.0;          // w[D - k] = (-1)^D w[k]
    const int nel = L * TC;
#ifdef EGR_PZ_ABL_NOHOOK
    if (false)
#endif
    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        const int t = e & (TC - 1), i = e >> lg;
        if (!((rmask >> t) & 1u)) continue;
        const int c = rs + t;
        const long long pos = (long long)i * nc + c;
        const long long k = pos - s;
        const bool selfcol = (selfmask >> t) & 1u;
        cplx* pa = at(i, t);                        // this element
        cplx* pb;                                   // its partner
        bool same = false, valid = k >= 0 && (unsigned long long)k < D;
        unsigned long long kk = valid ? (unsigned long long)k : 0ULL;
        if (valid) {
            if (k == 0) {
                pb = pa; same = true;               // index 0 pairs with itself (its alias D is not stored)
            } else {
                const int ip = p.iDr - i - (c > p.cDr ? 1 : 0);
                if (selfcol) {
                    if (i > ip) continue;           // handled by the partner
                    pb = at(ip, t);
                    same = i == ip;
                } else {
                    pb = at(ip, TC + (TC - 1 - t));
                }
            }
        } else if ((unsigned long long)k == D && !selfcol) {
            // position s + D mirrors position s: index 0 sits in the left tile; handle it from here
            const int ip = p.iDr - i - (c > p.cDr ? 1 : 0);
            cplx* p0 = at(ip, TC + (TC - 1 - t));
            if (!MAXONLY) *pa = make_float2(0.f, 0.f);
            pa = p0; pb = p0; same = true; valid = true; kk = 0ULL;
        } else {
            if (!MAXONLY) *pa = make_float2(0.f, 0.f);
            continue;
        }
        const dcplx w = pz_chirp(p, kk);
        const dcplx Za = dmulf(w, *pa);
        dcplx Zb;
        if (same) Zb = Za;
        else { Zb = dmulf(w, *pb); Zb.x *= sgn; Zb.y *= sgn; }
        // E = (Za + conj Zb) / 2 ; O = (Za - conj Zb) / (2i)
        const dcplx E = make_double2(0.5 * (Za.x + Zb.x), 0.5 * (Za.y - Zb.y));
        const dcplx O = make_double2(0.5 * (Za.y + Zb.y), -0.5 * (Za.x - Zb.x));
        dcplx Xk, Xm, Wk = make_double2(1.0, 0.0);
        if (p.kind == 1) {
            Wk = tw2d(p.w, (unsigned)kk);           // W_N^k = W_(2D)^k
            const dcplx WO = dcmul(Wk, O);
            Xk = make_double2(E.x + WO.x, E.y + WO.y);      // X[k]
            Xm = make_double2(E.x - WO.x, E.y - WO.y);      // conj X[D - k]
        } else {
            Xk = E;                                 // A[k]: channel a
            Xm = O;                                 // B[k]: channel b
        }
        const double ma2 = Xk.x * Xk.x + Xk.y * Xk.y, mb2 = Xm.x * Xm.x + Xm.y * Xm.y;
        if (MAXONLY) {
            if (p.kind == 1) mxa = fmaxf(mxa, fmaxf((float)ma2, (float)mb2));
            else { mxa = fmaxf(mxa, (float)ma2); mxb = fmaxf(mxb, (float)mb2); }
            continue;
        }
        if (h.band) {
            if (p.kind == 1) {
                if (kk < h.band_lo) Xk = make_double2(0.0, 0.0);
                if (D - kk < h.band_lo) Xm = make_double2(0.0, 0.0);
            } else if ((kk < D - kk ? kk : D - kk) < h.band_lo) {
                Xk = make_double2(0.0, 0.0);
                Xm = make_double2(0.0, 0.0);
            }
        } else if (h.soft) {
            const double ga = ma2 > ta2 ? 1.0 - ta / sqrt(ma2) : 0.0, gb = mb2 > tb2 ? 1.0 - tb / sqrt(mb2) : 0.0;
            Xk.x *= ga; Xk.y *= ga; Xm.x *= gb; Xm.y *= gb;
        } else {
            if (!(ma2 > ta2)) Xk = make_double2(0.0, 0.0);
            if (!(mb2 > tb2)) Xm = make_double2(0.0, 0.0);
        }
        if (h.max2_next) {                      // max |S(X)|^2: the spectrum of the next iteration
            const float na2 = (float)(Xk.x * Xk.x + Xk.y * Xk.y), nb2 = (float)(Xm.x * Xm.x + Xm.y * Xm.y);
            if (p.kind == 1) mxa = fmaxf(mxa, fmaxf(na2, nb2));
            else { mxa = fmaxf(mxa, na2); mxb = fmaxf(mxb, nb2); }
        }
        dcplx E2, O2;
        if (p.kind == 1) {
            E2 = make_double2(0.5 * (Xk.x + Xm.x), 0.5 * (Xk.y + Xm.y));
            const dcplx H = make_double2(0.5 * (Xk.x - Xm.x), 0.5 * (Xk.y - Xm.y));
            O2 = make_double2(H.x * Wk.x + H.y * Wk.y, H.y * Wk.x - H.x * Wk.y);        // H conj(Wk)
        } else {
            E2 = Xk; O2 = Xm;
        }
        // Za' = E2 + i O2 ; Zb' = conj(E2 - i O2) ; a' = Z' conj(w) / D
        const dcplx Za2 = make_double2(E2.x - O2.y, E2.y + O2.x);
        *pa = dmulc_f(Za2, w, p.inv_D);
        if (!same) {
            const dcplx Zb2 = make_double2(E2.x + O2.y, -(E2.y - O2.x));
            *pb = dmulc_f(Zb2, w, sgn * p.inv_D);
        }
    }
}

// Spectrum pass on a tile pair: right tile = columns rs .. rs + TC - 1, left tile = their mirror images (cDr - c) mod nc in
// ascending order.  LDS holds both tiles interleaved: element i of right column t at i * 2TC + t, of left column t at
// i * 2TC + TC + t.  MAXONLY: no write-back; max |X|^2 per channel -> h.max2_out (the relative threshold's reduction).
template <bool MAXONLY, class F>
__global__ __launch_bounds__(F::MAXT, F::MINW2) void k_pzpair(PzP p, PzHook h, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    __shared__ float red[32];
    const int g = (blockIdx.x & 7) * p.g_per_xcd + (blockIdx.x >> 3);
    if (g >= p.G) return;
    const int st = blockIdx.y;
    const int TC = p.TC, lg = p.TClog2, L = p.L, nc = p.nc, TC2 = 2 * TC;
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* W = work + (size_t)st * p.P;
    float m2a, m2b;
    pz_max_prefetch(p, h, st, m2a, m2b);
    const unsigned long long D = p.D;
    const long long s = p.s;
    const int rs = (p.c0 + g * TC) % nc;
    int ls = (p.cDr - rs - TC + 1) % nc;
    if (ls < 0) ls += nc;
    // column masks: with an integer reflection centre (D even) column c0 and column c0 + nc / 2 are their own mirror images
    unsigned rmask = (1u << TC) - 1u, lmask = rmask, selfmask = 0u;
    if (!p.odd) {
        if (g == 0) { selfmask = 1u; lmask &= ~(1u << (TC - 1)); }
        if (g == p.G - 1) { rmask = 1u; selfmask = 1u; lmask = 0u; }
    }
    const int nel2 = L * TC2;
    // this thread's column of the tile pair (the workgroup size is a multiple of 2 TC, at most 8 elements per thread) and its twiddles
    int tcol;
    bool tactive;
    {
        const int t2 = (int)threadIdx.x & (TC2 - 1);
        const bool right = t2 < TC;
        const int t = right ? t2 : t2 - TC;
        tcol = right ? rs + t : ls + t;
        if (tcol >= nc) tcol -= nc;
        tactive = ((right ? rmask : lmask) >> t) & 1u;
    }
    PzTwRun<F::NE2> twr;
    twr.init(p, tcol, (int)threadIdx.x >> (lg + 1), (int)blockDim.x >> (lg + 1));
#pragma unroll
    for (int k = 0; k < F::NE2; ++k) {
        const int e = threadIdx.x + k * blockDim.x;
        if (e < nel2) cur[e] = tactive ? cmulc(W[(size_t)(e >> (lg + 1)) * nc + tcol], twr.w[k]) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    F::run(cur, p, TC2, lg + 1, true);

    // ---- pair hook (pz_pair_hook) ----
    const int cha = p.kind == 1 ? st : 2 * st, chb = p.kind == 1 ? st : 2 * st + 1;
    const bool hasb = chb < p.C;
    float mxa = 0.f, mxb = 0.f;
    const int nel = L * TC;
    pz_pair_hook<MAXONLY>(p, h, [&](int i, int slot) { return cur + (size_t)i * TC2 + slot; }, st, rs, rmask, selfmask, mxa, mxb, m2a, m2b);
    if (MAXONLY) {
        mxa = block_max(mxa, red);
        if (p.kind == 2 && hasb) mxb = block_max(mxb, red);
        if (threadIdx.x == 0) {
            fl_max2_commit(h.max2_out, cha, mxa);
            if (p.kind == 2 && hasb) fl_max2_commit(h.max2_out, chb, mxb);
        }
        return;
    }
    // left-tile positions outside [s, s + D): nothing pairs with them
    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        const int t = e & (TC - 1), i = e >> lg;
        if (!((lmask >> t) & 1u)) continue;
        int c = ls + t;
        if (c >= nc) c -= nc;
        const long long k = (long long)i * nc + c - s;
        if (k < 0 || (unsigned long long)k >= D) cur[(size_t)i * TC2 + TC + t] = make_float2(0.f, 0.f);
    }
    pz_max_stage(h, mxa, mxb, red);
    __syncthreads();
    pz_max_commit(p, h, st, g, red);
    F::run(cur, p, TC2, lg + 1, false);
#pragma unroll
    for (int k = 0; k < F::NE2; ++k) {
        const int e = threadIdx.x + k * blockDim.x;
        if (e < nel2 && tactive) W[(size_t)(e >> (lg + 1)) * nc + tcol] = cmul(cur[e], twr.w[k]);
    }
}


// The spectrum pass on a tile pair with four workgroup barriers (the stage-by-stage k_pzpair: 14): the thread-per-(row class,
// column) scheme of k_pzcol_wl on the pair's 4 + 4 columns.  After the inverse transform thread (c, col) holds the natural-order
// rows n = c + LA d of its column; it parks them in ITS OWN LDS row (element i of column slot q at ((i % LA) LBP + i / LA) 8 + q),
// the pair hook runs on that layout (pz_pair_hook), the thread takes its rows back (cropping the left tile), and the forward
// transform proceeds as in k_pzcol_wl.  One 48 KB buffer, no write-after-read hazard between threads at any point.
template <int LA, int LB>
__global__ __launch_bounds__((PzColWl<LA, LB>::THREADS)) void k_pzpair_wl(PzP p, PzHook h, const cplx* __restrict__ tab, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    using G = PzColWl<LA, LB>;
    constexpr int TC = 4, TC2 = 8, LBP = G::LBP, L = LA * LB;
    __shared__ cplx ts[L];
    __shared__ float red[32];
    const int g = (blockIdx.x & 7) * p.g_per_xcd + (blockIdx.x >> 3);
    if (g >= p.G) return;
    const int st = blockIdx.y;
    const int nc = p.nc;
    cplx* lds = (cplx*)EGR_LDS_BASE(smem);
    const int tid = threadIdx.x;
    const int b = tid >> 3, cl = tid & 7;
    for (int e = tid; e < L; e += G::THREADS) ts[e] = tab[e];
    float m2a, m2b;
    pz_max_prefetch(p, h, st, m2a, m2b);
    const unsigned long long D = p.D;
    const long long s = p.s;
    const int rs = (p.c0 + g * TC) % nc;
    int ls = (p.cDr - rs - TC + 1) % nc;
    if (ls < 0) ls += nc;
    unsigned rmask = (1u << TC) - 1u, lmask = rmask, selfmask = 0u;
    if (!p.odd) {
        if (g == 0) { selfmask = 1u; lmask &= ~(1u << (TC - 1)); }
        if (g == p.G - 1) { rmask = 1u; selfmask = 1u; lmask = 0u; }
    }
    const bool right = cl < TC;
    const int t = right ? cl : cl - TC;
    int tcol = right ? rs + t : ls + t;
    if (tcol >= nc) tcol -= nc;
    const bool tactive = ((right ? rmask : lmask) >> t) & 1u;
    cplx* W = work + (size_t)st * p.P + tcol;
    dcplx w0 = make_double2(1.0, 0.0), wst = w0;
    cplx v[LA];
    if (b < LB) {
        if (tactive) {
#pragma unroll
            for (int a = 0; a < LA; ++a) v[a] = W[(size_t)(LB * a + b) * nc];
            w0 = tw2d(p.big, (unsigned)tcol * (unsigned)b);
            wst = tw2d(p.big, (unsigned)tcol * (unsigned)LB);
            dcplx cur = w0;
#pragma unroll
            for (int a = 0; a < LA; ++a) {
                v[a] = pz_tw_mulc(v[a], cur);
                cur = dcmul(cur, wst);
            }
            wl_bfly_inv<LA>(v);
        } else {
#pragma unroll
            for (int a = 0; a < LA; ++a) v[a] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < LA; ++c) lds[(c * LBP + b) * TC2 + cl] = v[c];
    }
    __syncthreads();
    cplx u[LB];
    if (b < LA) {
        const int c = b;
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const cplx z = lds[(c * LBP + j) * TC2 + cl];
            u[j] = (j && c) ? cmulc(z, ts[j * LA + c]) : z;
        }
        wl_bfly_inv<LB>(u);                              // u[d] = row c + LA d of this column, natural order
#pragma unroll
        for (int d = 0; d < LB; ++d) lds[(c * LBP + d) * TC2 + cl] = u[d];
    }
    __syncthreads();
    float mxa = 0.f, mxb = 0.f;
    pz_pair_hook<false>(p, h, [&](int i, int slot) { const int q = i / LA; return lds + ((i - q * LA) * LBP + q) * TC2 + slot; }, st, rs, rmask, selfmask, mxa, mxb, m2a, m2b);
    pz_max_stage(h, mxa, mxb, red);
    __syncthreads();
    pz_max_commit(p, h, st, g, red);
    if (b < LA) {
        const int c = b;
#pragma unroll
        for (int d = 0; d < LB; ++d) {
            u[d] = lds[(c * LBP + d) * TC2 + cl];
            if (!right) {                                // left-tile positions outside [s, s + D): nothing pairs with them
                const long long k = (long long)(c + LA * d) * nc + tcol - s;
                if (k < 0 || (unsigned long long)k >= D) u[d] = make_float2(0.f, 0.f);
            }
        }
        Bfly<LB>::run(u);
#pragma unroll
        for (int j = 0; j < LB; ++j) lds[(c * LBP + j) * TC2 + cl] = (j && c) ? cmul(u[j], ts[j * LA + c]) : u[j];
    }
    __syncthreads();
    if (b < LB && tactive) {
#pragma unroll
        for (int j = 0; j < LA; ++j) v[j] = lds[(j * LBP + b) * TC2 + cl];
        Bfly<LA>::run(v);
        dcplx cur = w0;
#pragma unroll
        for (int a = 0; a < LA; ++a) {
            W[(size_t)(LB * a + b) * nc] = pz_tw_mul(v[a], cur);
            cur = dcmul(cur, wst);
        }
    }
}
typedef void (*PzPairWlFn)(PzP, PzHook, const cplx*, cplx*);
struct PzPairWlEntry { int L; PzPairWlFn fn; };
#define PZ_PAIRWL_ENTRY(LEN, A, B) {LEN, k_pzpair_wl<A, B>},
static const PzPairWlEntry kPzPairWl[] = {PZ_COLWL_LIST(PZ_PAIRWL_ENTRY)};

// One row of length L per workgroup: FFT . x bhat (CONJ: x conj(bhat)) . IFFT, stages in place.  The row's bhat entries are
// requested before the forward transform (registers), so their latency is off the dependent chain.
template <bool CONJ>
__global__ __launch_bounds__(512) void k_pz_rowconv(FftDesc f, int L, const cplx* __restrict__ tw, const cplx* __restrict__ bhat,
                                                     long long P, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* g = work + (size_t)blockIdx.y * P + (size_t)blockIdx.x * L;
    const cplx* bh = bhat + (size_t)blockIdx.x * L;
    if ((L & 1) == 0) {
        float4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 2 * (int)threadIdx.x + j * 1024;
            b[j] = e < L ? *(const float4*)(bh + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e = 2 * threadIdx.x; e < L; e += 1024) *(float4*)(cur + e) = *(const float4*)(g + e);
        __syncthreads();
#ifndef EGR_PZ_ABL_NOFFT
        lds_fft_ip<false, 0, false>(cur, f, tw, 1, 0, 1, L, false);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 2 * (int)threadIdx.x + j * 1024;
            if (e < L) {
                const float4 c = *(const float4*)(cur + e);
                const cplx b0 = make_float2(b[j].x, CONJ ? -b[j].y : b[j].y), b1 = make_float2(b[j].z, CONJ ? -b[j].w : b[j].w);
                const cplx u = cmul(make_float2(c.x, c.y), b0), v = cmul(make_float2(c.z, c.w), b1);
                *(float4*)(cur + e) = make_float4(u.x, u.y, v.x, v.y);
            }
        }
        __syncthreads();
#ifndef EGR_PZ_ABL_NOFFT
        lds_fft_ip<false, 0, false>(cur, f, tw, 1, 0, 1, L, true);
#endif
        for (int e = 2 * threadIdx.x; e < L; e += 1024) *(float4*)(g + e) = *(const float4*)(cur + e);
    } else {
        for (int e = threadIdx.x; e < L; e += 512) cur[e] = g[e];
        __syncthreads();
        lds_fft_ip<false, 0, false>(cur, f, tw, 1, 0, 1, L, false);
        for (int e = threadIdx.x; e < L; e += 512) {
            cplx b = bh[e];
            if (CONJ) b.y = -b.y;
            cur[e] = cmul(cur[e], b);
        }
        __syncthreads();
        lds_fft_ip<false, 0, false>(cur, f, tw, 1, 0, 1, L, true);
        for (int e = threadIdx.x; e < L; e += 512) g[e] = cur[e];
    }
}

// The same for rows of NC = 1024 .. 16384 points with a compile-time schedule starting with radix 16 (in-place stages, rows
// padded by one element per 16: conflict-free first-stage writes); NC / 16 threads, 16 row elements and 16 bhat values per thread.
#ifndef EGR_PZ_ROW_PAD
#define EGR_PZ_ROW_PAD 4
#endif
template <bool CONJ, int NC, int R0, int R1, int R2, int R3>
__global__ __launch_bounds__(NC / 16) void k_pz_rowconv_s(const cplx* __restrict__ stw, const cplx* __restrict__ bhat, long long P,
                                                          cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    constexpr int L = NC, T = NC / 16, PSH = EGR_PZ_ROW_PAD;
    static_assert(R0 * R1 * R2 * R3 == NC, "schedule does not match the row length");
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* g = work + (size_t)blockIdx.y * P + (size_t)blockIdx.x * L;
    const cplx* bh = bhat + (size_t)blockIdx.x * L;
    float4 b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = *(const float4*)(bh + 2 * (int)threadIdx.x + j * 2 * T);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = 2 * (int)threadIdx.x + j * 2 * T;
        const float4 v = *(const float4*)(g + e);
        cplx* d = cur + lds_pad<PSH>(e);            // a pair (e, e + 1), e even, never straddles a pad position
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
    }
    __syncthreads();
#ifndef EGR_PZ_ABL_NOFFT
    lds_fft_sched_inplace<false, PSH, L, T, R0, R1, R2, R3>(cur, L, stw, 1, 0, 1, lds_pad<PSH>(L), false);
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = 2 * (int)threadIdx.x + j * 2 * T;
        cplx* d = cur + lds_pad<PSH>(e);
        const cplx b0 = make_float2(b[j].x, CONJ ? -b[j].y : b[j].y), b1 = make_float2(b[j].z, CONJ ? -b[j].w : b[j].w);
        d[0] = cmul(d[0], b0);
        d[1] = cmul(d[1], b1);
    }
    __syncthreads();
#ifndef EGR_PZ_ABL_NOFFT
    lds_fft_sched_inplace<false, PSH, L, T, R0, R1, R2, R3>(cur, L, stw, 1, 0, 1, lds_pad<PSH>(L), true);
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = 2 * (int)threadIdx.x + j * 2 * T;
        const cplx* d = cur + lds_pad<PSH>(e);
        *(float4*)(g + e) = make_float4(d[0].x, d[0].y, d[1].x, d[1].y);
    }
}


// Rows of NC = 16 R1 R2 points (1024 = 16 8 8, 2048 = 16 16 8, 4096 = 16 16 16) with the outer stages fused into the memory accesses.
// Thread tid of NC / 16 owns the residue class {tid + (NC / 16) m, m < 16}: those are the inputs of ITS first radix-16 butterfly
// (loaded straight from global memory, coalesced), the outputs of ITS last-stage butterflies of the forward transform (so the x Bhat
// product and the first radix-16 stage of the inverse run on registers, no LDS exchange between the two transforms), and the outputs
// of ITS last-stage butterflies of the inverse (stored straight to global memory).  Only the middle stage of each transform runs
// in place in LDS.  8 LDS passes per element instead of 16, 7 barriers instead of 13; same butterflies, tables and arithmetic per
// element as k_pz_rowconv_s (results agree to round-off; the order of the twiddle products is the same).
template <bool CONJ, int NC, int R1, int R2>
__global__ __launch_bounds__(NC / 16) void k_pz_rowconv_f(const cplx* __restrict__ stw, const cplx* __restrict__ bhat, long long P,
                                                          cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    constexpr int L = NC, T = NC / 16, PSH = EGR_PZ_ROW_PAD, NB2 = 16 / R2, NS2 = 16 * R1, RS2 = sched_row_stride(R2);
    static_assert(16 * R1 * R2 == NC && NS2 == NC / R2, "three-stage schedule 16 R1 R2");
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* g = work + (size_t)blockIdx.y * P + (size_t)blockIdx.x * L;
    const cplx* bh = bhat + (size_t)blockIdx.x * L;
    const cplx* stw2 = stw + 16 * sched_row_stride(R1);          // table of the last stage (lds_fft_sched_inplace's layout)
    const int tid = threadIdx.x;
    cplx v[16], bq[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = g[tid + T * m];
#pragma unroll
    for (int m = 0; m < 16; ++m) bq[m] = bh[tid + T * m];
    // ---- forward: stage 0 (radix 16, no twiddles) on registers -> LDS (autosort: butterfly j writes j 16 + t)
    Bfly<16>::run(v);
#pragma unroll
    for (int t = 0; t < 16; ++t) cur[lds_pad<PSH>(tid * 16 + t)] = v[t];
    __syncthreads();
    fft_stage_tab_inplace<R1, 16, false, PSH, sched_nbt(L, R1, T), T>(cur, L, stw, 1, 0, 1, lds_pad<PSH>(L), false, false);
    // ---- forward: last stage (radix R2, Ns = 16 R1 = NC / R2: k = j) LDS -> registers; element index j + t Ns = tid + T (n + NB2 t)
#pragma unroll
    for (int n = 0; n < NB2; ++n) {
        const int j = tid + n * T;
        cplx u[R2];
#pragma unroll
        for (int t = 0; t < R2; ++t) u[t] = cur[lds_pad<PSH>(j + t * NS2)];
        const float4* tp4 = (const float4*)(stw2 + (size_t)j * RS2);
#pragma unroll
        for (int t2 = 0; t2 < RS2 / 2; ++t2) {
            const float4 w = tp4[t2];
            if (2 * t2 >= 1 && 2 * t2 < R2) u[2 * t2] = cmul(u[2 * t2], make_float2(w.x, w.y));
            if (2 * t2 + 1 < R2) u[2 * t2 + 1] = cmul(u[2 * t2 + 1], make_float2(w.z, w.w));
        }
        Bfly<R2>::run(u);
#pragma unroll
        for (int t = 0; t < R2; ++t) v[n + NB2 * t] = u[t];
    }
    // ---- x Bhat, inverse stage 0 (swap . radix 16) on registers
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const cplx b = make_float2(bq[m].x, CONJ ? -bq[m].y : bq[m].y);
        const cplx x = cmul(v[m], b);
        v[m] = make_float2(x.y, x.x);
    }
    Bfly<16>::run(v);
    __syncthreads();                                     // every thread has read its last-stage inputs
#pragma unroll
    for (int t = 0; t < 16; ++t) cur[lds_pad<PSH>(tid * 16 + t)] = v[t];
    __syncthreads();
    fft_stage_tab_inplace<R1, 16, false, PSH, sched_nbt(L, R1, T), T>(cur, L, stw, 1, 0, 1, lds_pad<PSH>(L), false, false);
    // ---- inverse: last stage -> swap -> global
#pragma unroll
    for (int n = 0; n < NB2; ++n) {
        const int j = tid + n * T;
        cplx u[R2];
#pragma unroll
        for (int t = 0; t < R2; ++t) u[t] = cur[lds_pad<PSH>(j + t * NS2)];
        const float4* tp4 = (const float4*)(stw2 + (size_t)j * RS2);
#pragma unroll
        for (int t2 = 0; t2 < RS2 / 2; ++t2) {
            const float4 w = tp4[t2];
            if (2 * t2 >= 1 && 2 * t2 < R2) u[2 * t2] = cmul(u[2 * t2], make_float2(w.x, w.y));
            if (2 * t2 + 1 < R2) u[2 * t2 + 1] = cmul(u[2 * t2 + 1], make_float2(w.z, w.w));
        }
        Bfly<R2>::run(u);
#pragma unroll
        for (int t = 0; t < R2; ++t) g[j + t * NS2] = make_float2(u[t].y, u[t].x);
    }
}

// ---- plan-time double-precision transform of the chirp sequence ----
// b[j] = conj(w[|j|]) = exp(+i pi j^2 / D) for |j| < D (cyclic positions j and P - j), zero elsewhere
__global__ __launch_bounds__(256) void k_pz_chirp_b(unsigned long long D, double inv_2D, long long P, double2* __restrict__ b) {
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < P; j += (long long)gridDim.x * blockDim.x) {
        long long m = -1;
        if ((unsigned long long)j < D) m = j;
        else if ((unsigned long long)(P - j) < D) m = P - j;
        double2 v = make_double2(0.0, 0.0);
        if (m >= 0) {
            const unsigned long long mod = 2ULL * D, m2 = (unsigned long long)m * (unsigned long long)m;
            unsigned long long r;
            if (m2 < (1ULL << 53)) {
                const unsigned long long q = (unsigned long long)((double)m2 * inv_2D);
                long long d = (long long)(m2 - q * mod);
                if (d < 0) d += (long long)mod;
                if (d >= (long long)mod) d -= (long long)mod;
                r = (unsigned long long)d;
            } else {
                r = m2 % mod;
            }
            double sn, cs;
            sincospi((double)r / (double)D, &sn, &cs);          // exp(+i pi r / D)
            v = make_double2(cs, sn);
        }
        b[j] = v;
    }
}

// one Stockham stage of radix R (<= 16) over P points in global memory, direct R-point DFT per butterfly
__global__ __launch_bounds__(256) void k_pz_dfft_stage(const double2* __restrict__ in, double2* __restrict__ out, long long P, int R,
                                                        long long Ns) {
    __shared__ double2 wr[16];
    if ((int)threadIdx.x < R) {
        double sn, cs;
        sincospi(-2.0 * (double)threadIdx.x / (double)R, &sn, &cs);
        wr[threadIdx.x] = make_double2(cs, sn);
    }
    __syncthreads();
    const long long nb = P / R;
    const double inv = 1.0 / ((double)Ns * (double)R);
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < nb; j += (long long)gridDim.x * blockDim.x) {
        const long long k = j % Ns;
        double2 v[16];
        for (int t = 0; t < R; ++t) {
            const double2 x = in[j + (long long)t * nb];
            double sn, cs;
            sincospi(-2.0 * (double)(k * t) * inv, &sn, &cs);
            v[t] = make_double2(x.x * cs - x.y * sn, x.x * sn + x.y * cs);
        }
        const long long o0 = (j - k) * R + k;
        for (int u = 0; u < R; ++u) {
            double ax = 0.0, ay = 0.0;
            int idx = 0;
            for (int t = 0; t < R; ++t) {
                const double2 w = wr[idx];
                ax += v[t].x * w.x - v[t].y * w.y;
                ay += v[t].x * w.y + v[t].y * w.x;
                idx += u; if (idx >= R) idx -= R;
            }
            out[o0 + (long long)u * Ns] = make_double2(ax, ay);
        }
    }
}

// natural-order spectrum -> the passes' storage order, scaled and rounded to float: storage row r = ka Mb + kb, position k holds
// bin (ka + Ma kb) + (Ma Mb) k
__global__ __launch_bounds__(256) void k_pz_bhat_store(const double2* __restrict__ B, long long P, int Lrow, int Ma, int Mb, double scale,
                                                        cplx* __restrict__ bhat) {
    const long long R = (long long)Ma * Mb;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < P; idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / Lrow, k = idx - r * Lrow;
        const long long ka = r / Mb, kb = r - ka * Mb;
        const double2 v = B[ka + (long long)Ma * kb + R * k];
        bhat[idx] = make_float2((float)(v.x * scale), (float)(v.y * scale));
    }
}

typedef void (*PzPairFn)(PzP, PzHook, cplx*);
typedef void (*PzColFn)(PzP, float, cplx*, float*, unsigned*, const unsigned*);
typedef void (*PzRowFn)(const cplx*, const cplx*, long long, cplx*);

struct PzPlan {
    PzP p;
    cplx* d_bhat;
    int nstates;
    // kernels of this plan: the spectrum pass (and its reduction-only form), the crop pass, the opening / closing passes
    PzPairFn pair, pairmax;
    PzColFn crop, first, last;
    int threads_pair, threads_pairmax, threads_crop, threads_col;
    int col_sched;               // column length with a compile-time schedule (0: run-time schedule)
    const cplx* stw_row;         // stage tables of the compile-time row schedule (nullptr: run-time schedule)
    PzRowFn rowconv, rowconv_conj;
    int threads_row;
    const void* crop_wl;         // PzColWlEntry of the two-barrier crop pass (nullptr: k_pzcol<1>)
    const void* pair_wl;         // PzPairWlEntry::fn of the four-barrier spectrum pass (nullptr: k_pzpair)
    const cplx* crop_wl_tab;     // [LB][LA]: W_L^(b c)
    size_t lds_col, lds_pair, lds_row;
};

// Lengths with a compile-time schedule.  Columns X(L, R0, R1, R2): 512 .. 1024 in steps of 3 - 9 %; rows Y(NC, R0, R1, R2, R3):
// 1024 .. 16384 by powers of two -- so every convolution length from 0.5 M to 16.7 M points has a plan P = L x NC within 9 %
// (4 % on average) of the minimum whose two passes both run scheduled kernels, with column tile pairs of 32 - 64 KB whatever
// the length.  Any other plan runs the run-time schedule (the general path, same arithmetic).
#define PZ_SCHED_LIST(X) \
    X(512, 8, 8, 8) X(560, 8, 7, 10) X(600, 6, 10, 10) X(640, 8, 8, 10) X(672, 8, 7, 12) X(720, 9, 8, 10) X(768, 8, 8, 12) \
    X(800, 8, 10, 10) X(840, 7, 10, 12) X(900, 9, 10, 10) X(960, 8, 10, 12) X(1024, 8, 8, 16)
#define PZ_ROW_LIST(Y) \
    Y(1024, 16, 8, 8, 1) Y(2048, 16, 16, 8, 1) Y(4096, 16, 16, 16, 1) Y(8192, 16, 8, 8, 8) Y(16384, 16, 16, 8, 8)

struct PzSchedEntry {
    int L, r0, r1, r2;
    PzPairFn pair;
    PzColFn crop;
    int threads_pair, threads_crop;
};
#define PZ_ENTRY(LEN, A, B, C) {LEN, A, B, C, k_pzpair<false, PzSched<8, A, B, C>>, k_pzcol<1, PzSched<4, A, B, C>>, PzSched<8, A, B, C>::THREADS, PzSched<4, A, B, C>::THREADS},
static const PzSchedEntry kPzSched[] = {PZ_SCHED_LIST(PZ_ENTRY)};
struct PzRowEntry {
    int NC, r0, r1, r2, r3;
    PzRowFn fn, fn_conj;
};
#define PZ_ROW_ENTRY(NC, A, B, C, D) {NC, A, B, C, D, k_pz_rowconv_s<false, NC, A, B, C, D>, k_pz_rowconv_s<true, NC, A, B, C, D>},
static const PzRowEntry kPzRows[] = {PZ_ROW_LIST(PZ_ROW_ENTRY)};

}  // namespace egr

using namespace egr;

bool pz_sched_has(int L, int nc) {
    if (getenv("EGR_PZ_SCHED") && atoi(getenv("EGR_PZ_SCHED")) == 0) return false;
    bool col = false, row = false;
    for (const PzSchedEntry& e : kPzSched) col = col || e.L == L;
    for (const PzRowEntry& e : kPzRows) row = row || e.NC == nc;
    return col && row;
}

void pz_destroy(egr_fatllama_plan* p) {
    if (!p->pz) return;
    if (p->pz->d_bhat) hipFree(p->pz->d_bhat);
    delete p->pz;
    p->pz = nullptr;
}

static int pick_threads(int elements) {          // in-place stages hold up to 8 elements per thread; whole multiples of 256
    int t = ((elements + 7) / 8 + 255) / 256 * 256;
    return t < 256 ? 256 : (t > 1024 ? 1024 : t);
}

int pz_build(egr_fatllama_plan* plan, int kind) {
    PzPlan* z = new PzPlan();
    memset(&z->p, 0, sizeof(PzP));
    z->d_bhat = nullptr;
    z->stw_row = nullptr;
    z->col_sched = 0;
    plan->pz = z;
    const FlSplit& sp = plan->sp;
    const ColP& a = plan->colA;
    const RowP& r = plan->row;
    PzP& q = z->p;
    const long long N = plan->n_out;
    q.f = a.f; q.L = a.L; q.nc = a.ncols; q.TC = a.TC; q.TClog2 = a.TClog2;
    q.ntiles = a.ntiles; q.tiles_per_xcd = a.tiles_per_xcd;
    q.tw = a.tw; q.twd = a.twd; q.big = a.big;
    q.P = sp.M; q.N = N; q.C = plan->C; q.kind = kind;
    q.D = (unsigned long long)(kind == 1 ? N / 2 : N);
    q.inv_2D = 1.0 / (2.0 * (double)q.D);
    q.inv_D = 1.0 / (double)q.D;
    q.odd = (int)(q.D & 1ULL);
    EGR_CHECK(q.nc % (2 * q.TC) == 0, EGR_ERR_UNSUPPORTED, "paired chirp-z needs columns %% (2 TC) == 0 (nc=%d TC=%d)", q.nc, q.TC);
    EGR_CHECK(q.P >= 2 * (long long)q.D - 1, EGR_ERR_UNSUPPORTED, "convolution length %lld too short for D=%llu", q.P, q.D);
    {
        const int m = 2 * q.TC, dm = (int)(q.D % (unsigned long long)m);
        // D odd: Dr = D + 2 s = -1 (mod 2 TC) -> both tiles of a pair are whole aligned tiles; D even: Dr = 0 (mod 2 TC)
        q.s = q.odd ? ((m - 1 - dm) % m + m) % m / 2 : ((m - dm) % m) / 2;
        const unsigned long long Dr = q.D + 2ULL * (unsigned long long)q.s;
        q.iDr = (int)(Dr / (unsigned long long)q.nc);
        q.cDr = (int)(Dr % (unsigned long long)q.nc);
        q.c0 = q.odd ? (q.cDr + 1) / 2 : q.cDr / 2;
        q.G = q.nc / (2 * q.TC) + (q.odd ? 0 : 1);
        q.g_per_xcd = ceil_div(q.G, 8);
    }
    z->nstates = kind == 1 ? plan->C : (plan->C + 1) / 2;
    int rc;
    if ((rc = fl_make_tw2(plan, 2 * (int64_t)q.D, &q.w))) return rc;
    EGR_CHECK(q.L * q.TC <= 8 * 1024 && (r.L <= 8 * 512 || pz_sched_has(q.L, r.L)), EGR_ERR_UNSUPPORTED, "tile too large for the in-place stages (L=%d TC=%d row=%d)", q.L, q.TC, r.L);
    // ---- kernels: run-time schedule by default (in-place stages hold up to 8 elements per thread) ----
    const bool fuse = 2 * q.L * q.TC <= 8 * 1024;
    z->threads_col = z->threads_crop = pick_threads(q.L * q.TC);
    z->threads_pair = z->threads_pairmax = fuse ? pick_threads(2 * q.L * q.TC) : pick_threads(q.L * q.TC);
    z->first = k_pzcol<0, PzRt<true>>; z->crop = k_pzcol<1, PzRt<true>>; z->last = k_pzcol<2, PzRt<true>>;
    z->pair = fuse ? k_pzpair<false, PzRt<true>> : k_pzpair<false, PzRt<false>>;
    z->pairmax = fuse ? k_pzpair<true, PzRt<true>> : k_pzpair<true, PzRt<false>>;
    // compile-time schedules where the plan's lengths have one
    const bool sched_on = !(getenv("EGR_PZ_SCHED") && atoi(getenv("EGR_PZ_SCHED")) == 0);
    if (sched_on && q.TC == 4)
        for (const PzSchedEntry& e : kPzSched)
            if (e.L == q.L) {
                if ((rc = fl_upload_sched_tables(plan, {e.r0, e.r1, e.r2}, &q.stw))) return rc;
                z->pair = e.pair; z->crop = e.crop;
                z->threads_pair = e.threads_pair; z->threads_crop = e.threads_crop;
                z->col_sched = e.L;
            }
    // the crop pass on the two-barrier kernel where the column length has an instantiation (EGR_PZ_COLWL=0: k_pzcol<1>)
    z->crop_wl = nullptr; z->crop_wl_tab = nullptr; z->pair_wl = nullptr;
    if (!(getenv("EGR_PZ_COLWL") && atoi(getenv("EGR_PZ_COLWL")) == 0) && q.nc % 8 == 0 && !(sp.levels == 3))
        for (const PzColWlEntry& e : kPzColWl)
            if (e.L == q.L) {
                std::vector<float2> t((size_t)e.L);
                const long double two_pi = 6.283185307179586476925286766559L;
                for (int b = 0; b < e.lb; ++b)
                    for (int c = 0; c < e.la; ++c) {
                        const long double ang = -two_pi * (long double)((b * c) % e.L) / (long double)e.L;
                        t[(size_t)b * e.la + c] = make_float2((float)cosl(ang), (float)sinl(ang));
                    }
                if ((rc = fl_upload(plan, t, &z->crop_wl_tab))) return rc;
                z->crop_wl = &e;
                // the spectrum pass too, for even/odd packing (kind 1); channel pairs (kind 2, rows of 8192) measure 2 % slower with it
                if (q.TC == 4 && kind == 1 && !(getenv("EGR_PZ_PAIRWL") && atoi(getenv("EGR_PZ_PAIRWL")) == 0))
                    for (const PzPairWlEntry& pe : kPzPairWl) if (pe.L == q.L) z->pair_wl = (const void*)pe.fn;
            }
    z->rowconv = z->rowconv_conj = nullptr;
    z->threads_row = 512;
    if (sched_on)
        for (const PzRowEntry& e : kPzRows)
            if (e.NC == r.L) {
                if (e.r3 > 1) rc = fl_upload_sched_tables(plan, {e.r0, e.r1, e.r2, e.r3}, &z->stw_row);
                else rc = fl_upload_sched_tables(plan, {e.r0, e.r1, e.r2}, &z->stw_row);
                if (rc) return rc;
                z->rowconv = e.fn; z->rowconv_conj = e.fn_conj;
                z->threads_row = e.NC / 16;
            }
    // three-stage rows (1024 / 2048 / 4096 points): the kernel with the outer stages fused into the memory accesses (EGR_PZ_ROWF=0:
    // k_pz_rowconv_s); same tables, same LDS footprint
    if (z->rowconv && !(getenv("EGR_PZ_ROWF") && atoi(getenv("EGR_PZ_ROWF")) == 0)) {
        if (r.L == 4096) { z->rowconv = k_pz_rowconv_f<false, 4096, 16, 16>; z->rowconv_conj = k_pz_rowconv_f<true, 4096, 16, 16>; }
        else if (r.L == 2048) { z->rowconv = k_pz_rowconv_f<false, 2048, 16, 8>; z->rowconv_conj = k_pz_rowconv_f<true, 2048, 16, 8>; }
        else if (r.L == 1024) { z->rowconv = k_pz_rowconv_f<false, 1024, 8, 8>; z->rowconv_conj = k_pz_rowconv_f<true, 1024, 8, 8>; }
    }
    z->lds_col = EGR_LDS((size_t)q.L * q.TC * sizeof(cplx));
    z->lds_pair = EGR_LDS(2 * (size_t)q.L * q.TC * sizeof(cplx));
    z->lds_row = EGR_LDS(z->stw_row ? (size_t)(r.L + (EGR_PZ_ROW_PAD ? r.L >> EGR_PZ_ROW_PAD : 0)) * sizeof(cplx) : (size_t)r.L * sizeof(cplx));
    hipError_t e = hipSuccess;
    EGR_CHECK(z->lds_pair <= (size_t)EGR_LDS_MAX && z->lds_row <= (size_t)EGR_LDS_MAX, EGR_ERR_UNSUPPORTED, "plan needs %zu / %zu bytes of LDS", z->lds_pair, z->lds_row);
    // the attribute is a process-wide cap per kernel: always the CU's maximum (see build_plan)
    auto attr = [&](const void* fn, size_t) { if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, EGR_LDS_MAX); };
    attr((const void*)z->first, z->lds_col); attr((const void*)z->crop, z->lds_col); attr((const void*)z->last, z->lds_col);
    attr((const void*)z->pair, z->lds_pair); attr((const void*)z->pairmax, z->lds_pair);
    if (z->rowconv) { attr((const void*)z->rowconv, z->lds_row); attr((const void*)z->rowconv_conj, z->lds_row); }
    else { attr((const void*)k_pz_rowconv<false>, z->lds_row); attr((const void*)k_pz_rowconv<true>, z->lds_row); }
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) -> %s", hipGetErrorString(e));
        return EGR_ERR_HIP;
    }
    // ---- Bhat = FFT_P(b) / P in double precision, stored in the passes' order ----
    const long long P = q.P;
    double2 *b0 = nullptr, *b1 = nullptr;
    if (hipMalloc((void**)&b0, (size_t)P * sizeof(double2)) != hipSuccess || hipMalloc((void**)&b1, (size_t)P * sizeof(double2)) != hipSuccess ||
        hipMalloc((void**)&z->d_bhat, (size_t)P * sizeof(cplx)) != hipSuccess) {
        if (b0) hipFree(b0);
        if (b1) hipFree(b1);
        set_error("hipMalloc of the %lld-point chirp spectrum failed", P);
        return EGR_ERR_ALLOC;
    }
    const int nb = (int)std::min<long long>((P + 255) / 256, 8192);
    hipLaunchKernelGGL(k_pz_chirp_b, dim3(nb), dim3(256), 0, 0, q.D, q.inv_2D, P, b0);
    {
        long long rem = P, Ns = 1;
        auto stage = [&](int R) {
            hipLaunchKernelGGL(k_pz_dfft_stage, dim3((int)std::min<long long>((P / R + 255) / 256, 8192)), dim3(256), 0, 0, (const double2*)b0, b1, P, R, Ns);
            std::swap(b0, b1);
            Ns *= R; rem /= R;
        };
        while (rem % 16 == 0) stage(16);
        while (rem % 8 == 0) stage(8);
        while (rem % 4 == 0) stage(4);
        while (rem % 2 == 0) stage(2);
        while (rem % 9 == 0) stage(9);
        for (int pr : {3, 5, 7, 11, 13}) while (rem % pr == 0) stage(pr);
        if (rem != 1) {
            hipFree(b0); hipFree(b1);
            set_error("convolution length %lld has a prime factor above 13", P);
            return EGR_ERR_UNSUPPORTED;
        }
    }
    hipLaunchKernelGGL(k_pz_bhat_store, dim3(nb), dim3(256), 0, 0, (const double2*)b0, P, r.L, r.Ma, r.Mb, 1.0 / (double)P, z->d_bhat);
    const hipError_t se = hipDeviceSynchronize();
    hipFree(b0); hipFree(b1);
    if (se != hipSuccess || hipGetLastError() != hipSuccess) {
        set_error("building the chirp spectrum failed: %s", hipGetErrorString(se));
        return EGR_ERR_HIP;
    }
    return EGR_OK;
}

namespace {
struct PzRun {
    egr_fatllama_plan* plan;
    PzPlan* z;
    bool three;
    void conv(bool conj, cplx* work, int ns, hipStream_t st, bool prof, size_t* slot) const {
        const RowP& r = plan->row;
        if (three) fl_launch_inner(plan, true, work, ns, st);
        if (prof) fl_prof_begin(plan, 0, st, slot);
        if (z->rowconv) {
            hipLaunchKernelGGL(conj ? z->rowconv_conj : z->rowconv, dim3(r.R, ns), dim3(z->threads_row), z->lds_row, st, z->stw_row, (const cplx*)z->d_bhat, z->p.P, work);
        } else {
            if (conj) hipLaunchKernelGGL(k_pz_rowconv<true>, dim3(r.R, ns), dim3(512), z->lds_row, st, r.f, r.L, r.tw, (const cplx*)z->d_bhat, z->p.P, work);
            else hipLaunchKernelGGL(k_pz_rowconv<false>, dim3(r.R, ns), dim3(512), z->lds_row, st, r.f, r.L, r.tw, (const cplx*)z->d_bhat, z->p.P, work);
        }
        if (prof) fl_prof_end(plan, st, slot);
        if (three) fl_launch_inner(plan, false, work, ns, st);
    }
    void pair(const PzP& q, const PzHook& h, bool maxonly, cplx* work, int ns, hipStream_t st, bool prof, size_t* slot) const {
        const dim3 g(8 * q.g_per_xcd, ns);
        if (prof && !maxonly) fl_prof_begin(plan, 1, st, slot);
        if (maxonly) hipLaunchKernelGGL(z->pairmax, g, dim3(z->threads_pairmax), z->lds_pair, st, q, h, work);
        else if (z->pair_wl) {
            const PzColWlEntry* e = (const PzColWlEntry*)z->crop_wl;          // same geometry: 8 column slots, max(LA, LB) threads each
            hipLaunchKernelGGL((PzPairWlFn)z->pair_wl, g, dim3(e->threads), EGR_LDS(e->lds), st, q, h, z->crop_wl_tab, work);
        } else hipLaunchKernelGGL(z->pair, g, dim3(z->threads_pair), z->lds_pair, st, q, h, work);
        if (prof && !maxonly) fl_prof_end(plan, st, slot);
    }
};
}  // namespace

// The loop between k_prepare and k_finalize: `out` holds y, receives y + d; per-channel peaks -> peak_out.
int pz_loop(egr_fatllama_plan* plan, float* out, int max_iter, float thr, float thr0, const unsigned* thr0_rel, unsigned flags,
            unsigned* peak_out, hipStream_t st) {
    PzPlan* z = plan->pz;
    EGR_CHECK(z != nullptr && max_iter >= 1, EGR_ERR_ARG, "paired chirp-z plan / iteration count");
    const PzP& q = z->p;
    const int C = plan->C;
    const bool relative = (flags & EGR_FL_THR_RELATIVE) != 0;
    const bool recompute = relative && (flags & EGR_FL_THR_RECOMPUTE) != 0;
    auto max2_slot = [&](int it, int ch0) { return plan->d_max2 + ((size_t)(it % EGR_FL_MAX_RING) * C + ch0) * EGR_FL_MAX_STRIDE; };
    PzHook h;
    memset(&h, 0, sizeof(h));
    h.thr = thr;
    h.soft = (flags & EGR_FL_THR_SOFT) ? 1 : 0;
    PzRun run{plan, z, plan->sp.levels == 3};
    // states are independent until k_finalize: two groups of states run as concurrent pipelines on two streams
    const int ns_all = z->nstates;
    const int ngroups = (plan->nstreams == 2 && ns_all >= 2) ? 2 : 1;
    if (ngroups == 2 && !plan->side) {
        EGR_HIP(hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking));
        plan->side_owned = 1;
    }
    if (ngroups == 2 && !plan->ev_fork) {
        EGR_HIP(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
        EGR_HIP(hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming));
    }
    size_t slot = 0;
    const bool prof = plan->profiling;
    auto fork = [&](hipStream_t s0) -> int {
        if (ngroups == 2) {
            EGR_HIP(hipEventRecord(plan->ev_fork, s0));
            EGR_HIP(hipStreamWaitEvent(plan->side, plan->ev_fork, 0));
        }
        return EGR_OK;
    };
    auto join = [&](hipStream_t s0) -> int {
        if (ngroups == 2) {
            EGR_HIP(hipEventRecord(plan->ev_join, plan->side));
            EGR_HIP(hipStreamWaitEvent(s0, plan->ev_join, 0));
        }
        return EGR_OK;
    };
    // One group's launches: `first` adds the opening pass, iterations [it0, it1), `last` the closing pass.
    auto run_group = [&](hipStream_t s0, int g, int it0, int it1, bool first, bool last, bool profiling) {
        const int sb = g == 0 ? 0 : ns_all / 2, ns = ngroups == 1 ? ns_all : (g == 0 ? ns_all / 2 : ns_all - ns_all / 2);
        hipStream_t sg = g == 0 ? s0 : plan->side;
        cplx* work = plan->d_work + (size_t)sb * q.P;
        // the kernels index channels from the state id: shift the per-channel arrays so that state 0 of the group is local state 0
        const int ch0 = q.kind == 1 ? sb : 2 * sb;
        PzP qg = q;
        qg.C = C - ch0;
        float* og = out + (size_t)ch0 * q.N;
        unsigned* pk = peak_out + ch0;
        const dim3 gc(8 * q.tiles_per_xcd, ns), bc(z->threads_col);
        if (first) hipLaunchKernelGGL(z->first, gc, bc, z->lds_col, sg, qg, thr0, work, og, pk, thr0_rel ? thr0_rel + ch0 : nullptr);
        for (int it = it0; it < it1; ++it) {
            const bool p_it = profiling && g == 0 && it >= max_iter - 3;      // events around the last iterations of group 0
            run.conv(false, work, ns, sg, p_it, &slot);
            PzHook hg = h;
            if (relative) {
                // the spectrum maximum from a read-only pass only where no hook has left it: iteration 0 (or every one, on request)
                if (it == 0 || recompute) {
                    hg.max2_out = max2_slot(it, ch0);
                    run.pair(qg, hg, true, work, ns, sg, false, &slot);
                }
                hg.max2 = max2_slot(it, ch0);
                hg.max2_next = recompute ? nullptr : max2_slot(it + 1, ch0);
                hg.max2_zero = max2_slot(it + 2, ch0);
            }
            run.pair(qg, hg, false, work, ns, sg, p_it, &slot);
            run.conv(true, work, ns, sg, false, &slot);
            if (it + 1 < max_iter) {
                if (p_it) fl_prof_begin(plan, 2, sg, &slot);
                if (z->crop_wl) {
                    const PzColWlEntry* e = (const PzColWlEntry*)z->crop_wl;
                    const int nt8 = q.nc / 8, tpx8 = ceil_div(nt8, 8);
                    hipLaunchKernelGGL(e->fn, dim3(8 * tpx8, ns), dim3(e->threads), EGR_LDS(e->lds), sg, qg, z->crop_wl_tab, nt8, tpx8, work);
                } else
                hipLaunchKernelGGL(z->crop, gc, dim3(z->threads_crop), z->lds_col, sg, qg, thr, work, og, pk, (const unsigned*)nullptr);
                if (p_it) fl_prof_end(plan, sg, &slot);
            }
        }
        if (last) hipLaunchKernelGGL(z->last, gc, bc, z->lds_col, sg, qg, thr, work, og, pk, (const unsigned*)nullptr);
    };
    // The four launches of a middle iteration are the same every iteration and touch the plan's own state only: CH iterations of all
    // pipelines are captured once into a hipGraph and replayed (as the packed loop does, egr_fatllama.hip); keyed by (threshold,
    // pipelines, hook kind), never destroyed while a launch of it may be in flight.  Profiling and EGR_FL_THR_RECOMPUTE use plain
    // launches; the relative threshold's carried maxima live in a ring of CH slots addressed by iteration mod CH.
    constexpr int CH = 25;
    // (two pipelines: 113.9 -> 103.4 ms per 800 iterations of 60 s + 2 samples; a single pipeline measures the same either way)
    static_assert(CH == EGR_FL_MAX_RING, "the captured iterations address the ring of maxima by iteration mod CH");
    // relative threshold: iteration 0 (the one with a maximum pass of its own, behind its first convolution) stays outside the graph
    const int pre = relative ? 1 : 0;
    const int n_graph = (!prof && plan->use_graph && !recompute && ngroups == 2 && max_iter > 2 * CH) ? (max_iter - 1 - pre) / CH : 0;
    const int g_kind = h.soft | (relative ? 2 : 0);
    int rc = fork(st);
    if (rc) return rc;
    for (int g = 0; g < ngroups; ++g) run_group(st, g, 0, n_graph > 0 ? pre : 0, true, false, false);
    if (n_graph > 0) {
        rc = join(st);
        if (rc) return rc;
        if (!(plan->gexec && plan->g_thr == thr && plan->g_groups == ngroups && plan->g_iter_odd == g_kind)) {
            if (plan->gexec) { EGR_HIP(hipDeviceSynchronize()); EGR_HIP(hipGraphExecDestroy(plan->gexec)); plan->gexec = nullptr; }
            hipGraph_t graph = nullptr;
            if (!plan->cap) EGR_HIP(hipStreamCreateWithFlags(&plan->cap, hipStreamNonBlocking));
            EGR_HIP(hipStreamBeginCapture(plan->cap, hipStreamCaptureModeThreadLocal));
            rc = fork(plan->cap);
            if (!rc) for (int g = 0; g < ngroups; ++g) run_group(plan->cap, g, pre, pre + CH, false, false, false);
            if (!rc) rc = join(plan->cap);
            hipError_t ce = hipStreamEndCapture(plan->cap, &graph);
            if (rc || ce != hipSuccess) {
                if (graph) hipGraphDestroy(graph);
                hipStreamDestroy(plan->cap);
                plan->cap = nullptr;
                if (rc) return rc;
                EGR_HIP(ce);
            }
            hipError_t ie = hipGraphInstantiate(&plan->gexec, graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            if (ie != hipSuccess) { plan->gexec = nullptr; EGR_HIP(ie); }
            plan->g_out = out; plan->g_thr = thr; plan->g_groups = ngroups; plan->g_iter_odd = g_kind;
        }
        for (int i = 0; i < n_graph; ++i) EGR_HIP(hipGraphLaunch(plan->gexec, st));
        rc = fork(st);
        if (rc) return rc;
    }
    for (int g = 0; g < ngroups; ++g) run_group(st, g, n_graph > 0 ? pre + n_graph * CH : 0, max_iter, false, true, prof);
    return join(st);
}

// y = irfft(rfft(x) * [k >= band_lo]) per channel on a paired chirp-z plan (factor 1): first pass without a threshold, one
// convolution, the band hook, the conjugate convolution, closing pass onto a zeroed y.
int pz_band_filter(egr_fatllama_plan* plan, const float* x, int64_t band_lo, float* y, hipStream_t st) {
    PzPlan* z = plan->pz;
    EGR_CHECK(z != nullptr, EGR_ERR_ARG, "paired chirp-z plan wanted");
    const PzP& q = z->p;
    PzHook h;
    memset(&h, 0, sizeof(h));
    h.band = 1;
    h.band_lo = (unsigned long long)band_lo;
    PzRun run{plan, z, plan->sp.levels == 3};
    const int ns = z->nstates;
    unsigned* pk = plan->d_peaks + plan->C;
    size_t slot = 0;
    const dim3 gc(8 * q.tiles_per_xcd, ns), bc(z->threads_col);
    hipLaunchKernelGGL(z->first, gc, bc, z->lds_col, st, q, -1.0f, plan->d_work, const_cast<float*>(x), pk, (const unsigned*)nullptr);
    run.conv(false, plan->d_work, ns, st, false, &slot);
    run.pair(q, h, false, plan->d_work, ns, st, false, &slot);
    run.conv(true, plan->d_work, ns, st, false, &slot);
    EGR_HIP(hipMemsetAsync(y, 0, (size_t)plan->C * q.N * sizeof(float), st));      // the closing pass adds d to what y holds
    hipLaunchKernelGGL(z->last, gc, bc, z->lds_col, st, q, 0.f, plan->d_work, y, pk, (const unsigned*)nullptr);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

long long pz_canary_failures() {
#ifdef EGR_LDS_CANARY
    unsigned v = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&v, HIP_SYMBOL(egr::g_lds_canary_fail), sizeof(v)) != hipSuccess) return -2;
    return (long long)v;
#else
    return -1;
#endif
}
