// In-LDS mixed-radix Stockham FFT building blocks for gfx950 (wave64).
//
// A "sequence" is one length-L complex series living in LDS at addresses i*es + s*ss (element i of
// sequence s).  Each stage reads r inputs at stride L/r, applies the stage twiddle W_{Ns*r}^{k*t}
// (looked up in the global W_L table, L1/L2 resident), does an r-point DFT in registers and writes
// the autosorted outputs to the other LDS buffer (ping-pong), so input and output are both in natural
// order and no bit/digit reversal pass exists.  Inverse transforms use swap(FFT(swap(x))).
#pragma once
#include <hip/hip_runtime.h>

namespace egr {

typedef float2 cplx;

#define EGR_MAX_STAGES 14

struct FftDesc {
    int L;                       // transform length
    int nst;                     // number of radix stages
    int radix[EGR_MAX_STAGES];   // radix of stage s
    int ns[EGR_MAX_STAGES];      // product of the radices of stages < s
    float inv_ns[EGR_MAX_STAGES];
    int tw_pow;                  // 1: load W^1 (double precision) per butterfly and form W^2..W^(r-1) in fp64
};

typedef double2 dcplx;
__device__ __forceinline__ dcplx dcmul(dcplx a, dcplx b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// Products with a constant held as TWO floats (w = wh + wl to 2^-48): the low parts enter first, the result is rounded where a plain
// product is, but the constant itself carries no rounding -- |wh + wl|^2 = 1 to 1e-15, where |float(w)|^2 - 1 ~ 6e-8 is a gain the
// loop would apply to the same coefficient in the same direction in every iteration (W on the way in, conj W on the way out).
__device__ __forceinline__ cplx cmul2(cplx a, cplx bh, cplx bl) {
    return make_float2(fmaf(a.x, bh.x, fmaf(-a.y, bh.y, fmaf(a.x, bl.x, -a.y * bl.y))), fmaf(a.x, bh.y, fmaf(a.y, bh.x, fmaf(a.x, bl.y, a.y * bl.x))));
}
__device__ __forceinline__ cplx cmulc2(cplx a, cplx bh, cplx bl) {  // a * conj(b)
    return make_float2(fmaf(a.x, bh.x, fmaf(a.y, bh.y, fmaf(a.x, bl.x, a.y * bl.y))), fmaf(a.y, bh.x, fmaf(-a.x, bh.y, fmaf(a.y, bl.x, -a.x * bl.y))));
}
// The register butterflies' own constants (cos / sin of the odd radices, inner twiddles of the composites) as two floats -- bit mask by
// radix family: 1 radix 5 and 25 = 5 x 5; 2 radix 8 / 16 / 32 (W_16, W_32 inner twiddles); 4 radix 3 and 6 / 9 / 12; 8 every other radix
#ifndef EGR_BFLY_HILO
#define EGR_BFLY_HILO 0
#endif
constexpr bool bfly_hilo(int R) {
    return (R == 5 || R == 25) ? (EGR_BFLY_HILO & 1) != 0
         : (R == 8 || R == 16 || R == 32) ? (EGR_BFLY_HILO & 2) != 0
         : (R == 3 || R == 6 || R == 9 || R == 12) ? (EGR_BFLY_HILO & 4) != 0
         : (EGR_BFLY_HILO & 8) != 0;
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_float2(a.x - b.x, a.y - b.y); }

template <int R> struct Trig;
#include "egr_trig_tables.inc"

template <int R> struct Bfly {
    // forward DFT of odd prime length R using the (x_n + x_{R-n}), (x_n - x_{R-n}) symmetry
    static __device__ __forceinline__ void run(cplx (&v)[R]) {
        constexpr int H = (R - 1) / 2;
        cplx a[H], b[H];
#pragma unroll
        for (int n = 1; n <= H; ++n) {
            a[n - 1] = cadd(v[n], v[R - n]);
            b[n - 1] = csub(v[n], v[R - n]);
        }
        cplx x0 = v[0];
        cplx sum = x0;
#pragma unroll
        for (int n = 0; n < H; ++n) sum = cadd(sum, a[n]);
        v[0] = sum;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            cplx c = x0, s = make_float2(0.f, 0.f);
#pragma unroll
            for (int n = 1; n <= H; ++n) {
                const float cc = Trig<R>::c[(n * k) % R];
                const float ss = Trig<R>::s[(n * k) % R];
                if (bfly_hilo(R)) {
                    const float cl = Trig<R>::cl[(n * k) % R], sl = Trig<R>::sl[(n * k) % R];
                    c.x = fmaf(cc, a[n - 1].x, fmaf(cl, a[n - 1].x, c.x)); c.y = fmaf(cc, a[n - 1].y, fmaf(cl, a[n - 1].y, c.y));
                    s.x = fmaf(ss, b[n - 1].x, fmaf(sl, b[n - 1].x, s.x)); s.y = fmaf(ss, b[n - 1].y, fmaf(sl, b[n - 1].y, s.y));
                    continue;
                }
                c.x += cc * a[n - 1].x; c.y += cc * a[n - 1].y;
                s.x += ss * b[n - 1].x; s.y += ss * b[n - 1].y;
            }
            // X_k = c - i*s ; X_{R-k} = c + i*s
            v[k] = make_float2(c.x + s.y, c.y - s.x);
            v[R - k] = make_float2(c.x - s.y, c.y + s.x);
        }
    }
};

template <> struct Bfly<2> {
    static __device__ __forceinline__ void run(cplx (&v)[2]) {
        cplx a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <> struct Bfly<4> {
    static __device__ __forceinline__ void run(cplx (&v)[4]) {
        cplx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        cplx t2 = cadd(v[1], v[3]), t3 = csub(v[1], v[3]);
        v[0] = cadd(t0, t2);
        v[2] = csub(t0, t2);
        v[1] = make_float2(t1.x + t3.y, t1.y - t3.x);   // t1 - i*t3
        v[3] = make_float2(t1.x - t3.y, t1.y + t3.x);   // t1 + i*t3
    }
};

// Composite radix R = R1*R2 as two nested register DFTs (n = R2*n1 + n2, k = k1 + R1*k2) with compile-time
// inner twiddles W_R^(n2*k1): halves the number of LDS round trips of a transform.
template <int R1, int R2> struct BflyComp {
    static constexpr int R = R1 * R2;
    static __device__ __forceinline__ void run(cplx (&v)[R]) {
        cplx y[R];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) {
            cplx t[R1];
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[R2 * n1 + n2];
            Bfly<R1>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) {
                const int m = (n2 * k1) % R;
                if (bfly_hilo(R) && m != 0 && (4 * m) % R != 0) {          // (quarter turns are exact)
                    y[k1 * R2 + n2] = cmul2(t[k1], make_float2(Trig<R>::c[m], -Trig<R>::s[m]), make_float2(Trig<R>::cl[m], -Trig<R>::sl[m]));
                    continue;
                }
                y[k1 * R2 + n2] = (m == 0) ? t[k1] : cmul(t[k1], make_float2(Trig<R>::c[m], -Trig<R>::s[m]));
            }
        }
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            cplx t[R2];
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) t[n2] = y[k1 * R2 + n2];
            Bfly<R2>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
        }
    }
};
template <> struct Bfly<8> { static __device__ __forceinline__ void run(cplx (&v)[8]) { BflyComp<4, 2>::run(v); } };
template <> struct Bfly<6> { static __device__ __forceinline__ void run(cplx (&v)[6]) { BflyComp<3, 2>::run(v); } };
template <> struct Bfly<12> { static __device__ __forceinline__ void run(cplx (&v)[12]) { BflyComp<4, 3>::run(v); } };
template <> struct Bfly<10> { static __device__ __forceinline__ void run(cplx (&v)[10]) { BflyComp<5, 2>::run(v); } };
template <> struct Bfly<9> { static __device__ __forceinline__ void run(cplx (&v)[9]) { BflyComp<3, 3>::run(v); } };
template <> struct Bfly<16> { static __device__ __forceinline__ void run(cplx (&v)[16]) { BflyComp<4, 4>::run(v); } };
template <> struct Bfly<25> { static __device__ __forceinline__ void run(cplx (&v)[25]) { BflyComp<5, 5>::run(v); } };
// larger composites: register butterflies of the thread-per-(row class, column) column kernels (k_pzcol_wl)
template <> struct Bfly<20> { static __device__ __forceinline__ void run(cplx (&v)[20]) { BflyComp<4, 5>::run(v); } };
template <> struct Bfly<24> { static __device__ __forceinline__ void run(cplx (&v)[24]) { BflyComp<4, 6>::run(v); } };
template <> struct Bfly<28> { static __device__ __forceinline__ void run(cplx (&v)[28]) { BflyComp<4, 7>::run(v); } };
template <> struct Bfly<30> { static __device__ __forceinline__ void run(cplx (&v)[30]) { BflyComp<5, 6>::run(v); } };
template <> struct Bfly<32> { static __device__ __forceinline__ void run(cplx (&v)[32]) { BflyComp<4, 8>::run(v); } };
// 44.1 kHz-family plans (csrc/egr_fatllama_wl.h): columns of 441 = 21 x 21 points, cross radices 14 and 18
template <> struct Bfly<15> { static __device__ __forceinline__ void run(cplx (&v)[15]) { BflyComp<3, 5>::run(v); } };
template <> struct Bfly<14> { static __device__ __forceinline__ void run(cplx (&v)[14]) { BflyComp<7, 2>::run(v); } };
template <> struct Bfly<18> { static __device__ __forceinline__ void run(cplx (&v)[18]) { BflyComp<9, 2>::run(v); } };
template <> struct Bfly<21> { static __device__ __forceinline__ void run(cplx (&v)[21]) { BflyComp<3, 7>::run(v); } };

// One radix-R stage over `nseq` sequences.  SEQFAST: sequence index is the fastest thread index
// (column tiles, nseq = 1<<seq_log2); otherwise nseq is 1 or 2 and the butterfly index is fastest.
template <int R, bool SEQFAST>
__device__ __forceinline__ void fft_stage(const cplx* __restrict__ in, cplx* __restrict__ out, int L, int Ns,
                                          float inv_ns, const cplx* __restrict__ tw, const dcplx* __restrict__ twd,
                                          int nseq, int seq_log2, int es, int ss, bool swap_in, bool swap_out,
                                          bool tw_pow) {
    const int nb = L / R;
    const int total = nb * nseq;
    const int twstep = L / (Ns * R);
    for (int b = threadIdx.x; b < total; b += blockDim.x) {
        int s, j;
        if (SEQFAST) {
            s = b & (nseq - 1);
            j = b >> seq_log2;
        } else {
            s = (b >= nb) ? 1 : 0;
            j = b - s * nb;
        }
        // k = j mod Ns, exact for j,Ns < 2^13 (see DESIGN.md "integer division by float reciprocal")
        const int q = (int)(((float)j + 0.5f) * inv_ns);
        const int k = j - q * Ns;
        cplx v[R];
        const cplx* src = in + s * ss;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            cplx x = src[(j + t * nb) * es];
            v[t] = swap_in ? make_float2(x.y, x.x) : x;
        }
#ifndef EGR_FL_ABL_NOBFLY
        if (Ns > 1) {
            const int base = k * twstep;
            if (tw_pow) {
                // ONE 16-byte load of the fp64 table entry; W^t by a depth-log2(t) product tree in fp64, rounded once
                // to float: as accurate as reading every W^t from a float table, a third / a quarter of the loads.
                dcplx w[R];
#ifdef EGR_FL_ABL_NOTW
                w[1] = make_double2(1.0 - 1e-9 * base, 1e-9 * base);   // dev ablation: no table load (wrong results)
#else
                w[1] = twd[base];
#endif
#pragma unroll
                for (int t = 2; t < R; ++t) w[t] = dcmul(w[t >> 1], w[t - (t >> 1)]);
#pragma unroll
                for (int t = 1; t < R; ++t) v[t] = cmul(v[t], make_float2((float)w[t].x, (float)w[t].y));
            } else {
#pragma unroll
                for (int t = 1; t < R; ++t) v[t] = cmul(v[t], tw[base * t]);
            }
        }
#endif
#ifndef EGR_FL_ABL_NOBFLY
        Bfly<R>::run(v);
#endif
        cplx* dst = out + s * ss + ((j - k) * R + k) * es;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            cplx x = v[t];
            dst[t * Ns * es] = swap_out ? make_float2(x.y, x.x) : x;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// Compile-time schedules (the hot Fat-Llama plans): a kernel instantiated for ONE radix schedule keeps only that schedule's
// registers (the all-radix kernel needed 190 VGPRs with the composite butterflies in) and its stages run in place.  Measured on
// the C3 plan (gpurun_out/sw48 - sw50): rows 2304 = 16 12 12 beats 16 16 9, 8 8 6 6, 4 4 4 4 9 and every order starting with 12 or 9
// (the pad of lds_pad is matched to a first radix of 16); columns 625 = 25 25 beats 5 5 5 5 and 25 5 5.  Stage twiddles come
// from a per-stage table laid out for the butterfly: row k holds W^(k t), t = 0..R-1 (padded to an even count so rows are
// 16-byte aligned), rounded once from long double -- no fp64 power chains in the loop.
constexpr int sched_row_stride(int R) { return (R + 1) & ~1; }

// A stage runs IN PLACE: every
// butterfly is read into registers, a barrier retires the reads, then the results overwrite the same buffer.  Two barriers per
// stage instead of one, half the LDS (no ping-pong buffer) -- which is what lets a third workgroup share the CU.
// PSH > 0: element i of a sequence sits at i + (i >> PSH) (one pad element per 2^PSH): the first Stockham stage of a contiguous
// sequence writes butterfly j's outputs at j R + t -- a lane stride of 8 R bytes, every lane of a half-wave on the same two banks --
// and the pad turns that into a stride of 8 R + 8 (conflict-free); the strided reads stay conflict-free.
template <int PSH> __device__ __forceinline__ int lds_pad(int i) { return PSH ? i + (i >> PSH) : i; }

// NBT butterflies per thread (b = tid + n * THREADS, THREADS = the workgroup size), all held in registers across the first barrier.
template <int R, int NS, bool SEQFAST, int PSH, int NBT, int THREADS>
__device__ __forceinline__ void fft_stage_tab_inplace(cplx* __restrict__ buf, int L, const cplx* __restrict__ stw, int nseq, int seq_log2,
                                                      int es, int ss, bool swap_in, bool swap_out) {
    constexpr int RS = sched_row_stride(R);
    const int nb = L / R;
    const int total = nb * nseq;
    cplx v[NBT][R];
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
        const int b = threadIdx.x + n * THREADS;
        if (b < total) {
            int s, j;
            if (SEQFAST) { s = b & (nseq - 1); j = b >> seq_log2; }
            else { s = (b >= nb) ? 1 : 0; j = b - s * nb; }
            const int k = (NS == 1) ? 0 : j % NS;
            const cplx* src = buf + s * ss;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const cplx x = src[lds_pad<PSH>(j + t * nb) * es];
                v[n][t] = swap_in ? make_float2(x.y, x.x) : x;
            }
            if (NS > 1) {
                const float2* tp = (const float2*)(stw + (size_t)k * RS);
                if (RS % 2 == 0) {
                    const float4* tp4 = (const float4*)tp;
#pragma unroll
                    for (int t2 = 0; t2 < RS / 2; ++t2) {
                        const float4 w = tp4[t2];
                        if (2 * t2 >= 1 && 2 * t2 < R) v[n][2 * t2] = cmul(v[n][2 * t2], make_float2(w.x, w.y));
                        if (2 * t2 + 1 < R) v[n][2 * t2 + 1] = cmul(v[n][2 * t2 + 1], make_float2(w.z, w.w));
                    }
                }
            }
            Bfly<R>::run(v[n]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
        const int b = threadIdx.x + n * THREADS;
        if (b < total) {
            int s, j;
            if (SEQFAST) { s = b & (nseq - 1); j = b >> seq_log2; }
            else { s = (b >= nb) ? 1 : 0; j = b - s * nb; }
            const int k = (NS == 1) ? 0 : j % NS;
            cplx* dst = buf + s * ss;
            const int o0 = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const cplx x = v[n][t];
                dst[lds_pad<PSH>(o0 + t * NS) * es] = swap_out ? make_float2(x.y, x.x) : x;
            }
        }
    }
    __syncthreads();
}

// In-place transform with the compile-time schedule <R0 .. R4> (1 = unused).  LTOT: elements of all sequences of the workgroup (fixes
// the butterflies per thread of every stage).  Stage s > 0 reads its table behind those of the stages before it.
constexpr int sched_nbt(int ltot, int r, int threads) { return (ltot / r + threads - 1) / threads; }
template <bool SEQFAST, int PSH, int LTOT, int THREADS, int R0, int R1, int R2 = 1, int R3 = 1, int R4 = 1>
__device__ __forceinline__ void lds_fft_sched_inplace(cplx* buf, int L, const cplx* __restrict__ stw, int nseq, int seq_log2, int es, int ss,
                                                      bool inverse) {
    constexpr int NST = 2 + (R2 > 1) + (R3 > 1) + (R4 > 1);
    constexpr int o2 = R0 * sched_row_stride(R1), o3 = o2 + R0 * R1 * sched_row_stride(R2), o4 = o3 + R0 * R1 * R2 * sched_row_stride(R3);
    fft_stage_tab_inplace<R0, 1, SEQFAST, PSH, sched_nbt(LTOT, R0, THREADS), THREADS>(buf, L, stw, nseq, seq_log2, es, ss, inverse, false);
    fft_stage_tab_inplace<R1, R0, SEQFAST, PSH, sched_nbt(LTOT, R1, THREADS), THREADS>(buf, L, stw, nseq, seq_log2, es, ss, false, inverse && NST == 2);
    if constexpr (R2 > 1)
        fft_stage_tab_inplace<R2, R0 * R1, SEQFAST, PSH, sched_nbt(LTOT, R2, THREADS), THREADS>(buf, L, stw + o2, nseq, seq_log2, es, ss, false, inverse && NST == 3);
    if constexpr (R3 > 1)
        fft_stage_tab_inplace<R3, R0 * R1 * R2, SEQFAST, PSH, sched_nbt(LTOT, R3, THREADS), THREADS>(buf, L, stw + o3, nseq, seq_log2, es, ss, false, inverse && NST == 4);
    if constexpr (R4 > 1)
        fft_stage_tab_inplace<R4, R0 * R1 * R2 * R3, SEQFAST, PSH, sched_nbt(LTOT, R4, THREADS), THREADS>(buf, L, stw + o4, nseq, seq_log2, es, ss, false, inverse);
}

// One stage of ANY radix R (used for prime factors > 13): every thread produces ONE output element as a direct R-term sum read
// from LDS -- R times the LDS reads of a register butterfly, which is irrelevant for the single such stage an STFT frame has.
// Both twiddles come from the W_L table: stage twiddle W_L^(k twstep t) and DFT kernel W_R^(t u) = W_L^((t u mod R) L / R).
template <bool SEQFAST>
__device__ __forceinline__ void fft_stage_generic(const cplx* __restrict__ in, cplx* __restrict__ out, int L, int Ns, int R,
                                                  const cplx* __restrict__ tw, int nseq, int seq_log2, int es, int ss,
                                                  bool swap_in, bool swap_out) {
    const int nb = L / R;
    const int twstep = L / (Ns * R), rstep = L / R;
    const int total = L * nseq;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int s, o;
        if (SEQFAST) { s = e & (nseq - 1); o = e >> seq_log2; }
        else { s = (e >= L) ? 1 : 0; o = e - s * L; }
        const int u = o / nb, j = o - u * nb;          // output u of butterfly j
        const int k = j % Ns;
        const cplx* src = in + s * ss;
        float ax = 0.f, ay = 0.f;
        int i1 = 0, i2 = 0;                            // (k twstep t) mod L and (t u mod R) rstep, advanced incrementally
        const int d1 = (k * twstep) % L, d2 = u * rstep;
        for (int t = 0; t < R; ++t) {
            cplx x = src[(j + t * nb) * es];
            if (swap_in) x = make_float2(x.y, x.x);
            int idx = i1 + i2;
            if (idx >= L) idx -= L;
            const cplx w = tw[idx];
            ax += x.x * w.x - x.y * w.y;
            ay += x.x * w.y + x.y * w.x;
            i1 += d1; if (i1 >= L) i1 -= L;
            i2 += d2; if (i2 >= L) i2 -= L;
        }
        out[s * ss + ((j - k) * R + k + u * Ns) * es] = swap_out ? make_float2(ay, ax) : make_float2(ax, ay);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// In-place run-time-schedule transform (any plan): the stage of fft_stage with every butterfly of a thread held in registers
// across a barrier and written back to the SAME buffer -- half the LDS of the ping-pong form, i.e. a second workgroup per CU for
// the big tiles of the chirp-z passes.  A workgroup of T threads may hold up to 8 T elements (NBT = ceil(8 / R) butterflies per
// thread): 1024 threads cover the largest tiles the planner makes (2 x 4096 row points, 2048 x 4 / 1024 x 8 column points).
constexpr int ip_nbt(int R) { return (8 + R - 1) / R; }
template <int R, bool SEQFAST, int PSH, bool TWPOW>
__device__ __forceinline__ void fft_stage_ip(cplx* __restrict__ buf, int L, int Ns, float inv_ns, const cplx* __restrict__ tw,
                                             const dcplx* __restrict__ twd, int nseq, int seq_log2, int es, int ss, bool swap_in,
                                             bool swap_out, bool tw_pow) {
    constexpr int NBT = ip_nbt(R);
    const int nb = L / R;
    const int total = nb * nseq;
    const int twstep = L / (Ns * R);
    cplx v[NBT][R];
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
        const int b = threadIdx.x + n * blockDim.x;
        if (b < total) {
            int s, j;
            if (SEQFAST) { s = b & (nseq - 1); j = b >> seq_log2; }
            else { s = (b >= nb) ? 1 : 0; j = b - s * nb; }
            const int q = (int)(((float)j + 0.5f) * inv_ns);
            const int k = j - q * Ns;
            const cplx* src = buf + s * ss;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const cplx x = src[lds_pad<PSH>(j + t * nb) * es];
                v[n][t] = swap_in ? make_float2(x.y, x.x) : x;
            }
            if (Ns > 1) {
                const int base = k * twstep;
                if (TWPOW && tw_pow) {
                    dcplx w[R];
                    w[1] = twd[base];
#pragma unroll
                    for (int t = 2; t < R; ++t) w[t] = dcmul(w[t >> 1], w[t - (t >> 1)]);
#pragma unroll
                    for (int t = 1; t < R; ++t) v[n][t] = cmul(v[n][t], make_float2((float)w[t].x, (float)w[t].y));
                } else {
#pragma unroll
                    for (int t = 1; t < R; ++t) v[n][t] = cmul(v[n][t], tw[base * t]);
                }
            }
            Bfly<R>::run(v[n]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
        const int b = threadIdx.x + n * blockDim.x;
        if (b < total) {
            int s, j;
            if (SEQFAST) { s = b & (nseq - 1); j = b >> seq_log2; }
            else { s = (b >= nb) ? 1 : 0; j = b - s * nb; }
            const int q = (int)(((float)j + 0.5f) * inv_ns);
            const int k = j - q * Ns;
            cplx* dst = buf + s * ss;
            const int o0 = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const cplx x = v[n][t];
                dst[lds_pad<PSH>(o0 + t * Ns) * es] = swap_out ? make_float2(x.y, x.x) : x;
            }
        }
    }
    __syncthreads();
}

// In-place transform of the sequences in `buf` (radices 2 .. 13; the caller guarantees nseq * L <= 8 * blockDim.x).
template <bool SEQFAST, int PSH = 0, bool TWPOW = true>
__device__ __forceinline__ void lds_fft_ip(cplx* buf, const FftDesc& d, const cplx* __restrict__ tw, int nseq, int seq_log2, int es,
                                           int ss, bool inverse, const dcplx* __restrict__ twd = nullptr) {
    for (int s = 0; s < d.nst; ++s) {
        const bool si = inverse && (s == 0);
        const bool so = inverse && (s == d.nst - 1);
        const int Ns = d.ns[s];
        const float inv = d.inv_ns[s];
        const bool tp = d.tw_pow != 0 && twd != nullptr;
        switch (d.radix[s]) {
            case 2: fft_stage_ip<2, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 3: fft_stage_ip<3, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 4: fft_stage_ip<4, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 5: fft_stage_ip<5, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 7: fft_stage_ip<7, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 8: fft_stage_ip<8, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 9: fft_stage_ip<9, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 11: fft_stage_ip<11, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            default: fft_stage_ip<13, SEQFAST, PSH, TWPOW>(buf, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
        }
    }
}

// Full transform of the sequences in `cur`; result ends in `cur` (pointers are swapped per stage).
// Caller must __syncthreads() after filling `cur`.
template <bool SEQFAST>
__device__ __forceinline__ void lds_fft(cplx*& cur, cplx*& alt, const FftDesc& d, const cplx* __restrict__ tw,
                                        int nseq, int seq_log2, int es, int ss, bool inverse,
                                        const dcplx* __restrict__ twd = nullptr) {
#ifdef EGR_FL_ABL_NOSTAGE
    return;
#endif
    for (int s = 0; s < d.nst; ++s) {
        const bool si = inverse && (s == 0);
        const bool so = inverse && (s == d.nst - 1);
        const int Ns = d.ns[s];
        const float inv = d.inv_ns[s];
        const bool tp = d.tw_pow != 0 && twd != nullptr;
        switch (d.radix[s]) {
            case 2: fft_stage<2, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 3: fft_stage<3, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 4: fft_stage<4, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 5: fft_stage<5, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 7: fft_stage<7, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
#ifdef EGR_RADIX_8_9
            case 8: fft_stage<8, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 9: fft_stage<9, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
#endif
#ifdef EGR_COMPOSITE_RADIX   // measured slower in round 1: one kernel holding every radix needs 190 VGPRs (occupancy 2)
            case 16: fft_stage<16, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 25: fft_stage<25, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
#endif
            case 11: fft_stage<11, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            case 13: fft_stage<13, SEQFAST>(cur, alt, d.L, Ns, inv, tw, twd, nseq, seq_log2, es, ss, si, so, tp); break;
            default: fft_stage_generic<SEQFAST>(cur, alt, d.L, Ns, d.radix[s], tw, nseq, seq_log2, es, ss, si, so); break;
        }
        cplx* t = cur; cur = alt; alt = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Helpers of the wave-local transforms (egr_fatllama_wl.h, k_pz_rowconv_wl): sub-transforms whose lanes all sit in ONE wave exchange
// data through LDS without s_barrier.
__device__ __forceinline__ void wl_wave_sync() {
    // orders one wave's LDS accesses for the compiler (the hardware executes a wave's DS instructions in program order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int R> __device__ __forceinline__ void wl_bfly_inv(cplx (&v)[R]) {      // unnormalised inverse DFT: swap . forward . swap
#pragma unroll
    for (int t = 0; t < R; ++t) v[t] = make_float2(v[t].y, v[t].x);
    Bfly<R>::run(v);
#pragma unroll
    for (int t = 0; t < R; ++t) v[t] = make_float2(v[t].y, v[t].x);
}


}  // namespace egr
