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
// Long inputs add a third factor (state [M1][M2][M3], inner column pass k_col<3>/<4>); lengths without a packed
// plan (odd, large primes) run the chirp-z path further down (k_colz, k_rowconv); egr_spectral_gain reuses the
// passes as a zero-phase filter.  DESIGN.md section 2 has the derivations.
// the register butterflies of this translation unit (packed-real loop kernels) carry their cos / sin constants as two floats: a rounded
// constant is a gain the 800-iteration loop applies in the same direction every iteration (egr_fft_device.h, egr_fatllama_wl.h EGR_WL_HILO)
#ifndef EGR_BFLY_HILO
#define EGR_BFLY_HILO 15
#endif
#include "egr_fatllama_int.h"

namespace egr {

// MODE 0: first   (y -> time threshold -> FFT -> twiddle -> state)          [outermost pass only]
// MODE 1: middle  (state -> twiddle^-1 -> IFFT -> FFT -> twiddle -> state)  [outermost pass only]
// MODE 2: last    (state -> twiddle^-1 -> IFFT -> out = y + d, peak)        [outermost pass only]
// MODE 3: inverse (state -> twiddle^-1 -> IFFT -> state)                    [inner pass of a 3-level plan]
// MODE 4: forward (state -> FFT -> twiddle -> state)                        [inner pass of a 3-level plan]
// MODE 5: last, plain (state -> twiddle^-1 -> IFFT -> out = d)               [spectral-gain filter]
// thr_rel (MODE 0 only, optional): per-channel max|y| as float bits; the time-domain level becomes thr * max|y| (SPEC.md section 3).
// SCHED 0: run-time radix schedule (any plan).  SCHED 2: L = 625 as 25 x 25 with compile-time stages (egr_fft_device.h).
// The compile-time schedules run IN PLACE (one buffer, two barriers per stage): half the LDS of the ping-pong form, so that
// three 512-thread workgroups share a CU (6 waves per SIMD) and the two channels' kernels are fully resident together.
// rows of the in-place 2304-point schedule are stored with one pad element per 16 (lds_pad): conflict-free first-stage writes
#ifndef EGR_FL_ROW_PAD
#define EGR_FL_ROW_PAD 4
#endif
// radix schedules of the two hot lengths (each a product of up to five radices with a register butterfly)
#ifndef EGR_FL_ROW_RADICES
#define EGR_FL_ROW_RADICES 16, 12, 12
#endif
// column tile of the 625-point schedule: columns per workgroup (64-byte = 8, 32-byte = 4 segments per row) and workgroup size
#ifndef EGR_FL_COL_TC
#define EGR_FL_COL_TC 4
#endif
#ifndef EGR_FL_COL_THREADS
#define EGR_FL_COL_THREADS 512
#endif
#ifndef EGR_FL_COL_RADICES
#define EGR_FL_COL_RADICES 25, 25
#endif
// pad of the chirp-z row kernel's LDS rows (0: dense)
#ifndef EGR_FL_CONV_PAD
#define EGR_FL_CONV_PAD 0
#endif
// rows per workgroup and workgroup size of k_rowconv (in place: rows * L <= 8 * threads)
#ifndef EGR_FL_CONV_ROWS
#define EGR_FL_CONV_ROWS 1
#endif
#ifndef EGR_FL_CONV_THREADS
#define EGR_FL_CONV_THREADS 512
#endif
#ifndef EGR_FL_COL_STW_LDS
#define EGR_FL_COL_STW_LDS 1
#endif
#ifndef EGR_FL_SCHED_WAVES
#define EGR_FL_SCHED_WAVES 6
#endif
template <int SCHED>
__device__ __forceinline__ void col_fft(cplx*& cur, cplx*& alt, const ColP& p, int TC, int lg, bool inverse, const cplx* stw) {
    if (SCHED == 2) lds_fft_sched_inplace<true, 0, 625 * EGR_FL_COL_TC, EGR_FL_COL_THREADS, EGR_FL_COL_RADICES>(cur, p.L, stw, TC, lg, TC, 1, inverse);
    else lds_fft<true>(cur, alt, p.f, p.tw, TC, lg, TC, 1, inverse, p.twd);
}

template <int MODE, int SCHED = 0>
__global__ __launch_bounds__(SCHED ? 512 : 1024, SCHED ? EGR_FL_SCHED_WAVES : 1) void k_col(ColP p, long long M, long long N, float thr, cplx* __restrict__ work,
                                              float* __restrict__ out, unsigned* __restrict__ peak_out,
                                              const unsigned* __restrict__ thr_rel = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    __shared__ float red[16];
    // XCD-aware tile order: the dispatcher places block b on XCD b%8; give every XCD a contiguous run
    // of column tiles so the two tiles sharing a 128-byte line hit the same L2.
    const int tile = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int ch = blockIdx.y / p.nplanes, plane = blockIdx.y - ch * p.nplanes;
    const int TC = p.TC, lg = p.TClog2, L = p.L, nc = p.ncols;
    const int c0 = tile * TC;
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* alt = cur + (size_t)L * TC;
    const size_t poff = (size_t)plane * L * nc;
    cplx* W = work + (size_t)ch * M + poff;
    float2* Y = (float2*)(out + (size_t)ch * N) + poff;
    const int nel = L * TC;
    if (MODE == 0 && thr_rel) thr *= __uint_as_float(thr_rel[ch]);
    // the 25 x 26 stage-twiddle table of the 25 25 schedule is copied into LDS (5.2 KB): its reads then cost an LDS round trip
    constexpr int col_radices[] = {EGR_FL_COL_RADICES};
    static_assert(!EGR_FL_COL_STW_LDS || (sizeof(col_radices) / sizeof(int) == 2 && col_radices[0] == 25 && col_radices[1] == 25),
                  "the LDS copy of the stage table is sized for the 25 25 schedule");
    __shared__ cplx stw_lds[(SCHED == 2 && EGR_FL_COL_STW_LDS) ? 25 * 26 : 1];
    const cplx* stw = p.stw;
    if (SCHED == 2 && EGR_FL_COL_STW_LDS) {
        for (int e = threadIdx.x; e < 25 * 26; e += blockDim.x) stw_lds[e] = p.stw[e];
        stw = stw_lds;
    }
    EGR_STAMP(p, 0);

    // scheduled mid pass on full tiles: two adjacent columns per thread and step -- 16-byte state loads / stores and LDS accesses,
    // half the memory instructions of the element-wise loops below
    const bool pairwise = MODE == 1 && SCHED == 2 && c0 + TC <= nc;
    if (pairwise) {
        for (int e = 2 * threadIdx.x; e < nel; e += 2 * blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            const float4 w2 = *(const float4*)(W + (size_t)i * nc + col);
            const cplx a = cmulc(make_float2(w2.x, w2.y), tw2(p.big, (unsigned)col * (unsigned)i));
            const cplx b = cmulc(make_float2(w2.z, w2.w), tw2(p.big, (unsigned)(col + 1) * (unsigned)i));
            *(float4*)(cur + e) = make_float4(a.x, a.y, b.x, b.y);
        }
    } else
    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
        cplx v = make_float2(0.f, 0.f);
        if (col < nc) {
            const size_t g = (size_t)i * nc + col;
            if (MODE == 0) {
                const float2 y = Y[g];
                v.x = fabsf(y.x) > thr ? y.x : 0.f;
                v.y = fabsf(y.y) > thr ? y.y : 0.f;
            } else if (MODE == 4) {
                v = W[g];
            } else {
                v = cmulc(W[g], tw2(p.big, (unsigned)col * (unsigned)i));
            }
        }
        cur[e] = v;
    }
    __syncthreads();
    EGR_STAMP(p, 1);
    if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5) col_fft<SCHED>(cur, alt, p, TC, lg, true, stw);
    EGR_STAMP(p, 2);
    if (MODE == 5) {
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            if (col < nc) Y[(size_t)i * nc + col] = cur[e];
        }
        return;
    }
    if (MODE == 2) {
        float mx = 0.f;
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            if (col < nc) {
                const size_t g = (size_t)i * nc + col;
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
    if (MODE == 3) {
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            if (col < nc) W[(size_t)i * nc + col] = cur[e];
        }
        return;
    }
    col_fft<SCHED>(cur, alt, p, TC, lg, false, stw);
    EGR_STAMP(p, 3);
    if (pairwise) {
        for (int e = 2 * threadIdx.x; e < nel; e += 2 * blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            const float4 v2 = *(const float4*)(cur + e);
            const cplx a = cmul(make_float2(v2.x, v2.y), tw2(p.big, (unsigned)col * (unsigned)i));
            const cplx b = cmul(make_float2(v2.z, v2.w), tw2(p.big, (unsigned)(col + 1) * (unsigned)i));
            *(float4*)(W + (size_t)i * nc + col) = make_float4(a.x, a.y, b.x, b.y);
        }
    } else
    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
        if (col < nc) W[(size_t)i * nc + col] = cmul(cur[e], tw2(p.big, (unsigned)col * (unsigned)i));
    }
    EGR_STAMP(p, 4);
}

// Row-pair kernel: outer indices oa = pair, ob = R - pair of the in-place state.
// MAXONLY: forward transform + real split only, max_k |X[k]|^2 of the channel -> p.max2_out[ch] (nothing is written back): the
// reduction a threshold RELATIVE to the spectrum's maximum needs before any bin can be judged.
// SCHED 1: L = 2304 as 16 x 16 x 9 with compile-time stages.
template <int SCHED>
__device__ __forceinline__ void row_fft(cplx*& cur, cplx*& alt, const RowP& p, int nrows, int L, bool inverse) {
    if (SCHED == 1) lds_fft_sched_inplace<false, EGR_FL_ROW_PAD, 2304 * 2, 512, EGR_FL_ROW_RADICES>(cur, L, p.stw, nrows, 0, 1, lds_pad<EGR_FL_ROW_PAD>(L), inverse);
    else lds_fft<false>(cur, alt, p.f, p.tw, nrows, 0, 1, L, inverse, p.twd);
}

template <bool MAXONLY, int SCHED = 0>
__global__ __launch_bounds__(SCHED ? 512 : 1024, SCHED ? EGR_FL_SCHED_WAVES : 1) void k_row(RowP p, long long M, cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    __shared__ float red[16];
    const int L = p.L, R = p.R;
    const int oa = blockIdx.x;
    const int ob = (R - oa) % R;
    const bool self = (oa == ob);
    const int nrows = self ? 1 : 2;
    const int ch = blockIdx.y;
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* alt = cur + 2 * (size_t)L;
    cplx* W = work + (size_t)ch * M;
    const int ra = (oa % p.Ma) * p.Mb + oa / p.Ma;
    const int rbw = (ob % p.Ma) * p.Mb + ob / p.Ma;
    cplx* ga = W + (size_t)ra * L;
    cplx* gb = W + (size_t)rbw * L;

    constexpr int PSH = SCHED == 1 ? EGR_FL_ROW_PAD : 0;      // LDS row layout: element i at lds_pad<PSH>(i)
    const int Lp = lds_pad<PSH>(L);
    // default hook (hard threshold on |X|^2, two distinct rows) of the scheduled kernel: two (k, M-k) pairs per thread and step
    const bool fast2 = SCHED == 1 && !MAXONLY && !self && !p.phat && !p.band && !p.gain && !p.soft && p.max2 == nullptr;
    dcplx wk_next = p.wk[fast2 ? 2 * threadIdx.x : (threadIdx.x < (unsigned)L ? threadIdx.x : 0)];   // first pair twiddle(s) of this thread
    dcplx wk_next1 = p.wk[fast2 ? 2 * threadIdx.x + 1 : 0];
    float max2v = 0.f;                       // the carried spectrum maximum, requested before the transform (an L2 miss: egr_fatllama_wl.h)
    if (!MAXONLY && p.max2) max2v = fl_max2_read(p.max2, ch);
    EGR_STAMP(p, 0);
    if (SCHED == 1) {                     // 16-byte state loads (two elements; a pair never straddles a pad position)
        for (int e = 2 * threadIdx.x; e < L; e += 2 * blockDim.x) {
            const float4 a = *(const float4*)(ga + e);
            cplx* d = cur + lds_pad<PSH>(e);
            d[0] = make_float2(a.x, a.y);
            d[1] = make_float2(a.z, a.w);
            if (!self) {
                const float4 b = *(const float4*)(gb + e);
                d[Lp] = make_float2(b.x, b.y);
                d[Lp + 1] = make_float2(b.z, b.w);
            }
        }
    } else
    for (int e = threadIdx.x; e < L; e += blockDim.x) {
        cur[lds_pad<PSH>(e)] = ga[e];
        if (!self) cur[Lp + lds_pad<PSH>(e)] = gb[e];
    }
    __syncthreads();
    EGR_STAMP(p, 1);
    row_fft<SCHED>(cur, alt, p, nrows, L, false);
    EGR_STAMP(p, 2);

    // real-split, threshold, un-split on the (k, M-k) pairs; Z[o + R*k] sits at row(o)[k]
    const dcplx wa = tw2d(p.wo, (unsigned)oa);
    int cnt, boff;          // partner of row-a element k is row-b element (boff - k) mod L
    cplx* rb;
    if (!self) { cnt = L; boff = L - 1; rb = cur + Lp; }
    else if (oa == 0) { cnt = L / 2 + 1; boff = L; rb = cur; }
    else { cnt = (L + 1) / 2; boff = L - 1; rb = cur; }
    const float sc = p.inv_M;
    const double scd = p.inv_M_d;
    float thr2 = p.thr2, tlev = p.thr;
    if (!MAXONLY && p.max2) {
        tlev = p.thr * sqrtf(max2v);
        thr2 = tlev * tlev;
    }
    if (!MAXONLY && p.max2_zero && blockIdx.x == 0 && threadIdx.x < EGR_FL_MAX_SUB) fl_max2_clear(p.max2_zero, ch, threadIdx.x);
    const bool variant = !MAXONLY && (p.max2 != nullptr || p.soft);
    float mx2 = 0.f;
    if (fast2) {
        // the same arithmetic as the default branch of the loop below, on elements k2, k2 + 1 of row a and their partners
        // L - 1 - k2, L - 2 - k2 of row b: adjacent LDS elements on both sides (a pair never straddles a pad position)
        auto one = [&](const cplx Za, const cplx Zb, const dcplx wkd, cplx& na, cplx& nb) {
            const dcplx Wkd = dcmul(wa, wkd);
            const cplx Wk = make_float2((float)Wkd.x, (float)Wkd.y);
            const cplx E = make_float2(0.5f * (Za.x + Zb.x), 0.5f * (Za.y - Zb.y));
            const cplx O = make_float2(0.5f * (Za.y + Zb.y), -0.5f * (Za.x - Zb.x));
            const cplx WO = cmul(Wk, O);
            cplx Xk = cadd(E, WO), Xm = csub(E, WO);
            if (!(Xk.x * Xk.x + Xk.y * Xk.y > thr2)) Xk = make_float2(0.f, 0.f);
            if (!(Xm.x * Xm.x + Xm.y * Xm.y > thr2)) Xm = make_float2(0.f, 0.f);
            const cplx E2 = make_float2(0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y + Xm.y));
            const cplx H = make_float2(0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y - Xm.y));
            const cplx O2 = cmulc(H, Wk);
            na = make_float2((float)(scd * (double)(E2.x - O2.y)), (float)(scd * (double)(E2.y + O2.x)));
            nb = make_float2((float)(scd * (double)(E2.x + O2.y)), -(float)(scd * (double)(E2.y - O2.x)));
        };
        for (int k2 = 2 * threadIdx.x; k2 < L; k2 += 2 * blockDim.x) {
            const dcplx w0 = wk_next, w1 = wk_next1;
            if (k2 + 2 * (int)blockDim.x < L) { wk_next = p.wk[k2 + 2 * blockDim.x]; wk_next1 = p.wk[k2 + 2 * blockDim.x + 1]; }
            cplx* da = cur + lds_pad<PSH>(k2);
            cplx* db = rb + lds_pad<PSH>(L - 2 - k2);
            const cplx Za0 = da[0], Za1 = da[1], Zb1 = db[0], Zb0 = db[1];
            cplx na0, nb0, na1, nb1;
            one(Za0, Zb0, w0, na0, nb0);
            one(Za1, Zb1, w1, na1, nb1);
            da[0] = na0; da[1] = na1;
            db[1] = nb0; db[0] = nb1;
        }
    } else
    for (int k2 = threadIdx.x; k2 < cnt; k2 += blockDim.x) {
        // the table entry of the NEXT pair is requested before this one is worked on (the first was requested before the
        // forward transform): the 16-byte L2 round trip per pair is off the dependent chain
        const dcplx wk_cur = wk_next;
        if (k2 + (int)blockDim.x < cnt) wk_next = p.wk[k2 + blockDim.x];
        int pb = boff - k2;
        if (pb >= L) pb -= L;
        const bool same = self && (pb == k2);
        const int ia = lds_pad<PSH>(k2), ib = lds_pad<PSH>(pb);
        const cplx Za = cur[ia];
        const cplx Zb = rb[ib];
        const dcplx Wkd = dcmul(wa, wk_cur);
        const cplx Wk = make_float2((float)Wkd.x, (float)Wkd.y);
        // E = (Za + conj Zb)/2 ; O = (Za - conj Zb)/(2i)
        const cplx E = make_float2(0.5f * (Za.x + Zb.x), 0.5f * (Za.y - Zb.y));
        const cplx O = make_float2(0.5f * (Za.y + Zb.y), -0.5f * (Za.x - Zb.x));
        if (p.phat) {
            // two-for-one: A[k] = E, B[k] = O are the spectra of a and b; R = B conj(A) / (|B conj(A)| + 1e-12) is the spectrum
            // of the real GCC-PHAT sequence, so the new state is W[k] = R, W[M-k] = conj(R) (times 1/M for the inverse passes)
            cplx Rk = cmulc(O, E);
            const float inv = sc / (sqrtf(Rk.x * Rk.x + Rk.y * Rk.y) + 1e-12f);
            Rk.x *= inv; Rk.y *= inv;
            cur[ia] = Rk;
            if (!same) rb[ib] = make_float2(Rk.x, -Rk.y);
            continue;
        }
        const cplx WO = cmul(Wk, O);
        cplx Xk = cadd(E, WO);      // X[k]
        cplx Xm = csub(E, WO);      // conj X[M-k]
        if (MAXONLY) {
            mx2 = fmaxf(mx2, fmaxf(Xk.x * Xk.x + Xk.y * Xk.y, Xm.x * Xm.x + Xm.y * Xm.y));
            continue;
        }
        if (variant) {
            const float mk = sqrtf(Xk.x * Xk.x + Xk.y * Xk.y), mm = sqrtf(Xm.x * Xm.x + Xm.y * Xm.y);
            float gk = mk > tlev ? 1.f : 0.f, gm = mm > tlev ? 1.f : 0.f;
            if (p.soft) {
                if (mk > tlev) gk = 1.f - tlev / mk;
                if (mm > tlev) gm = 1.f - tlev / mm;
            }
            Xk.x *= gk; Xk.y *= gk; Xm.x *= gm; Xm.y *= gm;
            mx2 = fmaxf(mx2, fmaxf(Xk.x * Xk.x + Xk.y * Xk.y, Xm.x * Xm.x + Xm.y * Xm.y));      // what the next iteration's spectrum will hold
        } else if (p.band) {
            const long long k = (long long)oa + (long long)R * k2;
            if (k < p.band_lo) Xk = make_float2(0.f, 0.f);
            if (M - k < p.band_lo) Xm = make_float2(0.f, 0.f);
        } else if (p.gain) {
            const long long k = (long long)oa + (long long)R * k2;
            const float* g = p.gain + (size_t)ch * (M + 1);
            const float gk = g[k], gm = g[M - k];
            Xk.x *= gk; Xk.y *= gk; Xm.x *= gm; Xm.y *= gm;
        } else {
            if (!(Xk.x * Xk.x + Xk.y * Xk.y > thr2)) Xk = make_float2(0.f, 0.f);
            if (!(Xm.x * Xm.x + Xm.y * Xm.y > thr2)) Xm = make_float2(0.f, 0.f);
        }
        const cplx E2 = make_float2(0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y + Xm.y));
        const cplx H = make_float2(0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y - Xm.y));
        const cplx O2 = cmulc(H, Wk);
        // Za' = E2 + i*O2 ; Zb' = conj(E2 - i*O2)
        cur[ia] = make_float2((float)(scd * (double)(E2.x - O2.y)), (float)(scd * (double)(E2.y + O2.x)));
        if (!same) rb[ib] = make_float2((float)(scd * (double)(E2.x + O2.y)), -(float)(scd * (double)(E2.y - O2.x)));
    }
    if (MAXONLY) {
        mx2 = block_max(mx2, red);
        if (threadIdx.x == 0) fl_max2_commit(p.max2_out, ch, mx2);
        return;
    }
    if (p.max2_next) {                       // carried maximum: one commit per workgroup, behind the barrier that is there anyway
        mx2 = wave_max(mx2);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx2;
    }
    __syncthreads();
    if (p.max2_next && threadIdx.x == 0) {
        float r = red[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, red[i]);
        fl_max2_commit(p.max2_next, ch, r);
    }
    EGR_STAMP(p, 3);
    row_fft<SCHED>(cur, alt, p, nrows, L, true);
    EGR_STAMP(p, 4);
    if (SCHED == 1) {
        for (int e = 2 * threadIdx.x; e < L; e += 2 * blockDim.x) {
            const cplx* d = cur + lds_pad<PSH>(e);
            *(float4*)(ga + e) = make_float4(d[0].x, d[0].y, d[1].x, d[1].y);
            if (!self) *(float4*)(gb + e) = make_float4(d[Lp].x, d[Lp].y, d[Lp + 1].x, d[Lp + 1].y);
        }
    } else
    for (int e = threadIdx.x; e < L; e += blockDim.x) {
        ga[e] = cur[lds_pad<PSH>(e)];
        if (!self) gb[e] = cur[Lp + lds_pad<PSH>(e)];
    }
    EGR_STAMP(p, 5);
}

}  // namespace egr
#include "egr_fatllama_wl.h"
namespace egr {

// ------------------------------------------------------------------------------------------------
// Bluestein fallback for lengths the packed real transform cannot take (odd N, large prime factors):
// the exact length-N DFT as a cyclic convolution of length P >= 2N-1 (P smooth):
//   DFT(d)[k] = w[k] * (a (*) b)[k],  a[n] = d[n] w[n],  b[m] = conj(w[m]) for |m| < N,  w[n] = exp(-i pi n^2 / N).
// The convolution runs on the SAME column/row passes with the state holding P complex points:
//   k_colz (outer column pass, hook fused at its natural-order midpoint)  +  k_rowconv (row FFT . x Bhat . row IFFT).
// One loop iteration = spectrum hook (X = w c; threshold; a' = conj(X) w) and time hook (d = Re(w c')/N; a = d w),
// each followed by a convolution: 4 launches over 8P-byte states instead of 2 over 4N-byte ones (slow path).
// ------------------------------------------------------------------------------------------------
// MODE 0: first (y real -> time threshold -> a = d w -> FFT -> twiddle)
// MODE 1: mid   (twiddle^-1 -> IFFT -> hook -> FFT -> twiddle); HOOK 1 = spectrum side, HOOK 2 = time side
// MODE 2: last  (twiddle^-1 -> IFFT -> d = Re(w c)/N -> out = y + d, peak)
// MODE 3: spectrum maximum only (twiddle^-1 -> IFFT -> max |w c|^2 -> cp.max2_out[ch]; nothing written back)
template <int MODE, int HOOK>
__global__ __launch_bounds__(1024) void k_colz(ColP p, ChirpP cp, long long P, float thr, float thr2,
                                               cplx* __restrict__ work, float* __restrict__ out,
                                               unsigned* __restrict__ peak_out, const unsigned* __restrict__ thr_rel = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    __shared__ float red[16];                 // one slot per wave: up to 1024 threads
    const int tile = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= p.ntiles) return;
    const int ch = blockIdx.y;
    const int TC = p.TC, lg = p.TClog2, L = p.L, nc = p.ncols;
    const int c0 = tile * TC;
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* alt = cur + (size_t)L * TC;
    cplx* W = work + (size_t)ch * P;
    float* Y = out + (size_t)ch * cp.N;
    const int nel = L * TC;
    const unsigned long long N = cp.N;
    if (MODE == 0 && thr_rel) thr *= __uint_as_float(thr_rel[ch]);

    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
        cplx v = make_float2(0.f, 0.f);
        if (col < nc) {
            const unsigned long long n = (unsigned long long)i * nc + col;
            if (MODE == 0) {
                if (n < N) {
                    const float y = Y[n];
                    const float d0 = fabsf(y) > thr ? y : 0.f;
                    const cplx w = chirp(cp, n);
                    v = make_float2(d0 * w.x, d0 * w.y);
                }
            } else {
                v = cmulc(W[n], tw2(p.big, (unsigned)col * (unsigned)i));
            }
        }
        cur[e] = v;
    }
    __syncthreads();
    if (MODE != 0) lds_fft<true>(cur, alt, p.f, p.tw, TC, lg, TC, 1, true, p.twd);
    if (MODE == 3) {
        float mx = 0.f;
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            const unsigned long long n = (unsigned long long)i * nc + col;
            if (col < nc && n < N) {
                const cplx X = cmul(chirp(cp, n), cur[e]);
                mx = fmaxf(mx, X.x * X.x + X.y * X.y);
            }
        }
        mx = block_max(mx, red);
        if (threadIdx.x == 0) fl_max2_commit(cp.max2_out, ch, mx);
        return;
    }
    if (MODE == 1 && HOOK == 1 && cp.max2) {
        thr *= sqrtf(fl_max2_read(cp.max2, ch));
        thr2 = thr * thr;
    }
    if (MODE == 2) {
        float mx = 0.f;
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            const unsigned long long n = (unsigned long long)i * nc + col;
            if (col < nc && n < N) {
                const cplx w = chirp(cp, n);
                const cplx cc = cur[e];
                const float d = (w.x * cc.x - w.y * cc.y) * cp.inv_N;
                const float o = __fadd_rn(Y[n], d);
                Y[n] = o;
                mx = fmaxf(mx, fabsf(o));
            }
        }
        mx = block_max(mx, red);
        if (threadIdx.x == 0) atomic_max_abs(peak_out + ch, mx);
        return;
    }
    if (MODE == 1) {
        for (int e = threadIdx.x; e < nel; e += blockDim.x) {
            const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
            const unsigned long long n = (unsigned long long)i * nc + col;
            cplx v = make_float2(0.f, 0.f);
            if (col < nc && n < N) {
                const cplx w = chirp(cp, n);
                const cplx cc = cur[e];
                if (HOOK == 1) {                 // X = w c ; threshold ; a' = conj(X) w
                    cplx X = cmul(w, cc);
                    if (cp.band) {
                        if ((n < N - n ? n : N - n) < cp.band_lo) X = make_float2(0.f, 0.f);
                    } else if (cp.soft) {
                        const float mg = sqrtf(X.x * X.x + X.y * X.y);
                        const float g = mg > thr ? 1.f - thr / mg : 0.f;
                        X.x *= g; X.y *= g;
                    } else if (!(X.x * X.x + X.y * X.y > thr2)) X = make_float2(0.f, 0.f);
                    v = cmul(make_float2(X.x, -X.y), w);
                } else {                         // d = Re(w c)/N ; a = d w
                    const float d = (w.x * cc.x - w.y * cc.y) * cp.inv_N;
                    v = make_float2(d * w.x, d * w.y);
                }
            }
            cur[e] = v;
        }
        __syncthreads();
    }
    lds_fft<true>(cur, alt, p.f, p.tw, TC, lg, TC, 1, false, p.twd);
    for (int e = threadIdx.x; e < nel; e += blockDim.x) {
        const int c = e & (TC - 1), i = e >> lg, col = c0 + c;
        if (col < nc) W[(size_t)i * nc + col] = cmul(cur[e], tw2(p.big, (unsigned)col * (unsigned)i));
    }
}

// rows r0 = EGR_FL_CONV_ROWS * blockIdx.x (, r0 + 1) of R rows of length L.  CONV: FFT . x bhat[row][k] . IFFT ; else FFT . x scale
// (used once to build bhat itself).
template <bool CONV>
__global__ __launch_bounds__(EGR_FL_CONV_THREADS) void k_rowconv(FftDesc f, int L, int R, const cplx* __restrict__ tw,
                                                  const cplx* __restrict__ bhat, float scale, long long P,
                                                  cplx* __restrict__ work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    const int r0 = EGR_FL_CONV_ROWS * blockIdx.x;
    const int nrows = (r0 + 1 < R && EGR_FL_CONV_ROWS == 2) ? 2 : 1;
    constexpr int PSH = EGR_FL_CONV_PAD;                 // rows in LDS: element i at lds_pad<PSH>(i), stages in place
    const int Lp = lds_pad<PSH>(L);
    cplx* cur = (cplx*)EGR_LDS_BASE(smem);
    cplx* g = work + (size_t)blockIdx.y * P + (size_t)r0 * L;
    // 16-byte accesses (two elements per thread and step) when rows start 16-byte aligned; one row per workgroup then
    const bool pairwise = PSH == 0 && EGR_FL_CONV_ROWS == 1 && (L & 1) == 0;
    if (pairwise) {
        for (int e = 2 * threadIdx.x; e < L; e += 2 * blockDim.x) *(float4*)(cur + e) = *(const float4*)(g + e);
    } else
    for (int e = threadIdx.x; e < nrows * L; e += blockDim.x) {
        const int r = e >= L ? 1 : 0, i = e - r * L;
        cur[r * Lp + lds_pad<PSH>(i)] = g[e];
    }
    __syncthreads();
    lds_fft_ip<false, PSH, false>(cur, f, tw, nrows, 0, 1, Lp, false);
    if (CONV) {
        const cplx* bh = bhat + (size_t)r0 * L;
        if (pairwise) {
            for (int e = 2 * threadIdx.x; e < L; e += 2 * blockDim.x) {
                const float4 c = *(const float4*)(cur + e), b = *(const float4*)(bh + e);
                const cplx u = cmul(make_float2(c.x, c.y), make_float2(b.x, b.y)), v = cmul(make_float2(c.z, c.w), make_float2(b.z, b.w));
                *(float4*)(cur + e) = make_float4(u.x, u.y, v.x, v.y);
            }
        } else
        for (int e = threadIdx.x; e < nrows * L; e += blockDim.x) {
            const int r = e >= L ? 1 : 0, i = e - r * L, a = r * Lp + lds_pad<PSH>(i);
            cur[a] = cmul(cur[a], bh[e]);
        }
        __syncthreads();
        lds_fft_ip<false, PSH, false>(cur, f, tw, nrows, 0, 1, Lp, true);
        if (pairwise) {
            for (int e = 2 * threadIdx.x; e < L; e += 2 * blockDim.x) *(float4*)(g + e) = *(const float4*)(cur + e);
        } else
        for (int e = threadIdx.x; e < nrows * L; e += blockDim.x) {
            const int r = e >= L ? 1 : 0, i = e - r * L;
            g[e] = cur[r * Lp + lds_pad<PSH>(i)];
        }
    } else {
        for (int e = threadIdx.x; e < nrows * L; e += blockDim.x) {
            const int r = e >= L ? 1 : 0, i = e - r * L;
            const cplx c = cur[r * Lp + lds_pad<PSH>(i)];
            g[e] = make_float2(c.x * scale, c.y * scale);
        }
    }
}

// b[j] = conj(w[j]) for j < N, b[P-j] = conj(w[j]) for 0 < j < N, zero elsewhere
__global__ __launch_bounds__(256) void k_chirp_b(ChirpP cp, long long P, cplx* __restrict__ b) {
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < P; j += (long long)gridDim.x * blockDim.x) {
        long long m = -1;
        if ((unsigned long long)j < cp.N) m = j;
        else if ((unsigned long long)(P - j) < cp.N) m = P - j;
        cplx v = make_float2(0.f, 0.f);
        if (m >= 0) {
            const cplx w = chirp(cp, (unsigned long long)m);
            v = make_float2(w.x, -w.y);
        }
        b[j] = v;
    }
}

// x (optionally PCM_16-quantised) -> y = linear up-rate by f, per-channel max|x_q|.
// zero_stuff: y[i*f] = x[i], zeros between (SPEC.md "interp").  peak_y[ch] = max|y| (what a relative time-domain threshold refers to).
// linspace (SPEC.md "interp"): y[j] = interp(j (n_in - 1) / (n_out - 1), arange(n_in), x) -- numpy.interp on the endpoint-inclusive
// grid numpy.linspace(0, n_in - 1, n_out), evaluated like numpy (double: slope * (pos - i) + x[i], last point exactly x[n_in - 1]);
// n_out need not be a multiple of n_in (factor_mode "ratio_then_int").
__global__ __launch_bounds__(256) void k_prepare(const float* __restrict__ x, float* __restrict__ y, long long n_in,
                                                  int f, int pcm_in, int zero_stuff, unsigned* __restrict__ peak_in,
                                                  unsigned* __restrict__ peak_y, int linspace = 0, long long n_out = 0) {
#pragma clang fp contract(off)      // the roundings below are the PCM arithmetic of the reference, written out
    __shared__ float red[8];
    const int ch = blockIdx.y;
    const float* xc = x + (size_t)ch * n_in;
    float* yc = y + (size_t)ch * n_in * f;
    const float ff = (float)f;
    float mx = 0.f, my = 0.f;
    if (linspace) {
        float* yo = y + (size_t)ch * n_out;
        auto q = [&](float a) -> float {
            if (!pcm_in) return a;
            long long qa = (long long)rintf(__fmul_rn(a, 32767.0f));
            return (float)(((qa + 32768) & 65535) - 32768);
        };
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += (long long)gridDim.x * blockDim.x)
            mx = fmaxf(mx, fabsf(q(xc[i])));
        const double step = n_out > 1 ? (double)(n_in - 1) / (double)(n_out - 1) : 0.0;
        for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += (long long)gridDim.x * blockDim.x) {
            float v;
            if (j == n_out - 1 && n_out > 1) {
                v = q(xc[n_in - 1]);
            } else {
                const double pos = (double)j * step;
                long long i0 = (long long)pos;
                if (i0 > n_in - 2) i0 = n_in - 2;
                if (i0 < 0) i0 = 0;
                const double a = (double)q(xc[i0]), b = n_in > 1 ? (double)q(xc[i0 + 1]) : a;
                v = (float)((b - a) * (pos - (double)i0) + a);
            }
            yo[j] = v;
            my = fmaxf(my, fabsf(v));
        }
        mx = block_max(mx, red);
        my = block_max(my, red);
        if (threadIdx.x == 0) {
            atomic_max_abs(peak_in + ch, mx);
            atomic_max_abs(peak_y + ch, my);
        }
        return;
    }
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
        if (zero_stuff) {
            yc[i * f] = a;
            for (int j = 1; j < f; ++j) yc[i * f + j] = 0.f;
            my = fmaxf(my, fabsf(a));
        } else if (i + 1 < n_in) {
            for (int j = 0; j < f; ++j) {
                const float t = __fdiv_rn((float)j, ff);
                const float u = __fsub_rn(1.0f, t);
                const float v = __fadd_rn(__fmul_rn(u, a), __fmul_rn(t, b));
                yc[i * f + j] = v;
                my = fmaxf(my, fabsf(v));
            }
        } else {
            for (int j = 0; j < f; ++j) yc[i * f + j] = 0.f;   // upstream leaves the last sample's slots zero
        }
    }
    mx = block_max(mx, red);
    my = block_max(my, red);
    if (threadIdx.x == 0) {
        atomic_max_abs(peak_in + ch, mx);
        atomic_max_abs(peak_y + ch, my);
    }
}

// max_iter == 0 path: out = y + (|y|>thr ? y : 0), peaks
__global__ __launch_bounds__(256) void k_noiter(float* __restrict__ y, long long N, float thr,
                                                 unsigned* __restrict__ peak_out, const unsigned* __restrict__ thr_rel) {
    __shared__ float red[8];
    const int ch = blockIdx.y;
    if (thr_rel) thr *= __uint_as_float(thr_rel[ch]);
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
// joint_ext (optional): the peak of channels this plan does NOT hold -- the other ranks' of a channel-parallel run
// (egr_fatllama_joint_peak on every rank, one all-reduce(MAX) of that float, egr_fatllama_finalize): max is exact, so the result is
// the single-plan result bit for bit.  joint_dst (optional, k_joint_peak): where this plan's own joint peak goes; nothing else is done.
__global__ __launch_bounds__(256) void k_finalize(float* __restrict__ out, long long N, int C, unsigned flags,
                                                   const unsigned* __restrict__ peak_in,
                                                   const unsigned* __restrict__ peak_out,
                                                   const float* __restrict__ joint_ext, float* __restrict__ joint_dst) {
#pragma clang fp contract(off)      // the roundings below are the PCM arithmetic of the reference, written out
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
    if (joint_dst) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *joint_dst = joint;
        return;
    }
    if (joint_ext) joint = fmaxf(joint, *joint_ext);
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

int fl_upload(egr_fatllama_plan* p, const std::vector<float2>& h, const cplx** d) {
    void* ptr = nullptr;
    EGR_HIP(hipMalloc(&ptr, h.size() * sizeof(float2)));
    p->dev_allocs.push_back(ptr);
    EGR_HIP(hipMemcpy(ptr, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    *d = (const cplx*)ptr;
    return EGR_OK;
}

int fl_upload_d(egr_fatllama_plan* p, int L, const dcplx** d) {
    std::vector<double2> h;
    make_twiddles_d(h, L, 1, L);
    void* ptr = nullptr;
    EGR_HIP(hipMalloc(&ptr, h.size() * sizeof(double2)));
    p->dev_allocs.push_back(ptr);
    EGR_HIP(hipMemcpy(ptr, h.data(), h.size() * sizeof(double2), hipMemcpyHostToDevice));
    *d = (const dcplx*)ptr;
    return EGR_OK;
}

// exp(-2 pi i j num / den), j < count, as a double-precision device table
int fl_upload_dtab(egr_fatllama_plan* p, int64_t count, int64_t num, int64_t den, const dcplx** d) {
    std::vector<double2> h;
    make_twiddles_d(h, count, num, den);
    void* ptr = nullptr;
    EGR_HIP(hipMalloc(&ptr, h.size() * sizeof(double2)));
    p->dev_allocs.push_back(ptr);
    EGR_HIP(hipMemcpy(ptr, h.data(), h.size() * sizeof(double2), hipMemcpyHostToDevice));
    *d = (const dcplx*)ptr;
    return EGR_OK;
}

// butterfly-ordered stage tables of a compile-time schedule (R0, R1, R2; R2 = 1: two stages): for stage s >= 1 with
// Ns = product of the earlier radices, row k < Ns holds W_(Ns R)^(k t), t = 0 .. R-1, padded to an even entry count
int fl_upload_sched_tables(egr_fatllama_plan* p, std::initializer_list<int> radices, const cplx** d) {
    std::vector<float2> h;
    const long double two_pi = 6.283185307179586476925286766559L;
    auto add = [&](int Ns, int R) {
        const int RS = (R + 1) & ~1;
        for (int k = 0; k < Ns; ++k)
            for (int t = 0; t < RS; ++t) {
                const long double ang = t < R ? -two_pi * (long double)((long long)k * t % ((long long)Ns * R)) / (long double)((long long)Ns * R) : 0.0L;
                h.push_back(make_float2((float)cosl(ang), (float)sinl(ang)));
            }
    };
    int ns = 1, s = 0;
    for (int r : radices) {
        if (s > 0 && r > 1) add(ns, r);
        ns *= r;
        ++s;
    }
    return fl_upload(p, h, d);
}

// tables for W_T^r, r < T
int fl_make_tw2(egr_fatllama_plan* p, int64_t T, Tw2* out) {
    int sh = 0;
    while ((1LL << (2 * sh)) < T) ++sh;          // 2^sh >= sqrt(T)
    int rc;
    if ((rc = fl_upload_dtab(p, (T >> sh) + 1, 1LL << sh, T, &out->hi))) return rc;
    if ((rc = fl_upload_dtab(p, 1LL << sh, 1, T, &out->lo))) return rc;
    out->sh = sh;
    return EGR_OK;
}

static void fill_info(const FlSplit& sp, int64_t info[EGR_FL_INFO_LEN]) {
    info[0] = 1;
    info[3] = sp.M1; info[4] = sp.M2; info[5] = sp.TC; info[6] = sp.f1.nst; info[7] = sp.f2.nst;
    for (int i = 0; i < EGR_MAX_STAGES; ++i) { info[8 + i] = sp.f1.radix[i]; info[22 + i] = sp.f2.radix[i]; }
    info[36] = (int64_t)sp.lds_col;
    info[37] = (int64_t)sp.lds_row;
    info[38] = sp.M3;
    info[39] = sp.levels;
}

static bool bluestein_length(int64_t want, FlSplit* sp_out);
static bool pz_length(int64_t D, FlSplit* sp_out);
// paired chirp-z kind for N real samples: 1 = even/odd packing (N even, D = N / 2), 2 = channel pairs (N odd, D = N)
static inline int pz_kind_for(int64_t N) { return (N & 1) ? 2 : 1; }
// LDS of k_rowconv: two rows, stages in place
static size_t rowconv_lds(int L) { return EGR_LDS((size_t)EGR_FL_CONV_ROWS * (L + (EGR_FL_CONV_PAD ? L >> EGR_FL_CONV_PAD : 0)) * sizeof(cplx)); }
// workgroup size of the chirp-z loop kernels (their tiles take most of a CU's LDS: one workgroup per CU, so a large one)
static int blue_threads() {
    static const int t = [] { const char* e = getenv("EGR_FL_BLUE_THREADS"); const int v = e ? atoi(e) : 1024; return (v == 256 || v == 512 || v == 1024) ? v : 1024; }();
    return t;
}

static const char* kUnsupported =
    "length %lld unsupported: needs even N whose half factors into 2 or 3 lengths (outer columns <= 2048, inner columns <= 1024, row <= 4096) with "
    "prime factors <= 13";

extern "C" int egr_fatllama_plan_query(int64_t n_in, int factor, int m1_hint, int64_t info[EGR_FL_INFO_LEN]) {
    EGR_CHECK(info != nullptr, EGR_ERR_ARG, "info is null");
    memset(info, 0, sizeof(int64_t) * EGR_FL_INFO_LEN);
    EGR_CHECK(n_in >= 1 && factor >= 1, EGR_ERR_ARG, "n_in=%lld factor=%d out of range", (long long)n_in, factor);
    const int64_t N = n_in * factor;
    FlSplit sp = plan_split(N, m1_hint, 0);
    info[1] = N;
    info[2] = N / 2;
    if (!sp.ok) {
        const int kind = pz_kind_for(N);
        if (N >= 2 && pz_length(kind == 1 ? N / 2 : N, &sp)) {      // paired chirp-z over P = sp.M complex points per state
            fill_info(sp, info);
            info[0] = 2;
            info[2] = sp.M;
            info[40] = kind;
            info[41] = kind == 1 ? N / 2 : N;
            return EGR_OK;
        }
        set_error(kUnsupported, (long long)N);
        return EGR_ERR_UNSUPPORTED;
    }
    fill_info(sp, info);
    return EGR_OK;
}

extern "C" int egr_fatllama_plan_destroy(egr_fatllama_plan* p) {
    if (!p) return EGR_OK;
    hipDeviceSynchronize();          // nothing of the plan may still be in flight when its graph, streams and buffers go
    pz_destroy(p);
    for (void* q : p->dev_allocs) hipFree(q);
    hipFree(p->d_work); hipFree(p->d_peaks); hipFree(p->d_bhat); hipFree(p->d_max2);
    if (p->gexec) hipGraphExecDestroy(p->gexec);
    if (p->cap) hipStreamDestroy(p->cap);
    if (p->side && p->side_owned) hipStreamDestroy(p->side);
    if (p->ev_fork) hipEventDestroy(p->ev_fork);
    if (p->ev_join) hipEventDestroy(p->ev_join);
    for (auto e : p->ev) hipEventDestroy(e);
    delete p;
    return EGR_OK;
}

static int build_plan(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, const FlSplit& sp,
                      int64_t bluestein_n = 0, int pz_kind = 0, int64_t n_out = 0) {
    egr_fatllama_plan* p = new egr_fatllama_plan();
    p->n_in = n_in; p->C = channels; p->factor = factor; p->sp = sp; p->profiling = false;
    p->n_out = n_out > 0 ? n_out : n_in * factor;
    p->bluestein = bluestein_n > 0;
    p->pz_kind = pz_kind; p->pz = nullptr;
    p->nstreams = 2;
    if (const char* e = getenv("EGR_FL_STREAMS")) { const int t = atoi(e); if (t == 1 || t == 2) p->nstreams = t; }
    p->side = nullptr; p->side_owned = 0; p->ev_fork = nullptr; p->ev_join = nullptr;
    p->cap = nullptr; p->gexec = nullptr; p->g_out = nullptr; p->g_thr = 0.f; p->g_groups = 0; p->g_iter_odd = 0;
    p->use_graph = !(getenv("EGR_FL_GRAPH") && atoi(getenv("EGR_FL_GRAPH")) == 0);
    p->threads = 512;    // 16 waves per CU at 2 workgroups per CU: measured 1.4x over 256
    if (const char* e = getenv("EGR_FL_THREADS")) { const int t = atoi(e); if (t == 256 || t == 512 || t == 1024) p->threads = t; }
    p->d_bhat = nullptr;
    p->d_work = nullptr;
    p->d_peaks = nullptr;
    p->d_max2 = nullptr; p->max2_cap = 0;
    hipGetDevice(&p->device);
    const int64_t M = sp.M, N = sp.N;
    int rc = EGR_OK;
    std::vector<float2> h;
    auto fail = [&](int code) { egr_fatllama_plan_destroy(p); return code; };
    memset(&p->colA, 0, sizeof(ColP)); memset(&p->colB, 0, sizeof(ColP)); memset(&p->row, 0, sizeof(RowP));
    // ---- outer column pass A: L = M1, columns = M / M1 ----
    ColP& a = p->colA;
    a.f = sp.f1; a.L = sp.M1; a.ncols = (int)(M / sp.M1); a.nplanes = 1;
    a.TC = sp.TC; a.TClog2 = sp.TClog2; a.ntiles = ceil_div(a.ncols, a.TC); a.tiles_per_xcd = ceil_div(a.ntiles, 8);
    make_twiddles(h, sp.M1, 1, sp.M1);
    if ((rc = fl_upload(p, h, &a.tw))) return fail(rc);
    if ((rc = fl_upload_d(p, sp.M1, &a.twd))) return fail(rc);
    if ((rc = fl_make_tw2(p, M, &a.big))) return fail(rc);
    // ---- inner column pass B (3 levels): per k1 plane, L = M2, columns = M3 ----
    RowP& r = p->row;
    if (sp.levels == 3) {
        ColP& b = p->colB;
        b.f = sp.f2; b.L = sp.M2; b.ncols = sp.M3; b.nplanes = sp.M1;
        b.TC = sp.TCb; b.TClog2 = sp.TCblog2; b.ntiles = ceil_div(b.ncols, b.TC); b.tiles_per_xcd = ceil_div(b.ntiles, 8);
        make_twiddles(h, sp.M2, 1, sp.M2);
        if ((rc = fl_upload(p, h, &b.tw))) return fail(rc);
        if ((rc = fl_upload_d(p, sp.M2, &b.twd))) return fail(rc);
        if ((rc = fl_make_tw2(p, (int64_t)sp.M2 * sp.M3, &b.big))) return fail(rc);
        r.f = sp.f3; r.L = sp.M3; r.R = sp.M1 * sp.M2; r.Ma = sp.M1; r.Mb = sp.M2;
    } else {
        r.f = sp.f2; r.L = sp.M2; r.R = sp.M1; r.Ma = sp.M1; r.Mb = 1;
    }
    make_twiddles(h, r.L, 1, r.L);
    if ((rc = fl_upload(p, h, &r.tw))) return fail(rc);
    if ((rc = fl_upload_d(p, r.L, &r.twd))) return fail(rc);
    if ((rc = fl_upload_dtab(p, r.L, 1, 2 * (int64_t)r.L, &r.wk))) return fail(rc);
    {   // W_N^o for o < R, as hi/lo tables over the range [0, R)
        int sh = 0;
        while ((1LL << (2 * sh)) < r.R) ++sh;
        if ((rc = fl_upload_dtab(p, ((int64_t)r.R >> sh) + 1, 1LL << sh, N, &r.wo.hi))) return fail(rc);
        if ((rc = fl_upload_dtab(p, 1LL << sh, 1, N, &r.wo.lo))) return fail(rc);
        r.wo.sh = sh;
    }
    r.inv_M = (float)(1.0 / (double)M);
    r.inv_M_d = 1.0 / (double)M;
    // compile-time schedules for the hot factor lengths (EGR_FL_SCHED=0: always the run-time schedule)
    p->row_sched = p->col_sched = 0;
    {
        const bool off = getenv("EGR_FL_SCHED") && atoi(getenv("EGR_FL_SCHED")) == 0;
        if (!off && !p->bluestein && r.L == 2304) { if ((rc = fl_upload_sched_tables(p, {EGR_FL_ROW_RADICES}, &r.stw))) return fail(rc); p->row_sched = 1; }
        if (!off && !p->bluestein && a.L == 625 && a.TC >= 2 && a.TC <= EGR_FL_COL_TC) { if ((rc = fl_upload_sched_tables(p, {EGR_FL_COL_RADICES}, &a.stw))) return fail(rc); p->col_sched = 2; }
    }
    // the two-barrier loop kernels (egr_fatllama_wl.h) for rows of 2304 / outer columns of 625 points; EGR_FL_WL=0 keeps the
    // stage-by-stage kernels
    p->wl_row = p->wl_col = 0;
    {
        const bool off = getenv("EGR_FL_WL") && atoi(getenv("EGR_FL_WL")) == 0;
        auto table = [&](int n, int m, int T, const cplx** d, const cplx** dlo) {          // [n][m]: W_T^(i j), and its low parts
            std::vector<float2> t, tl;
            t.resize((size_t)n * m); tl.resize((size_t)n * m);
            const long double two_pi = 6.283185307179586476925286766559L;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < m; ++j) {
                    const long double ang = -two_pi * (long double)(((long long)i * j) % T) / (long double)T;
                    const long double c = cosl(ang), sn = sinl(ang);
                    const float2 h = make_float2((float)c, (float)sn);
                    t[(size_t)i * m + j] = h;
                    tl[(size_t)i * m + j] = make_float2((float)(c - (long double)h.x), (float)(sn - (long double)h.y));
                }
            const int rc2 = fl_upload(p, t, d);
            return rc2 ? rc2 : fl_upload(p, tl, dlo);
        };
        p->wl_row_entry = nullptr;
        if (!off && !p->bluestein)
            for (const WlRowEntry& e : kWlRows)
                if (e.L == r.L) {
                    const int qq = e.q * e.q;
                    if ((rc = table(qq, e.n1, e.L, &p->wl_rt.t1, &p->wl_rt.t1l))) return fail(rc);
                    if ((rc = table(e.q, e.q, qq, &p->wl_rt.t2, &p->wl_rt.t2l))) return fail(rc);
                    if ((rc = fl_upload_dtab(p, e.q, 1, qq, &p->wl_rt.t2d))) return fail(rc);
                    const long double ang = -3.14159265358979323846264338327950288L / (long double)e.q;         // W_(2Q)
                    p->wl_rt.hook_step = make_double2((double)cosl(ang), (double)sinl(ang));
                    p->wl_row_entry = &e;
                    p->wl_row = 1;
                }
        if (!off && !p->bluestein && wl_col_radix(a.L)) {
            const int r = wl_col_radix(a.L);
            if ((rc = table(r, r, r * r, &p->wl_ct.t3, &p->wl_ct.t3l))) return fail(rc);
            p->wl_col = 1;
        }
        p->wl_inner = 0; p->wl_it = nullptr;
        if (!off && !p->bluestein && sp.levels == 3 && wl_inner_supported(sp.M2)) {
            int la, lb, tcw;
            wl_inner_geometry(sp.M2, &la, &lb, &tcw);
            const cplx* itl = nullptr;
            if ((rc = table(lb, la, sp.M2, &p->wl_it, &itl))) return fail(rc);
            p->wl_colB = p->colB;
            p->wl_colB.TC = tcw; p->wl_colB.TClog2 = 0;
            p->wl_colB.ntiles = ceil_div(p->colB.ncols, tcw); p->wl_colB.tiles_per_xcd = ceil_div(p->wl_colB.ntiles, 8);
            p->wl_inner = 1;
        }
    }
    const int nstates = pz_kind == 2 ? (channels + 1) / 2 : channels;      // a paired chirp-z state of kind 2 carries two channels
    if (hipMalloc((void**)&p->d_work, (size_t)nstates * M * sizeof(float2)) != hipSuccess ||
        hipMalloc((void**)&p->d_peaks, 3 * channels * sizeof(unsigned)) != hipSuccess) {
        set_error("hipMalloc of the %lld-byte loop state failed", (long long)(channels * M * 8));
        return fail(EGR_ERR_ALLOC);
    }
    // dynamic LDS above the 64 KiB default needs an explicit opt-in per kernel.  The attribute is a process-wide cap per kernel, so it
    // is always raised to the CU's 160 KiB (a later, smaller plan must not lower it under an earlier plan's needs); what a plan
    // actually requests is checked here.
    const int lc = EGR_LDS_MAX, lrow = EGR_LDS_MAX;
    if (!pz_kind && (std::max(sp.lds_col, sp.lds_colb) > (size_t)EGR_LDS_MAX || sp.lds_row > (size_t)EGR_LDS_MAX)) {
        set_error("plan needs %zu / %zu bytes of LDS per workgroup (limit %d)", std::max(sp.lds_col, sp.lds_colb), sp.lds_row, EGR_LDS_MAX);
        return fail(EGR_ERR_UNSUPPORTED);
    }
    hipError_t e = hipSuccess;
    e = hipFuncSetAttribute((const void*)k_col<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_row<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lrow);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_row<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lrow);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_row<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lrow);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_row<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lrow);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_col<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_colz<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_colz<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_colz<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_colz<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_colz<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lc);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_rowconv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lrow);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_rowconv<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lrow);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) -> %s", hipGetErrorString(e));
        return fail(EGR_ERR_HIP);
    }
    if (p->pz_kind) {
        if ((rc = pz_build(p, p->pz_kind))) return fail(rc);
    } else if (p->bluestein) {
        // chirp tables for W_(2N)^r and bhat = FFT_P(b)/P computed once with the plan's own passes
        ChirpP& c = p->chirp;
        c.N = (unsigned long long)bluestein_n;
        c.inv_N = (float)(1.0 / (double)bluestein_n);
        c.inv_2N_d = 1.0 / (2.0 * (double)bluestein_n);
        if ((rc = fl_make_tw2(p, 2 * bluestein_n, &c.w))) return fail(rc);
        if (hipMalloc((void**)&p->d_bhat, (size_t)M * sizeof(float2)) != hipSuccess) {
            set_error("hipMalloc of the %lld-byte chirp spectrum failed", (long long)(M * 8));
            return fail(EGR_ERR_ALLOC);
        }
        const dim3 blk(256);
        hipLaunchKernelGGL(k_chirp_b, dim3(2048), blk, 0, 0, c, (long long)M, p->d_bhat);
        hipLaunchKernelGGL(k_col<4>, dim3(8 * a.tiles_per_xcd, 1), blk, EGR_LDS(sp.lds_col), 0, a, (long long)M, (long long)N, 0.f, p->d_bhat,
                           (float*)nullptr, (unsigned*)nullptr);
        if (sp.levels == 3)
            hipLaunchKernelGGL(k_col<4>, dim3(8 * p->colB.tiles_per_xcd, p->colB.nplanes), blk, EGR_LDS(sp.lds_colb), 0, p->colB,
                               (long long)M, (long long)N, 0.f, p->d_bhat, (float*)nullptr, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_rowconv<false>, dim3((r.R + EGR_FL_CONV_ROWS - 1) / EGR_FL_CONV_ROWS, 1), dim3(EGR_FL_CONV_THREADS), rowconv_lds(r.L), 0, r.f, r.L, r.R, r.tw,
                           (const cplx*)nullptr, (float)(1.0 / (double)M), (long long)M, p->d_bhat);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
            set_error("building the Bluestein chirp spectrum failed");
            return fail(EGR_ERR_HIP);
        }
    }
    {   // stage twiddles W^2..W^(r-1) by multiplication from one loaded W^1 (default) or all loaded (EGR_FL_TWPOW=0)
        int v = 1;
        if (const char* e2 = getenv("EGR_FL_TWPOW")) v = atoi(e2);
        p->colA.f.tw_pow = v; p->colB.f.tw_pow = v; p->row.f.tw_pow = v;
    }
    *out = p;
    return EGR_OK;
}

// smallest P >= want whose packed plan (M = P) exists; P has only the prime factors 2, 3, 5, 7
static bool bluestein_length(int64_t want, FlSplit* sp_out) {
    // smallest cost, not smallest length: a three-level plan runs ~1.6x the passes of a two-level one per transform
    int64_t best = -1;
    double best_cost = 1e300;
    FlSplit best_sp;
    for (int64_t p7 = 1; p7 <= 8 * want; p7 *= 7)
        for (int64_t p5 = p7; p5 <= 8 * want; p5 *= 5)
            for (int64_t p3 = p5; p3 <= 8 * want; p3 *= 3) {
                int64_t v = p3;
                while (v < want) v *= 2;
                for (int rep = 0; rep < 2; ++rep, v *= 2) {          // v and 2v: the first may not factor into tiles
                    if ((double)v >= best_cost) break;
                    FlSplit sp = plan_split(2 * v, 0, 0, 2048);
                    if (!sp.ok) continue;
                    const double cost = (double)v * (sp.levels == 3 ? 1.6 : 1.0);
                    if (cost < best_cost) { best_cost = cost; best = v; best_sp = sp; }
                    break;
                }
            }
    if (best < 0) return false;
    *sp_out = best_sp;
    return true;
}

// Convolution length and factorisation of a paired chirp-z plan for a length-D complex transform: P = L x nc >= 2 D - 1 with
// nc % 8 == 0 (4-column tiles in mirrored pairs), rows of at most 4096 points (one row per workgroup, stages in place), columns
// of at most 2048; smallest cost, where a plan whose column-tile pair does not fit one in-place call (L > 1024) and a third level
// pay their extra barriers / passes.
static bool pz_length(int64_t D, FlSplit* sp_out) {
    const int64_t want = 2 * D - 1;
    const int64_t limit = want < 64 ? 4096 : want + want / 2;
    if (const char* e = getenv("EGR_PZ_SPLIT")) {            // dev: "L,nc" forces the factorisation
        int L = 0, nc = 0;
        if (sscanf(e, "%d,%d", &L, &nc) == 2 && (int64_t)L * nc >= want && nc % 8 == 0) {
            FlSplit sp = plan_split_explicit(2 * (int64_t)L * nc, L, nc, 1, 4);
            if (sp.ok && sp.TC == 4) { *sp_out = sp; return true; }
        }
    }
    std::vector<int64_t> cand;                    // 13-smooth numbers in [want, limit]
    {
        const int primes[6] = {2, 3, 5, 7, 11, 13};
        std::vector<int64_t> cur{1};
        for (int pi = 0; pi < 6; ++pi) {
            std::vector<int64_t> next;
            for (int64_t v : cur)
                for (int64_t x = v; x <= limit; x *= primes[pi]) next.push_back(x);
            cur.swap(next);
        }
        for (int64_t v : cur) if (v >= want && v >= 8) cand.push_back(v);
        std::sort(cand.begin(), cand.end());
    }
    double best_cost = 1e300;
    FlSplit best;
    best.ok = false;
    for (int64_t v : cand) {
        if ((double)v >= best_cost) break;
        // two levels
        double c2 = 1e300;
        int bl = 0;
        for (int64_t L = 1; L <= 2048 && L <= v; ++L) {
            if (v % L) continue;
            const int64_t nc = v / L;
            if (nc % 8) continue;
            const bool sched = nc <= 16384 && pz_sched_has((int)L, (int)nc);       // compile-time schedules on both passes
            if (nc > 4096 && !sched) continue;
            FftDesc f1, f2;
            if (!make_schedule((int)L, &f1) || !make_schedule((int)nc, &f2)) continue;
            double c = (double)v * (1.0 + 0.02 * (f1.nst + f2.nst));
            if (L > 1024) c *= 1.15;
            if (sched) c *= 0.55;
            if (c < c2) { c2 = c; bl = (int)L; }
        }
        if (bl) {
            if (c2 < best_cost) {
                FlSplit sp = plan_split_explicit(2 * v, bl, (int)(v / bl), 1, 4, 16384);
                if (sp.ok && sp.TC == 4) { best_cost = c2; best = sp; }
            }
            continue;
        }
        if (v <= 1024LL * 16384LL) continue;
        FlSplit sp = plan_split(2 * v, 0, 4, 2048);
        if (!sp.ok || sp.levels != 3 || sp.TC != 4 || ((int64_t)sp.M2 * sp.M3) % 8) continue;
        const double c3 = (double)v * 1.6;
        if (c3 < best_cost) { best_cost = c3; best = sp; }
    }
    if (!best.ok) return false;
    *sp_out = best;
    return true;
}

static int create_for_length(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, int64_t N, int m1_hint, int tc_hint) {
    if (m1_hint <= 0) { if (const char* e = getenv("EGR_FL_M1")) m1_hint = atoi(e); }
    if (tc_hint <= 0) { if (const char* e = getenv("EGR_FL_TC")) tc_hint = atoi(e); }
    FlSplit sp = plan_split(N, m1_hint, tc_hint);
    // the 625-point column schedule is instantiated for one tile width
    if (tc_hint <= 0 && sp.ok && sp.levels == 2 && sp.M1 == 625 && sp.TC != EGR_FL_COL_TC) sp = plan_split(N, m1_hint, EGR_FL_COL_TC);
    if (!sp.ok) {
        const int kind = pz_kind_for(N);
        const int64_t D = kind == 1 ? N / 2 : N;
        if (N >= 2 && !(getenv("EGR_FL_LEGACY_CHIRPZ") && atoi(getenv("EGR_FL_LEGACY_CHIRPZ"))) && pz_length(D, &sp))
            return build_plan(out, n_in, channels, factor, sp, D, kind, N);
        if (N >= 2 && bluestein_length(2 * N - 1, &sp)) return build_plan(out, n_in, channels, factor, sp, N, 0, N);
        set_error(kUnsupported, (long long)N);
        return EGR_ERR_UNSUPPORTED;
    }
    return build_plan(out, n_in, channels, factor, sp, 0, 0, N);
}

extern "C" int egr_fatllama_plan_create(egr_fatllama_plan** out, int64_t n_in, int channels, int factor,
                                        int m1_hint, int tc_hint) {
    EGR_CHECK(out != nullptr, EGR_ERR_ARG, "out is null");
    *out = nullptr;
    EGR_CHECK(n_in >= 1 && factor >= 1 && channels >= 1 && channels <= 64, EGR_ERR_ARG,
              "n_in=%lld channels=%d factor=%d out of range", (long long)n_in, channels, factor);
    return create_for_length(out, n_in, channels, factor, n_in * factor, m1_hint, tc_hint);
}

// A plan whose output length is given explicitly (n_out >= n_in, not necessarily a multiple of it): SPEC.md factor_mode
// "ratio_then_int".  egr_fatllama_enhance on such a plan needs EGR_FL_INTERP_LINSPACE (the only up-rating defined for a ratio).
extern "C" int egr_fatllama_plan_create_n(egr_fatllama_plan** out, int64_t n_in, int64_t n_out, int channels) {
    EGR_CHECK(out != nullptr, EGR_ERR_ARG, "out is null");
    *out = nullptr;
    EGR_CHECK(n_in >= 1 && n_out >= n_in && channels >= 1 && channels <= 64, EGR_ERR_ARG,
              "n_in=%lld n_out=%lld channels=%d out of range", (long long)n_in, (long long)n_out, channels);
    return create_for_length(out, n_in, channels, 0, n_out, 0, 0);
}

// Force a chirp-z plan on any length (tests, A/B runs): kind 1 = paired, even/odd packing (N even); 2 = paired, channel pairs
// (any N); 3 = the legacy full-complex form (one P >= 2N - 1 state per channel).
extern "C" int egr_fatllama_plan_create_chirpz(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, int kind) {
    EGR_CHECK(out != nullptr, EGR_ERR_ARG, "out is null");
    *out = nullptr;
    EGR_CHECK(n_in >= 1 && factor >= 1 && channels >= 1 && channels <= 64 && n_in * factor >= 2, EGR_ERR_ARG,
              "n_in=%lld channels=%d factor=%d out of range", (long long)n_in, channels, factor);
    const int64_t N = n_in * factor;
    if (kind == 0) kind = pz_kind_for(N);
    if (kind == 3) return egr_fatllama_plan_create_bluestein(out, n_in, channels, factor);
    EGR_CHECK(kind == 2 || (kind == 1 && !(N & 1)), EGR_ERR_ARG, "chirp-z kind %d does not fit N=%lld", kind, (long long)N);
    FlSplit sp;
    const int64_t D = kind == 1 ? N / 2 : N;
    if (!pz_length(D, &sp)) {
        set_error("no convolution length found for D=%lld", (long long)D);
        return EGR_ERR_UNSUPPORTED;
    }
    return build_plan(out, n_in, channels, factor, sp, D, kind);
}

extern "C" int egr_fatllama_plan_create_bluestein(egr_fatllama_plan** out, int64_t n_in, int channels, int factor) {
    EGR_CHECK(out != nullptr, EGR_ERR_ARG, "out is null");
    *out = nullptr;
    EGR_CHECK(n_in >= 1 && factor >= 1 && channels >= 1 && channels <= 64 && n_in * factor >= 2, EGR_ERR_ARG,
              "n_in=%lld channels=%d factor=%d out of range", (long long)n_in, channels, factor);
    const int64_t N = n_in * factor;
    FlSplit sp;
    if (!bluestein_length(2 * N - 1, &sp)) {
        set_error("no convolution length found for N=%lld", (long long)N);
        return EGR_ERR_UNSUPPORTED;
    }
    return build_plan(out, n_in, channels, factor, sp, N);
}

extern "C" int egr_fatllama_plan_create_ex(egr_fatllama_plan** out, int64_t n_in, int channels, int factor, int m1,
                                           int m2, int m3, int tc_hint) {
    EGR_CHECK(out != nullptr, EGR_ERR_ARG, "out is null");
    *out = nullptr;
    EGR_CHECK(n_in >= 1 && factor >= 1 && channels >= 1 && channels <= 64, EGR_ERR_ARG,
              "n_in=%lld channels=%d factor=%d out of range", (long long)n_in, channels, factor);
    FlSplit sp = plan_split_explicit(n_in * factor, m1, m2, m3, tc_hint);
    if (!sp.ok) {
        set_error("explicit split %d x %d x %d does not fit N=%lld", m1, m2, m3, (long long)(n_in * factor));
        return EGR_ERR_UNSUPPORTED;
    }
    return build_plan(out, n_in, channels, factor, sp);
}

// dev: one k_row and one k_col<1> launch over the plan's state with per-workgroup phase stamps (100 MHz wall clock);
// prints min / median / max of each phase in microseconds and the launch skew.  Not part of the public header.
extern "C" int egr_fatllama_trace_once(egr_fatllama_plan* p, void* stream) {
    EGR_CHECK(p && !p->bluestein, EGR_ERR_ARG, "packed plan wanted");
    hipStream_t st = (hipStream_t)stream;
    const int C = p->C;
    const long long M = p->sp.M, N = p->sp.N;
    ColP A = p->colA; RowP R = p->row;
    R.thr2 = 0.36f; R.thr = 0.6f;
    const dim3 gA(8 * A.tiles_per_xcd, C), grow(R.R / 2 + 1, C), blk(p->threads);
    const size_t nblk = (size_t)std::max(gA.x * gA.y, grow.x * grow.y);
    long long* tr = nullptr;
    EGR_HIP(hipMalloc((void**)&tr, nblk * 8 * sizeof(long long)));
    std::vector<long long> h(nblk * 8);
    for (int which = 0; which < 2; ++which) {
        EGR_HIP(hipMemsetAsync(tr, 0, nblk * 8 * sizeof(long long), st));
        R.trace = tr; A.trace = tr;
        for (int rep = 0; rep < 3; ++rep) {          // the last repetition's stamps survive
            if (which == 0) {
                if (p->wl_row) {
                    const WlRowEntry* e = (const WlRowEntry*)p->wl_row_entry;
                    hipLaunchKernelGGL(e->fn, grow, dim3(e->threads), EGR_LDS(e->lds), st, R, p->wl_rt, M, p->d_work);
                }
                else if (p->row_sched == 1) hipLaunchKernelGGL((k_row<false, 1>), grow, blk, EGR_LDS((size_t)2 * (R.L + (EGR_FL_ROW_PAD ? R.L >> EGR_FL_ROW_PAD : 0)) * sizeof(cplx)), st, R, M, p->d_work);
                else hipLaunchKernelGGL(k_row<false>, grow, blk, EGR_LDS(p->sp.lds_row), st, R, M, p->d_work);
            } else {
                if (p->wl_col) {
                    wl_launch_col(A, p->wl_ct, M, p->d_work, C, st);
                } else if (p->col_sched == 2) hipLaunchKernelGGL((k_col<1, 2>), gA, dim3(EGR_FL_COL_THREADS), EGR_LDS(p->sp.lds_col / 2), st, A, M, N, 0.6f, p->d_work, (float*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr);
                else hipLaunchKernelGGL(k_col<1>, gA, blk, EGR_LDS(p->sp.lds_col), st, A, M, N, 0.6f, p->d_work, (float*)nullptr, (unsigned*)nullptr);
            }
        }
        EGR_HIP(hipStreamSynchronize(st));
        EGR_HIP(hipMemcpy(h.data(), tr, nblk * 8 * sizeof(long long), hipMemcpyDeviceToHost));
        const size_t nb = which == 0 ? (size_t)grow.x * grow.y : (size_t)gA.x * gA.y;
        const int nph = which == 0 ? 5 : 4;
        long long t0 = -1, t1 = 0;
        std::vector<std::vector<double>> ph(nph);
        std::vector<double> starts;
        for (size_t b = 0; b < nb; ++b) {
            const long long* s = &h[b * 8];
            if (s[0] == 0) continue;
            if (t0 < 0 || s[0] < t0) t0 = s[0];
            if (s[nph] > t1) t1 = s[nph];
            for (int i = 0; i < nph; ++i) ph[i].push_back((s[i + 1] - s[i]) * 0.01);
            starts.push_back((double)s[0]);
        }
        printf("%s: %zu workgroups stamped, first start -> last end %.2f us\n", which == 0 ? "k_row" : "k_col<1>", starts.size(), (t1 - t0) * 0.01);
        std::sort(starts.begin(), starts.end());
        if (!starts.empty()) printf("   start skew: median %.2f us, max %.2f us after the first\n", (starts[starts.size() / 2] - starts[0]) * 0.01, (starts.back() - starts[0]) * 0.01);
        static const char* rn[5] = {"load", "fwd fft", "hook", "inv fft", "store"};
        static const char* cn[4] = {"load+tw", "inv fft", "fwd fft", "tw+store"};
        for (int i = 0; i < nph; ++i) {
            auto& v = ph[i];
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("   %-9s min %.2f  median %.2f  max %.2f us\n", which == 0 ? rn[i] : cn[i], v[0], v[v.size() / 2], v.back());
        }
    }
    hipFree(tr);
    return EGR_OK;
}

extern "C" int egr_fatllama_set_profiling(egr_fatllama_plan* p, int enable) {
    EGR_CHECK(p != nullptr, EGR_ERR_ARG, "plan is null");
    p->profiling = enable != 0;
    return EGR_OK;
}

void fl_prof_begin(egr_fatllama_plan* p, int kind, hipStream_t st, size_t* slot) {
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
void fl_prof_end(egr_fatllama_plan* p, hipStream_t st, size_t* slot) {
    if (!p->profiling) return;
    hipEventRecord(p->ev[*slot + 1], st);
    *slot += 2;
}

extern "C" int egr_fatllama_enhance(egr_fatllama_plan* p, const float* x, float* out, int max_iter, float thr,
                                    unsigned flags, void* stream) {
    EGR_CHECK(p && x && out, EGR_ERR_ARG, "null plan/x/out");
    EGR_CHECK(max_iter >= 0, EGR_ERR_ARG, "max_iter=%d < 0", max_iter);
    hipStream_t st = (hipStream_t)stream;
    const int C = p->C;
    const long long M = p->sp.M, N = p->sp.N;
    const bool three = p->sp.levels == 3;
    ColP A = p->colA, B = p->colB;
    RowP R = p->row;
    R.thr2 = thr * thr;
    R.thr = thr;
    R.soft = (flags & EGR_FL_THR_SOFT) ? 1 : 0;
    const bool relative = (flags & EGR_FL_THR_RELATIVE) != 0;
    const bool no_init = (flags & EGR_FL_NO_INIT_THR) != 0;
    unsigned* peak_in = p->d_peaks;
    unsigned* peak_out = p->d_peaks + C;
    unsigned* peak_y = p->d_peaks + 2 * C;
    EGR_HIP(hipMemsetAsync(p->d_peaks, 0, 3 * C * sizeof(unsigned), st));
    // EGR_FL_THR_RECOMPUTE: the maximum of every iteration's spectrum from a read-only pass of its own (rounds 1-5) instead of the
    // one the previous iteration's hook carried forward; the two agree to the round-off of one float32 transform pair
    const bool recompute = relative && (flags & EGR_FL_THR_RECOMPUTE) != 0;
    if (relative && max_iter > 0) {
        // a ring of EGR_FL_MAX_RING slots (fl_max2_*): iteration it reads slot it, leaves the next maximum in slot it + 1 and clears
        // slot it + 2 (mod ring).  The legacy chirp-z loop keeps one slot per iteration.
        const size_t nslots = (p->bluestein && !p->pz) ? (size_t)max_iter : (size_t)EGR_FL_MAX_RING;
        const size_t need = nslots * C * EGR_FL_MAX_STRIDE;
        if (need > p->max2_cap) {
            if (p->gexec) { EGR_HIP(hipDeviceSynchronize()); EGR_HIP(hipGraphExecDestroy(p->gexec)); p->gexec = nullptr; }      // it recorded the old ring
            if (p->d_max2) { EGR_HIP(hipFree(p->d_max2)); p->d_max2 = nullptr; p->max2_cap = 0; }
            EGR_HIP(hipMalloc((void**)&p->d_max2, need * sizeof(unsigned)));
            p->max2_cap = need;
        }
        EGR_HIP(hipMemsetAsync(p->d_max2, 0, need * sizeof(unsigned), st));
    }
    auto max2_slot = [&](int it, int ch0) { return p->d_max2 + ((size_t)(it % EGR_FL_MAX_RING) * C + ch0) * EGR_FL_MAX_STRIDE; };
    // time-domain level of the opening pass: thr, thr * max|y| (relative) or none (every sample kept)
    const float thr0 = no_init ? -1.0f : thr;
    const unsigned* thr0_rel = (relative && !no_init) ? peak_y : nullptr;
    {
        const int nb = (int)((p->n_in + 255) / 256 < 2048 ? (p->n_in + 255) / 256 : 2048);
        const int lin = (flags & EGR_FL_INTERP_LINSPACE) ? 1 : 0;
        EGR_CHECK(lin || p->factor >= 1, EGR_ERR_ARG, "a plan with an explicit output length (egr_fatllama_plan_create_n) needs EGR_FL_INTERP_LINSPACE");
        EGR_CHECK(!(lin && (flags & EGR_FL_ZERO_STUFF)), EGR_ERR_ARG, "EGR_FL_INTERP_LINSPACE and EGR_FL_ZERO_STUFF exclude each other");
        hipLaunchKernelGGL(k_prepare, dim3(nb, C), dim3(256), 0, st, x, out, (long long)p->n_in, p->factor > 0 ? p->factor : 1,
                           (flags & EGR_FL_PCM_IN) ? 1 : 0, (flags & EGR_FL_ZERO_STUFF) ? 1 : 0, peak_in, peak_y, lin, (long long)p->n_out);
    }
    const dim3 gA(8 * A.tiles_per_xcd, C), gB(8 * B.tiles_per_xcd, C * (three ? B.nplanes : 1)), grow(R.R / 2 + 1, C),
        blk(p->bluestein ? blue_threads() : p->threads), blk256(256);
    const size_t lc = EGR_LDS(p->sp.lds_col), lb = EGR_LDS(p->sp.lds_colb), lr = EGR_LDS(p->sp.lds_row);
    size_t slot = 0;
    if (max_iter == 0) {
        const long long Nr = (long long)p->n_out;
        const int nb = (int)((Nr + 255) / 256 < 2048 ? (Nr + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_noiter, dim3(nb, C), blk256, 0, st, out, Nr, thr0, peak_out, thr0_rel);
    } else if (p->pz) {
        const int rc = pz_loop(p, out, max_iter, thr, thr0, thr0_rel, flags, peak_out, st);
        if (rc) return rc;
    } else if (p->bluestein) {
        ChirpP cp = p->chirp;
        cp.soft = R.soft;
        const long long P = M;
        const dim3 grc((R.R + EGR_FL_CONV_ROWS - 1) / EGR_FL_CONV_ROWS, C);
        const float thr2 = thr * thr;
        auto conv = [&]() {
            if (three) hipLaunchKernelGGL(k_col<4>, gB, blk, lb, st, B, P, N, thr, p->d_work, out, peak_out);
            hipLaunchKernelGGL(k_rowconv<true>, grc, dim3(EGR_FL_CONV_THREADS), rowconv_lds(R.L), st, R.f, R.L, R.R, R.tw, (const cplx*)p->d_bhat, 1.0f, P,
                               p->d_work);
            if (three) hipLaunchKernelGGL(k_col<3>, gB, blk, lb, st, B, P, N, thr, p->d_work, out, peak_out);
        };
        hipLaunchKernelGGL((k_colz<0, 0>), gA, blk, lc, st, A, cp, P, thr0, thr2, p->d_work, out, peak_out, thr0_rel);
        for (int it = 0; it < max_iter; ++it) {
            conv();
            if (relative) {                 // this iteration's spectrum maximum first (same pass, no write-back)
                cp.max2_out = p->d_max2 + (size_t)it * C * EGR_FL_MAX_STRIDE;
                hipLaunchKernelGGL((k_colz<3, 1>), gA, blk, lc, st, A, cp, P, thr, thr2, p->d_work, out, peak_out, (const unsigned*)nullptr);
                cp.max2 = cp.max2_out;
            }
            hipLaunchKernelGGL((k_colz<1, 1>), gA, blk, lc, st, A, cp, P, thr, thr2, p->d_work, out, peak_out);
            conv();
            if (it + 1 < max_iter)
                hipLaunchKernelGGL((k_colz<1, 2>), gA, blk, lc, st, A, cp, P, thr, thr2, p->d_work, out, peak_out);
        }
        hipLaunchKernelGGL((k_colz<2, 0>), gA, blk, lc, st, A, cp, P, thr, thr2, p->d_work, out, peak_out);
    } else {
        // The channels are independent until k_finalize.  Each loop kernel alone fills barely more than one wave of
        // workgroups, so two channel groups run as concurrent pipelines on two streams (fork/join by events) and one
        // group's k_row overlaps the other's k_col.
        const int ngroups = (p->nstreams == 2 && C >= 2) ? 2 : 1;
        // the in-place schedules need one thread per butterfly (288 / 512 row, 200 column butterflies per stage): 512 threads
        const bool sched_ok = p->threads == 512;
        const dim3 blkc(EGR_FL_COL_THREADS);      // the scheduled column kernels' own workgroup size
        const bool rs1 = p->row_sched == 1 && sched_ok, cs2 = p->col_sched == 2 && sched_ok && !three;
        // the two-barrier kernels serve the default hook (hard threshold against an absolute level) and the middle column pass
        const WlRowEntry* wle = (const WlRowEntry*)p->wl_row_entry;
        const bool wl_variant = relative || R.soft != 0;          // k_row_wl<.., 1>: level from the iteration's maximum and / or soft shrink
        const bool wlr = p->wl_row != 0 && wle, wlc = p->wl_col != 0;
        // no ping-pong buffer; the row kernel's two rows are padded by one element per 2^EGR_FL_ROW_PAD
        const size_t lrs = EGR_LDS((size_t)2 * (R.L + (EGR_FL_ROW_PAD ? R.L >> EGR_FL_ROW_PAD : 0)) * sizeof(cplx));
        const size_t lcs = EGR_LDS(p->sp.lds_col / 2);
        if (ngroups == 2 && !p->side) {
            EGR_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
            p->side_owned = 1;
        }
        if (ngroups == 2 && !p->ev_fork) {
            EGR_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
            EGR_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
        }
        // One group's launches for iterations [it0, it1); `first` adds the opening time-domain pass, `last` the closing one.
        auto run_group = [&](hipStream_t s0, int g, int it0, int it1, bool first, bool last, bool prof) {
            const int c0 = g == 0 ? 0 : C / 2, cn = ngroups == 1 ? C : (g == 0 ? C / 2 : C - C / 2);
            hipStream_t sg = g == 0 ? s0 : p->side;
            cplx* wk = p->d_work + (size_t)c0 * M;
            float* og = out + (size_t)c0 * N;
            unsigned* pk = peak_out + c0;
            const dim3 gAg(gA.x, cn), gBg(gB.x, cn * (three ? B.nplanes : 1)), growg(grow.x, cn);
            if (first) {
                if (cs2) hipLaunchKernelGGL((k_col<0, 2>), gAg, blkc, lcs, sg, A, M, N, thr0, wk, og, pk, thr0_rel ? thr0_rel + c0 : nullptr);
                else hipLaunchKernelGGL(k_col<0>, gAg, blk, lc, sg, A, M, N, thr0, wk, og, pk, thr0_rel ? thr0_rel + c0 : nullptr);
                if (three) {
                    if (p->wl_inner) wl_launch_inner(p->wl_colB, p->wl_it, true, M, wk, cn, sg);
                    else hipLaunchKernelGGL(k_col<4>, gBg, blk, lb, sg, B, M, N, thr, wk, og, pk);
                }
            }
            // the spectrum maximum by a pass of its own (forward row transforms + split, no write-back): only the FIRST iteration
            // needs it -- d0 = T0(y) is not the image of a shrunk spectrum -- every later one finds the maximum its predecessor's hook left
            auto max_pass = [&](int it) {
                RowP Rm = R;
                Rm.max2_out = max2_slot(it, c0);
                if (wlr) hipLaunchKernelGGL(wle->fn_max, growg, dim3(wle->threads), EGR_LDS(wle->lds), sg, Rm, p->wl_rt, M, wk);
                else if (rs1) hipLaunchKernelGGL((k_row<true, 1>), growg, blk, lrs, sg, Rm, M, wk);
                else hipLaunchKernelGGL(k_row<true>, growg, blk, lr, sg, Rm, M, wk);
            };
            if (first && relative) max_pass(0);
            for (int it = it0; it < it1; ++it) {
                RowP Rg = R;
                if (relative) {
                    if (recompute && it > 0) max_pass(it);
                    Rg.max2 = max2_slot(it, c0);
                    Rg.max2_next = recompute ? nullptr : max2_slot(it + 1, c0);
                    Rg.max2_zero = max2_slot(it + 2, c0);
                }
                if (prof) fl_prof_begin(p, 0, sg, &slot);
                if (wlr && wl_variant) hipLaunchKernelGGL(wle->fn_variant, growg, dim3(wle->threads), EGR_LDS(wle->lds), sg, Rg, p->wl_rt, M, wk);
                else if (wlr) hipLaunchKernelGGL(wle->fn, growg, dim3(wle->threads), EGR_LDS(wle->lds), sg, Rg, p->wl_rt, M, wk);
                else if (rs1) hipLaunchKernelGGL((k_row<false, 1>), growg, blk, lrs, sg, Rg, M, wk);
                else hipLaunchKernelGGL(k_row<false>, growg, blk, lr, sg, Rg, M, wk);
                if (prof) fl_prof_end(p, sg, &slot);
                if (three) {
                    if (prof) fl_prof_begin(p, 2, sg, &slot);
                    if (p->wl_inner) wl_launch_inner(p->wl_colB, p->wl_it, false, M, wk, cn, sg);
                    else hipLaunchKernelGGL(k_col<3>, gBg, blk, lb, sg, B, M, N, thr, wk, og, pk);
                    if (prof) fl_prof_end(p, sg, &slot);
                }
                if (it + 1 < max_iter) {
                    if (prof) fl_prof_begin(p, 1, sg, &slot);
                    if (wlc) wl_launch_col(A, p->wl_ct, M, wk, cn, sg);
                    else if (cs2) hipLaunchKernelGGL((k_col<1, 2>), gAg, blkc, lcs, sg, A, M, N, thr, wk, og, pk, (const unsigned*)nullptr);
                    else hipLaunchKernelGGL(k_col<1>, gAg, blk, lc, sg, A, M, N, thr, wk, og, pk);
                    if (prof) fl_prof_end(p, sg, &slot);
                    if (three) {
                        if (prof) fl_prof_begin(p, 2, sg, &slot);
                        if (p->wl_inner) wl_launch_inner(p->wl_colB, p->wl_it, true, M, wk, cn, sg);
                        else hipLaunchKernelGGL(k_col<4>, gBg, blk, lb, sg, B, M, N, thr, wk, og, pk);
                        if (prof) fl_prof_end(p, sg, &slot);
                    }
                }
            }
            if (last) {
                if (cs2) hipLaunchKernelGGL((k_col<2, 2>), gAg, blkc, lcs, sg, A, M, N, thr, wk, og, pk, (const unsigned*)nullptr);
                else hipLaunchKernelGGL(k_col<2>, gAg, blk, lc, sg, A, M, N, thr, wk, og, pk);
            }
        };
        auto fork = [&](hipStream_t s0) -> int {
            if (ngroups == 2) {
                EGR_HIP(hipEventRecord(p->ev_fork, s0));
                EGR_HIP(hipStreamWaitEvent(p->side, p->ev_fork, 0));
            }
            return EGR_OK;
        };
        auto join = [&](hipStream_t s0) -> int {
            if (ngroups == 2) {
                EGR_HIP(hipEventRecord(p->ev_join, p->side));
                EGR_HIP(hipStreamWaitEvent(s0, p->ev_join, 0));
            }
            return EGR_OK;
        };
        // The loop body is the same pair of launches every iteration: CH iterations of both pipelines are captured once
        // into a hipGraph and replayed (inter-kernel gaps of ~8 us on the streams shrink to the graph's ~1 us); the
        // executable graph is kept while (out, threshold, geometry) stay the same.  Profiling runs use plain launches.
        constexpr int CH = 25;
        const bool profiling = p->profiling != 0;
        static_assert(CH == EGR_FL_MAX_RING, "the captured iterations address the ring of maxima by iteration mod CH");
        const int n_graph = (!profiling && p->use_graph && !recompute && max_iter > 2 * CH) ? (max_iter - 1) / CH : 0;
        const int g_kind = R.soft | (relative ? 2 : 0);
        int rc = fork(st);
        if (rc) return rc;
        for (int g = 0; g < ngroups; ++g) run_group(st, g, 0, 0, true, false, false);
        if (n_graph > 0) {
            rc = join(st);
            if (rc) return rc;
            // The captured middle iterations touch the plan's own state only (the row pass and the middle column pass never see
            // `out`), so the executable graph is keyed by (threshold, pipelines, hook kind) and survives calls with other buffers.
            // An executable graph is never destroyed while a launch of it may still be in flight (calls return asynchronously):
            // the device is drained first -- a re-capture is a rare, millisecond-scale event anyway.
            if (!(p->gexec && p->g_thr == thr && p->g_groups == ngroups && p->g_iter_odd == g_kind)) {
                if (p->gexec) { EGR_HIP(hipDeviceSynchronize()); EGR_HIP(hipGraphExecDestroy(p->gexec)); p->gexec = nullptr; }
                hipGraph_t graph = nullptr;
                // captured on a private stream (the caller's may be the legacy default stream, which cannot capture)
                if (!p->cap) EGR_HIP(hipStreamCreateWithFlags(&p->cap, hipStreamNonBlocking));
                EGR_HIP(hipStreamBeginCapture(p->cap, hipStreamCaptureModeThreadLocal));
                rc = fork(p->cap);
                // iterations 0..CH-1 of a run with more than CH + 1 iterations: every one is a "middle" iteration
                if (!rc) for (int g = 0; g < ngroups; ++g) run_group(p->cap, g, 0, CH, false, false, false);
                if (!rc) rc = join(p->cap);
                hipError_t ce = hipStreamEndCapture(p->cap, &graph);
                if (rc || ce != hipSuccess) {          // leave no half-captured state behind: the next call starts from scratch
                    if (graph) hipGraphDestroy(graph);
                    hipStreamDestroy(p->cap);
                    p->cap = nullptr;
                    if (rc) return rc;
                    EGR_HIP(ce);
                }
                hipError_t ie = hipGraphInstantiate(&p->gexec, graph, nullptr, nullptr, 0);
                hipGraphDestroy(graph);
                if (ie != hipSuccess) { p->gexec = nullptr; EGR_HIP(ie); }
                p->g_out = out; p->g_thr = thr; p->g_groups = ngroups; p->g_iter_odd = g_kind;
            }
            for (int i = 0; i < n_graph; ++i) EGR_HIP(hipGraphLaunch(p->gexec, st));
            rc = fork(st);
            if (rc) return rc;
        }
        for (int g = 0; g < ngroups; ++g) run_group(st, g, n_graph * CH, max_iter, false, true, profiling);
        rc = join(st);
        if (rc) return rc;
    }
    if ((flags & (EGR_FL_NORMALIZE | EGR_FL_AUTOSCALE | EGR_FL_NODE_POST)) && !(flags & EGR_FL_DEFER_FINALIZE)) {
        const long long Nr = (long long)p->n_out;
        const int nb = (int)((Nr + 255) / 256 < 2048 ? (Nr + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_finalize, dim3(nb, C), blk256, 0, st, out, Nr, C, flags, peak_in, peak_out, (const float*)nullptr, (float*)nullptr);
    }
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// Channel-parallel runs (the Fat-Llama path shards over channels only; C = 2 => at most 2 GPUs, nothing is
// exchanged until the joint normalise): every rank runs egr_fatllama_enhance on ITS channels with EGR_FL_DEFER_FINALIZE, asks for its
// joint peak (a device float), the host all-reduces that one float with MAX, and egr_fatllama_finalize applies autoscale / normalise /
// write patch / PCM_16 with the all-reduced value.  `flags` as given to enhance.
extern "C" int egr_fatllama_joint_peak(egr_fatllama_plan* p, unsigned flags, float* joint_dev, void* stream) {
    EGR_CHECK(p && joint_dev, EGR_ERR_ARG, "null plan / joint_dev");
    hipLaunchKernelGGL(k_finalize, dim3(1, 1), dim3(64), 0, (hipStream_t)stream, (float*)nullptr, 0LL, p->C, flags, p->d_peaks, p->d_peaks + p->C,
                       (const float*)nullptr, joint_dev);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

extern "C" int egr_fatllama_finalize(egr_fatllama_plan* p, float* out, unsigned flags, const float* joint_dev, void* stream) {
    EGR_CHECK(p && out, EGR_ERR_ARG, "null plan / out");
    if (!(flags & (EGR_FL_NORMALIZE | EGR_FL_AUTOSCALE | EGR_FL_NODE_POST))) return EGR_OK;
    const long long Nr = (long long)p->n_out;
    const int nb = (int)((Nr + 255) / 256 < 2048 ? (Nr + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_finalize, dim3(nb, p->C), dim3(256), 0, (hipStream_t)stream, out, Nr, p->C, flags, p->d_peaks, p->d_peaks + p->C, joint_dev, (float*)nullptr);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// The second pipeline's stream.  HIP multiplexes streams onto a few hardware queues; a side stream that lands on the caller's
// queue serialises the two channel pipelines, so the host may hand in one it has verified (egr_streams_overlap_us).  The plan
// does not own it.  Drops a captured loop graph (it recorded the previous stream).
extern "C" int egr_fatllama_set_side_stream(egr_fatllama_plan* p, void* stream) {
    EGR_CHECK(p && stream, EGR_ERR_ARG, "null plan / stream");
    if (p->side == (hipStream_t)stream) return EGR_OK;
    if (p->side && p->side_owned) EGR_HIP(hipStreamDestroy(p->side));
    p->side = (hipStream_t)stream;
    p->side_owned = 0;
    if (p->gexec) { EGR_HIP(hipDeviceSynchronize()); EGR_HIP(hipGraphExecDestroy(p->gexec)); p->gexec = nullptr; }
    return EGR_OK;
}

// Replay of the loop from a captured hipGraph on / off (default on unless EGR_FL_GRAPH=0).  The two ways give identical bits;
// which is faster depends on how the runtime maps the graph's branches onto hardware queues, so the host may time both.
extern "C" int egr_fatllama_set_graph(egr_fatllama_plan* p, int enable) {
    EGR_CHECK(p != nullptr, EGR_ERR_ARG, "plan is null");
    p->use_graph = enable ? 1 : 0;
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
    double sum[3] = {0, 0, 0};
    int64_t cnt[3] = {0, 0, 0};
    EGR_HIP(hipDeviceSynchronize());
    for (size_t i = 0; i + 1 < p->ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]) != hipSuccess) continue;
        const int k = p->ev_kind[i / 2];
        sum[k] += ms;
        cnt[k] += 1;
    }
    // the inner column pass (3-level plans) is reported together with the outer one
    if (row_ms_avg) *row_ms_avg = cnt[0] ? sum[0] / cnt[0] : 0.0;
    if (col_ms_avg) *col_ms_avg = (cnt[1] + cnt[2]) ? (sum[1] + sum[2]) / (cnt[1] + cnt[2]) : 0.0;
    if (row_launches) *row_launches = cnt[0];
    if (col_launches) *col_launches = cnt[1] + cnt[2];
    return EGR_OK;
}

extern "C" int egr_fatllama_kernel_times3(egr_fatllama_plan* p, double ms_avg[3], int64_t launches[3]) {
    EGR_CHECK(p && ms_avg && launches, EGR_ERR_ARG, "null argument");
    double sum[3] = {0, 0, 0};
    int64_t cnt[3] = {0, 0, 0};
    EGR_HIP(hipDeviceSynchronize());
    for (size_t i = 0; i + 1 < p->ev.size() && i / 2 < p->ev_kind.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]) != hipSuccess) continue;
        const int k = p->ev_kind[i / 2];
        if (k < 0 || k > 2) continue;
        sum[k] += ms;
        cnt[k] += 1;
    }
    for (int k = 0; k < 3; ++k) { ms_avg[k] = cnt[k] ? sum[k] / cnt[k] : 0.0; launches[k] = cnt[k]; }
    return EGR_OK;
}

// inner column pass of a three-level plan over `nstates` consecutive states (used by the chirp-z loops around their row pass)
void fl_launch_inner(egr_fatllama_plan* p, bool forward, cplx* work, int nstates, hipStream_t st) {
    const ColP& B = p->colB;
    const dim3 gB(8 * B.tiles_per_xcd, nstates * B.nplanes), blk(1024);
    const long long M = p->sp.M, N = p->sp.N;
    if (forward) hipLaunchKernelGGL(k_col<4>, gB, blk, EGR_LDS(p->sp.lds_colb), st, B, M, N, 0.f, work, (float*)nullptr, (unsigned*)nullptr);
    else hipLaunchKernelGGL(k_col<3>, gB, blk, EGR_LDS(p->sp.lds_colb), st, B, M, N, 0.f, work, (float*)nullptr, (unsigned*)nullptr);
}

// y = irfft(rfft(x) * gain): one forward transform, a real per-bin gain, one inverse, on the plan's passes.
extern "C" int egr_spectral_gain(egr_fatllama_plan* p, const float* x, const float* gain, float* y, void* stream) {
    EGR_CHECK(p && x && gain && y, EGR_ERR_ARG, "null argument");
    EGR_CHECK(!p->bluestein && p->factor == 1, EGR_ERR_UNSUPPORTED, "spectral gain needs a packed-real plan with factor 1");
    hipStream_t st = (hipStream_t)stream;
    const int C = p->C;
    const long long M = p->sp.M, N = p->sp.N;
    const bool three = p->sp.levels == 3;
    ColP A = p->colA, B = p->colB;
    RowP R = p->row;
    R.gain = gain;
    R.phat = 0;
    const dim3 gA(8 * A.tiles_per_xcd, C), gB(8 * B.tiles_per_xcd, C * (three ? B.nplanes : 1)), grow(R.R / 2 + 1, C), blk(256);
    const size_t lc = EGR_LDS(p->sp.lds_col), lb = EGR_LDS(p->sp.lds_colb), lr = EGR_LDS(p->sp.lds_row);
    hipLaunchKernelGGL(k_col<0>, gA, blk, lc, st, A, M, N, -1.0f, p->d_work, const_cast<float*>(x), (unsigned*)nullptr);
    if (three) hipLaunchKernelGGL(k_col<4>, gB, blk, lb, st, B, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
    hipLaunchKernelGGL(k_row<false>, grow, blk, lr, st, R, M, p->d_work);
    if (three) hipLaunchKernelGGL(k_col<3>, gB, blk, lb, st, B, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
    hipLaunchKernelGGL(k_col<5>, gA, blk, lc, st, A, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// y = irfft(rfft(x) * [k >= band_lo]) per channel: the brick-wall high band of x at ITS OWN length (any plan kind, factor 1).
// Serves the null-test suite's _band_energy_hi_db (egregora_null_test_suite.py:192-199): by Parseval the band energies of the
// length-N rfft follow from time-domain sums over x and y (egr_band_sums).
extern "C" int egr_band_filter(egr_fatllama_plan* p, const float* x, int64_t band_lo, float* y, void* stream) {
    EGR_CHECK(p && x && y && band_lo >= 0, EGR_ERR_ARG, "bad argument");
    EGR_CHECK(p->factor == 1, EGR_ERR_UNSUPPORTED, "band filter needs a plan with factor 1");
    hipStream_t st = (hipStream_t)stream;
    if (p->pz) return pz_band_filter(p, x, band_lo, y, st);
    const int C = p->C;
    const long long M = p->sp.M, N = p->sp.N;
    const bool three = p->sp.levels == 3;
    ColP A = p->colA, B = p->colB;
    RowP R = p->row;
    R.gain = nullptr; R.phat = 0; R.band = 1; R.band_lo = band_lo;
    const dim3 gA(8 * A.tiles_per_xcd, C), gB(8 * B.tiles_per_xcd, C * (three ? B.nplanes : 1)), blk(256);
    const size_t lc = EGR_LDS(p->sp.lds_col), lb = EGR_LDS(p->sp.lds_colb), lr = EGR_LDS(p->sp.lds_row);
    float* xs = const_cast<float*>(x);          // read only (MODE 0 passes)
    if (p->bluestein) {
        ChirpP cp = p->chirp;
        cp.band = 1; cp.band_lo = (unsigned long long)band_lo;
        const long long P = M;
        const dim3 grc((R.R + EGR_FL_CONV_ROWS - 1) / EGR_FL_CONV_ROWS, C);
        unsigned* pk = p->d_peaks + C;
        auto conv = [&]() {
            if (three) hipLaunchKernelGGL(k_col<4>, gB, blk, lb, st, B, P, N, 0.f, p->d_work, y, pk);
            hipLaunchKernelGGL(k_rowconv<true>, grc, dim3(EGR_FL_CONV_THREADS), rowconv_lds(R.L), st, R.f, R.L, R.R, R.tw, (const cplx*)p->d_bhat, 1.0f, P, p->d_work);
            if (three) hipLaunchKernelGGL(k_col<3>, gB, blk, lb, st, B, P, N, 0.f, p->d_work, y, pk);
        };
        hipLaunchKernelGGL((k_colz<0, 0>), gA, blk, lc, st, A, cp, P, -1.0f, 0.f, p->d_work, xs, pk);
        conv();
        hipLaunchKernelGGL((k_colz<1, 1>), gA, blk, lc, st, A, cp, P, 0.f, 0.f, p->d_work, y, pk);
        conv();
        EGR_HIP(hipMemsetAsync(y, 0, (size_t)C * cp.N * sizeof(float), st));      // the closing pass adds d to what y holds
        hipLaunchKernelGGL((k_colz<2, 0>), gA, blk, lc, st, A, cp, P, 0.f, 0.f, p->d_work, y, pk);
    } else {
        const dim3 grow(R.R / 2 + 1, C);
        hipLaunchKernelGGL(k_col<0>, gA, blk, lc, st, A, M, N, -1.0f, p->d_work, xs, (unsigned*)nullptr);
        if (three) hipLaunchKernelGGL(k_col<4>, gB, blk, lb, st, B, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_row<false>, grow, blk, lr, st, R, M, p->d_work);
        if (three) hipLaunchKernelGGL(k_col<3>, gB, blk, lb, st, B, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_col<5>, gA, blk, lc, st, A, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
    }
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// GCC-PHAT delay estimate on the same transform passes (reference _xcorr_delay, egregora_null_test_suite.py:213-237):
// z = a + i b (zero-padded to n = 2^k >= len(a) + len(b)) is the packed state of a plan for N = 2n "real" samples; the row
// pass's hook turns it into the PHAT-weighted cross-spectrum, the inverse passes return the correlation in the real parts.
namespace egr {

__global__ __launch_bounds__(256) void k_phat_pack(const float* __restrict__ a, long long na, const float* __restrict__ b,
                                                    long long nb, long long n, float2* __restrict__ z) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        z[i] = make_float2(i < na ? a[i] : 0.f, i < nb ? b[i] : 0.f);
}

// cc_c = concat(cc[-(n/2-1):], cc[:n/2+1]), centre = n/2: cc_c[j] = cc[(j + n/2 + 1) mod n].  First maximum of
// cc_c[centre - ms .. centre + ms]; out = {index (int bits), cc_c[idx-1], cc_c[idx], cc_c[idx+1]}.  ONE workgroup.
__global__ __launch_bounds__(1024) void k_phat_peak(const float2* __restrict__ y, long long n, long long ms,
                                                     float* __restrict__ out4) {
    __shared__ float bv[1024];
    __shared__ long long bi[1024];
    const long long centre = n / 2, sl = centre - ms, cnt = 2 * ms + 1;
    float best = -INFINITY;
    long long besti = sl;
    for (long long t = threadIdx.x; t < cnt; t += 1024) {
        const long long j = sl + t;
        const float v = y[(j + n / 2 + 1) % n].x;
        if (v > best) { best = v; besti = j; }          // ascending t per thread: keeps the first maximum
    }
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = besti;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v2 = bv[threadIdx.x + o];
            const long long i2 = bi[threadIdx.x + o];
            if (v2 > bv[threadIdx.x] || (v2 == bv[threadIdx.x] && i2 < bi[threadIdx.x])) { bv[threadIdx.x] = v2; bi[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const long long idx = bi[0];
        out4[0] = __int_as_float((int)(idx - centre));
        const bool inner = idx >= 1 && idx < n - 1;
        out4[1] = inner ? y[(idx - 1 + n / 2 + 1) % n].x : 0.f;
        out4[2] = y[(idx + n / 2 + 1) % n].x;
        out4[3] = inner ? y[(idx + 1 + n / 2 + 1) % n].x : 0.f;
    }
}

}  // namespace egr

using namespace egr;

// plan: egr_fatllama_plan_create(&plan, 2 n, 1, 1, ...) with n = 2^k >= na + nb; work: 4 n floats of device memory
// (z and the correlation, interleaved pairs); out4 (device, 4 floats): {peak index - n/2 as int bits, y0, y1, y2}.
extern "C" int egr_gcc_phat(egr_fatllama_plan* p, const float* a, int64_t na, const float* b, int64_t nb, int64_t max_shift,
                            float* work, float* out4, void* stream) {
    EGR_CHECK(p && a && b && work && out4 && na >= 1 && nb >= 1, EGR_ERR_ARG, "null / empty argument");
    EGR_CHECK(!p->bluestein && p->factor == 1 && p->C == 1, EGR_ERR_UNSUPPORTED, "GCC-PHAT needs a one-channel packed-real plan");
    const long long M = p->sp.M, N = p->sp.N, n = M;
    EGR_CHECK((n & (n - 1)) == 0 && n >= na + nb, EGR_ERR_ARG, "plan length must be 2 n with n = 2^k >= na + nb");
    EGR_CHECK(max_shift >= 0 && max_shift < n / 2 - 1, EGR_ERR_ARG, "max_shift out of range");
    hipStream_t st = (hipStream_t)stream;
    const bool three = p->sp.levels == 3;
    ColP A = p->colA, B = p->colB;
    RowP R = p->row;
    R.gain = nullptr;
    R.phat = 1;
    float* z = work;
    float* y = work + 2 * n;
    long long nbk = (n + 255) / 256;
    if (nbk > 4096) nbk = 4096;
    hipLaunchKernelGGL(k_phat_pack, dim3((unsigned)nbk), dim3(256), 0, st, a, (long long)na, b, (long long)nb, n, (float2*)z);
    const dim3 gA(8 * A.tiles_per_xcd, 1), gB(8 * B.tiles_per_xcd, three ? B.nplanes : 1), grow(R.R / 2 + 1, 1), blk(256);
    const size_t lc = EGR_LDS(p->sp.lds_col), lb = EGR_LDS(p->sp.lds_colb), lr = EGR_LDS(p->sp.lds_row);
    hipLaunchKernelGGL(k_col<0>, gA, blk, lc, st, A, M, N, -1.0f, p->d_work, z, (unsigned*)nullptr);
    if (three) hipLaunchKernelGGL(k_col<4>, gB, blk, lb, st, B, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
    hipLaunchKernelGGL(k_row<false>, grow, blk, lr, st, R, M, p->d_work);
    if (three) hipLaunchKernelGGL(k_col<3>, gB, blk, lb, st, B, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
    hipLaunchKernelGGL(k_col<5>, gA, blk, lc, st, A, M, N, 0.f, p->d_work, y, (unsigned*)nullptr);
    hipLaunchKernelGGL(k_phat_peak, dim3(1), dim3(1024), 0, st, (const float2*)y, n, (long long)max_shift, out4);
    EGR_HIP(hipGetLastError());
    return EGR_OK;
}

// ---- EGR_LDS_CANARY (debug builds): guard-band failures since the library was loaded, and a self-test of the guards' placement ----
namespace egr {
__global__ void k_canary_selftest(int where, int payload_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    EGR_LDS_CANARY_ARM(smem);
    float* cur = (float*)EGR_LDS_BASE(smem);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (where == 1) cur[payload_bytes / 4] = 1.f;          // one element past the payload
        if (where == 2) cur[-1] = 1.f;                         // one element before it
        if (where == 0) cur[payload_bytes / 4 - 1] = 1.f;      // the last payload element: legal
    }
}
}  // namespace egr

static long long canary_read() {
#ifdef EGR_LDS_CANARY
    unsigned v = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&v, HIP_SYMBOL(egr::g_lds_canary_fail), sizeof(v)) != hipSuccess) return -2;
    return (long long)v;
#else
    return -1;
#endif
}

extern "C" long long egr_lds_canary_failures(void) {
#ifdef EGR_LDS_CANARY
    const long long a = canary_read(), b = pz_canary_failures();
    return (a < 0 || b < 0) ? -2 : a + b;
#else
    return -1;
#endif
}

extern "C" long long egr_lds_canary_selftest(int where) {
#ifdef EGR_LDS_CANARY
    const long long before = canary_read();
    hipLaunchKernelGGL(egr::k_canary_selftest, dim3(1), dim3(64), EGR_LDS(1024), 0, where, 1024);
    const long long after = canary_read();
    return (before < 0 || after < 0) ? -2 : after - before;
#else
    (void)where;
    return -1;
#endif
}
