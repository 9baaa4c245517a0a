// Internal definitions shared by the Fat-Llama translation units (egr_fatllama.hip: packed-real plans, legacy chirp-z;
// egr_fatllama_pz.hip: paired chirp-z for every other length).  Not part of the public header.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <initializer_list>
#include <vector>

#include "egr_common.h"
#include "egr_fft_device.h"
#include "egr_plan.h"

#define EGR_LDS_MAX (152 * 1024)      // dynamic LDS a kernel may ask for: the 160 KiB of a gfx950 CU minus room for its static arrays

// Debug build flag EGR_LDS_CANARY (make CANARY=1): every dynamic LDS region of the Fat-Llama kernels gets a 64-byte guard band in
// front of the payload and one behind it, filled with a sentinel when the kernel starts and checked when it returns (a
// workgroup-local overflow -- a reduction scratch sized for fewer waves, a tile index past its end -- lands in a guard instead
// of silently in a neighbour's data).  The host adds EGR_LDS_GUARD bytes to every dynamic-LDS request (EGR_LDS); the kernel finds
// the end of its region from the dispatch packet (group_segment_size minus its static LDS).  Failures are counted per translation
// unit and read with egr_lds_canary_failures().
#ifdef EGR_LDS_CANARY
#define EGR_LDS_GUARD 128
#define EGR_LDS_HEAD 64
#else
#define EGR_LDS_GUARD 0
#define EGR_LDS_HEAD 0
#endif
#define EGR_LDS(bytes) ((size_t)(bytes) + EGR_LDS_GUARD)
#define EGR_LDS_BASE(smem) ((smem) + EGR_LDS_HEAD)

namespace egr {

#ifdef EGR_LDS_CANARY
static __device__ unsigned g_lds_canary_fail;
struct LdsCanary {
    char* head;
    char* tail;
    __device__ __forceinline__ LdsCanary(char* smem) {
        const unsigned total = ((const unsigned*)__builtin_amdgcn_dispatch_ptr())[7];          // group_segment_size
        const unsigned dyn = total - __builtin_amdgcn_groupstaticsize();
        head = smem;
        tail = smem + dyn - 64;
        if (threadIdx.x < 16) {
            ((unsigned*)head)[threadIdx.x] = 0xC0FFEE00u + threadIdx.x;
            ((unsigned*)tail)[threadIdx.x] = 0xC0FFEE00u + threadIdx.x;
        }
    }
    __device__ __forceinline__ ~LdsCanary() {
        __syncthreads();
        if (threadIdx.x < 16 && (((unsigned*)head)[threadIdx.x] != 0xC0FFEE00u + threadIdx.x || ((unsigned*)tail)[threadIdx.x] != 0xC0FFEE00u + threadIdx.x))
            atomicAdd(&g_lds_canary_fail, 1u);
    }
};
#define EGR_LDS_CANARY_ARM(smem) LdsCanary lds_canary__(smem)
#else
#define EGR_LDS_CANARY_ARM(smem) do { } while (0)
#endif

#define EGR_STAMP(P, SLOT) do { if ((P).trace && threadIdx.x == 0) (P).trace[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (SLOT)] = wall_clock64(); } while (0)

// Twiddle W_T^r for r < T as a product of two table entries: thi[r >> sh] * tlo[r & (2^sh - 1)]
// (tables of ~sqrt(T) entries each, generated in long double).  The tables and the product are DOUBLE precision and the
// result is rounded to float once: every loop iteration multiplies element e by the same W on the way in and conj(W) on
// the way out, so |W|^2 - 1 is a per-element gain that compounds over the iterations -- (1 + eps)^800.  A product of two
// float-rounded factors leaves eps ~ 1.2e-7, the single rounding ~ 4e-8 (what any float32 twiddle table has); measured
// on 800 iterations: rms error vs float64 2.9x -> 1.xx the pocketfft oracle's (tests/test_gpu_fatllama.py).
struct Tw2 {
    const dcplx* hi;
    const dcplx* lo;
    int sh;
};
__device__ __forceinline__ dcplx tw2d(const Tw2& t, unsigned r) {
    return dcmul(t.hi[r >> t.sh], t.lo[r & ((1u << t.sh) - 1u)]);
}
__device__ __forceinline__ cplx tw2(const Tw2& t, unsigned r) {
    const dcplx w = tw2d(t, r);
    return make_float2((float)w.x, (float)w.y);
}

// One strided ("column") pass: `nplanes` matrices [L][ncols] (row-major), transform along L for a tile of TC
// adjacent columns, twiddle W_(L*ncols)^(col*k).
struct ColP {
    FftDesc f;
    int L, ncols, nplanes;
    int TC, TClog2, ntiles, tiles_per_xcd;
    const cplx* tw;        // W_L stage table
    const dcplx* twd;      // the same table in double precision (power-twiddle path)
    const cplx* stw;       // per-stage butterfly-ordered tables of a compile-time schedule (k_col<MODE, SCHED > 0>)
    long long* trace;      // dev: 100 MHz wall-clock stamps per phase, [block][8] (EGR_FL_TRACE)
    Tw2 big;               // W_(L*ncols)^r
};

// The contiguous ("row") pass: R rows of length L; row rho(o) holds Z[o + R*k], o = ka + Ma*kb, rho = ka*Mb + kb.
struct RowP {
    FftDesc f;
    int L, R, Ma, Mb;
    const cplx* tw;        // W_L stage table
    const dcplx* twd;      // the same table in double precision (power-twiddle path)
    const cplx* stw;       // per-stage butterfly-ordered tables of a compile-time schedule (k_row<., SCHED > 0>)
    long long* trace;      // dev: 100 MHz wall-clock stamps per phase, [block][8] (EGR_FL_TRACE)
    Tw2 wo;                // W_N^o, o < R
    const dcplx* wk;       // W_(2L)^k = W_N^(R*k), k < L (double: multiplied with W_N^o in double, rounded once)
    float thr2, inv_M;
    double inv_M_d;        // 1/M in double: the loop's scaling is applied in double and rounded once (a float 1/M is off by up to
                           // 6e-8 relative, the SAME way every iteration -- 5e-5 after 800)
    const float* gain;     // optional [C][M+1] real gain per half-spectrum bin (replaces the threshold)
    int phat;              // 1: the state is z = a + i b of two REAL signals; replace it by the PHAT-weighted cross-spectrum
    long long band_lo;     // > 0: keep half-spectrum bins k >= band_lo, zero the others (replaces the threshold); band = 1 selects it
    int band;
    // threshold variants (SPEC.md section 3).  max2 != nullptr: the level is thr * sqrt(max2[ch]) with max2[ch] = max_k |X[k]|^2 of
    // THIS iteration's spectrum (float bits, written by k_row<true>); soft: X max(0, 1 - t/|X|) instead of X [|X| > t].
    const unsigned* max2;
    unsigned* max2_out;    // k_row<true> only: where the maximum goes
    // carried maximum (round 6): the spectrum the NEXT iteration will see is this iteration's post-shrink spectrum (S keeps the
    // Hermitian symmetry, so fft(real(ifft(S(X)))) = S(X) up to round-off) -- the hook leaves max |S(X)|^2 in max2_next, and
    // clears the ring slot two iterations ahead (max2_zero)
    unsigned* max2_next;
    unsigned* max2_zero;
    float thr;
    int soft;
};

// tables of the two-barrier loop kernels (egr_fatllama_wl.h)
struct WlRowT {
    const cplx* t1;        // [Q^2][N1]: W_L^(n2 k1)
    const cplx* t2;        // [Q][Q]:    W_(Q^2)^(b c)
    const cplx* t1l;       // the low parts of the same tables: t + tl = W to 2^-48 (cmul2)
    const cplx* t2l;
    const dcplx* t2d;      // [Q]: W_(Q^2)^l in double (the ratio of a lane's run over c)
    dcplx hook_step;       // W_(2Q): ratio of the pair twiddles between a lane's consecutive registers
};
struct WlColT {
    const cplx* t3;        // [25][25]: W_625^(b c)
    const cplx* t3l;       // low parts
};

__device__ __forceinline__ void atomic_max_abs(unsigned* slot, float v) {
    // non-negative IEEE floats order like unsigned ints
    atomicMax(slot, __float_as_uint(v));
}

// Slots of the spectrum maxima of the relative threshold: per (ring slot, channel) EGR_FL_MAX_SUB words on lines of their own
// (a line serves about one device-scope atomic per 25 ns, DESIGN.md 4.2b: a workgroup commits to sub-slot blockIdx.x mod 8 and
// the reader takes the maximum of the eight).  All pointers below are the (slot, first channel of the launch) base.
#define EGR_FL_MAX_SUB 8
#define EGR_FL_MAX_LINE 32                                        // words per 128-byte line
#define EGR_FL_MAX_STRIDE (EGR_FL_MAX_SUB * EGR_FL_MAX_LINE)      // words per (slot, channel)
#define EGR_FL_MAX_RING 25                                        // ring slots = iterations of the captured loop graph
__device__ __forceinline__ float fl_max2_read(const unsigned* base, int ch) {
    const unsigned* b = base + (size_t)ch * EGR_FL_MAX_STRIDE;
    unsigned m = b[0];
#pragma unroll
    for (int s = 1; s < EGR_FL_MAX_SUB; ++s) { const unsigned v = b[s * EGR_FL_MAX_LINE]; m = v > m ? v : m; }
    return __uint_as_float(m);                                    // non-negative floats order like unsigned ints
}
__device__ __forceinline__ void fl_max2_commit(unsigned* base, int ch, float v) {
    atomicMax(base + (size_t)ch * EGR_FL_MAX_STRIDE + (blockIdx.x & (EGR_FL_MAX_SUB - 1)) * EGR_FL_MAX_LINE, __float_as_uint(v));
}
// called by threads tid < EGR_FL_MAX_SUB of ONE workgroup per channel
__device__ __forceinline__ void fl_max2_clear(unsigned* base, int ch, int tid) {
    base[(size_t)ch * EGR_FL_MAX_STRIDE + tid * EGR_FL_MAX_LINE] = 0u;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
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


struct ChirpP {
    Tw2 w;                   // W_(2N)^r
    unsigned long long N;    // transform length
    float inv_N;
    double inv_2N_d;         // 1 / (2 N) for the remainder estimate of chirp()
    int band;                // 1: the spectrum hook keeps bins min(n, N - n) >= band_lo instead of thresholding
    unsigned long long band_lo;
    int soft;                // 1: soft shrink X max(0, 1 - thr/|X|) instead of the hard threshold
    const unsigned* max2;    // relative threshold: max_k |X[k]|^2 of this iteration per channel (fl_max2_read); level = thr sqrt(.)
    unsigned* max2_out;      // k_colz<3, 1> only: where that maximum goes
};
__device__ __forceinline__ cplx chirp(const ChirpP& c, unsigned long long n) {
    // n^2 mod 2N without the 64-bit division (~150 instructions per element of every hook): n^2 < 2^53 is exact in double, the
    // quotient estimate is off by at most one, two conditional corrections make the remainder exact
    const unsigned long long m = 2ULL * c.N, n2 = n * n;
    unsigned long long r;
    if (n2 < (1ULL << 53)) {
        const unsigned long long q = (unsigned long long)((double)n2 * c.inv_2N_d);
        long long d = (long long)(n2 - q * m);
        if (d < 0) d += (long long)m;
        if (d >= (long long)m) d -= (long long)m;
        r = (unsigned long long)d;
    } else {
        r = n2 % m;
    }
    return tw2(c.w, (unsigned)r);
}


struct PzPlan;          // egr_fatllama_pz.hip
}  // namespace egr

struct egr_fatllama_plan {
    int64_t n_in;
    int64_t n_out;        // real samples per channel the loop runs on: n_in * factor, or the explicit length of egr_fatllama_plan_create_n (factor 0)
    int C, factor, device;
    egr::FlSplit sp;
    egr::ColP colA, colB;
    egr::RowP row;
    std::vector<void*> dev_allocs;
    bool bluestein;       // lengths outside the packed-real plans: chirp-z over P = sp.M complex points
    int pz_kind;          // 0: packed-real plan or the legacy full-complex chirp-z; 1 / 2: paired chirp-z (egr_fatllama_pz.hip)
    egr::PzPlan* pz;
    egr::ChirpP chirp;
    egr::cplx* d_bhat;         // FFT_P(b) / P in the passes' transposed layout
    egr::cplx* d_work;
    unsigned* d_peaks;   // [3*C]: peak_in[C], peak_out[C], peak_y[C]
    unsigned* d_max2;    // [max2_cap]: ring of per-(iteration, channel) max |X|^2 slots of the relative-threshold variant (fl_max2_*)
    size_t max2_cap;
    bool profiling;
    int threads;                  // workgroup size of the loop kernels (256 or 512)
    int row_sched, col_sched;     // compile-time schedule ids of the loop kernels (0: run-time schedule)
    int wl_row, wl_col;           // the two-barrier kernels of egr_fatllama_wl.h serve the row / outer column pass of the loop
    const void* wl_row_entry;     // WlRowEntry of the row length (k_row_wl<N1, Q>)
    int wl_inner;                 // ... and k_colb_wl the inner column pass of a three-level plan
    egr::ColP wl_colB;            // colB with the tile geometry of k_colb_wl
    const egr::cplx* wl_it;       // [LB][LA]: W_L^(b c) of the inner length
    egr::WlRowT wl_rt;
    egr::WlColT wl_ct;
    int nstreams;                 // channel groups run as concurrent pipelines (1 or 2)
    hipStream_t side;             // second pipeline's stream (forked from / joined to the caller's stream by events)
    int side_owned;               // 0: `side` was handed in by egr_fatllama_set_side_stream (not destroyed with the plan)
    hipEvent_t ev_fork, ev_join;
    hipStream_t cap;              // private capture stream
    hipGraphExec_t gexec;         // CH captured loop iterations of all pipelines (egr_fatllama_enhance)
    const float* g_out; float g_thr; int g_groups, g_iter_odd, use_graph;
    std::vector<hipEvent_t> ev;   // pairs (start, stop) tagged by kind
    std::vector<int> ev_kind;     // 0 = row, 1 = outer column pass, 2 = inner column pass
};


// host helpers defined in egr_fatllama.hip
int fl_upload(egr_fatllama_plan* p, const std::vector<float2>& h, const egr::cplx** d);
int fl_upload_d(egr_fatllama_plan* p, int L, const egr::dcplx** d);                                        // W_L^j in double
int fl_upload_dtab(egr_fatllama_plan* p, int64_t count, int64_t num, int64_t den, const egr::dcplx** d);   // exp(-2 pi i j num / den)
int fl_make_tw2(egr_fatllama_plan* p, int64_t T, egr::Tw2* out);                                           // hi / lo tables of W_T^r
// butterfly-ordered stage tables of a compile-time schedule (stages after the first), egr_fft_device.h lds_fft_sched_inplace
int fl_upload_sched_tables(egr_fatllama_plan* p, std::initializer_list<int> radices, const egr::cplx** d);
void fl_prof_begin(egr_fatllama_plan* p, int kind, hipStream_t st, size_t* slot);
void fl_prof_end(egr_fatllama_plan* p, hipStream_t st, size_t* slot);
// inner column pass of a three-level plan over `nstates` states of the plan (forward: FFT . twiddle, else twiddle^-1 . IFFT)
void fl_launch_inner(egr_fatllama_plan* p, bool forward, egr::cplx* work, int nstates, hipStream_t st);

// paired chirp-z (egr_fatllama_pz.hip)
int pz_build(egr_fatllama_plan* p, int kind);
long long pz_canary_failures();    // -1 without EGR_LDS_CANARY
bool pz_sched_has(int L, int nc);  // column length L and row length nc both have compile-time schedules
void pz_destroy(egr_fatllama_plan* p);
int pz_loop(egr_fatllama_plan* p, float* out, int max_iter, float thr, float thr0, const unsigned* thr0_rel, unsigned flags,
            unsigned* peak_out, hipStream_t st);
int pz_band_filter(egr_fatllama_plan* p, const float* x, int64_t band_lo, float* y, hipStream_t st);
