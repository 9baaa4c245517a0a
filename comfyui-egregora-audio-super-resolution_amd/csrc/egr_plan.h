// Host-side transform planning shared by the Fat-Llama engine and the STFT kernel.
#pragma once
#include <stdint.h>
#include <vector>

#include "egr_fft_device.h"

namespace egr {

// Radix schedule for an in-LDS transform of length L: 4s first, one 2, then odd primes ascending.
// Returns false when L has a prime factor > max_prime or needs more than EGR_MAX_STAGES stages.  Primes <= 13 are register
// butterflies; max_prime > 13 admits generic stages (direct DFT from LDS) for the prime factors 17 .. max_prime.
bool make_schedule(int L, FftDesc* d, int max_prime = 13);

// W_L^j = exp(-2*pi*i*j/L * mul), j in [0,count), evaluated in double, rounded once to float.
// (general form: exp(-2*pi*i * j * num / den))
void make_twiddles(std::vector<float2>& out, int64_t count, int64_t num, int64_t den);
void make_twiddles_d(std::vector<double2>& out, int64_t count, int64_t num, int64_t den);

struct FlSplit {
    bool ok;
    int levels;               // 2: M = M1*M2 ; 3: M = M1*M2*M3 (long inputs)
    int64_t N, M;
    int M1, M2, M3;           // M3 = 1 for two levels; the LAST factor is the contiguous row length
    int TC, TClog2;           // column-tile width of pass A
    int TCb, TCblog2;         // column-tile width of pass B (3 levels)
    FftDesc f1, f2, f3;
    size_t lds_col, lds_colb, lds_row;
};

// Choose the factorisation of M = N/2 for the multi-step transform of the packed half-length complex sequence.
FlSplit plan_split(int64_t N, int m1_hint, int tc_hint, int max_col = 2048);
// Explicit factorisation (tests / tuning): m3 == 1 selects two levels.  ok == false when it does not fit.
FlSplit plan_split_explicit(int64_t N, int m1, int m2, int m3, int tc_hint, int max_row = 4096);

}  // namespace egr
