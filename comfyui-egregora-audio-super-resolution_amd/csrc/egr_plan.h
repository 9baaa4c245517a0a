// Host-side transform planning shared by the Fat-Llama engine and the STFT kernel.
#pragma once
#include <stdint.h>
#include <vector>

#include "egr_fft_device.h"

namespace egr {

// Radix schedule for an in-LDS transform of length L: 4s first, one 2, then odd primes ascending.
// Returns false when L has a prime factor > 13 or needs more than EGR_MAX_STAGES stages.
bool make_schedule(int L, FftDesc* d);

// W_L^j = exp(-2*pi*i*j/L * mul), j in [0,count), evaluated in double, rounded once to float.
// (general form: exp(-2*pi*i * j * num / den))
void make_twiddles(std::vector<float2>& out, int64_t count, int64_t num, int64_t den);

struct FlSplit {
    bool ok;
    int64_t N, M;
    int M1, M2, TC, TClog2;
    FftDesc f1, f2;
    size_t lds_col, lds_row;
};

// Choose M = M1*M2 for the four-step transform of the packed half-length complex sequence.
FlSplit plan_split(int64_t N, int m1_hint, int tc_hint);

}  // namespace egr
