// Row lengths with a two-barrier row kernel k_row_wl<N1, Q> (csrc/egr_fatllama_wl.h), L = N1 Q^2 -- shared by the device code, which
// instantiates them, and the host planner (csrc/egr_plan.cpp), which prefers plans whose rows are on the list.
#pragma once
// X(L, N1, Q)
#define EGR_WL_ROW_LIST(X) \
    X(384, 6, 8) X(576, 4, 12) X(768, 12, 8) X(1152, 8, 12) X(1536, 24, 8) X(1728, 12, 12) X(1920, 30, 8) X(2304, 16, 12) \
    X(2880, 20, 12) X(3072, 12, 16) X(3456, 24, 12) X(4032, 28, 12) X(4096, 16, 16) X(4608, 32, 12) \
    /* rows of 50 T points next to columns of 441: T seconds at 44.1 kHz, T = 2 N1 */ \
    X(400, 4, 10) X(600, 6, 10) X(800, 8, 10) X(1000, 10, 10) X(1200, 12, 10) X(1400, 14, 10) X(1600, 16, 10) X(1800, 18, 10) \
    X(2000, 20, 10) X(2400, 24, 10) X(2800, 28, 10) X(3000, 30, 10) X(3200, 32, 10) \
    /* odd cross radices: 6 / 10 / 14 / 18 / 22 / 26 / 30 / 42 / 50 s at 44.1 kHz, 25 / 35 / 100 s at 48 kHz */ \
    X(300, 3, 10) X(500, 5, 10) X(700, 7, 10) X(900, 9, 10) X(1100, 11, 10) X(1300, 13, 10) X(1500, 15, 10) X(2100, 21, 10) X(2500, 25, 10) \
    X(960, 15, 8) X(1344, 21, 8) X(3840, 15, 16)
