// Common host-side helpers for libegregora_amd.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>

#include "../../include/egregora_amd.h"

namespace egr {

void set_error(const char* fmt, ...);

#define EGR_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            egr::set_error("%s -> %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return EGR_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define EGR_CHECK(cond, code, ...)          \
    do {                                    \
        if (!(cond)) {                      \
            egr::set_error(__VA_ARGS__);    \
            return (code);                  \
        }                                   \
    } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// egr_nn_conv3x3.hip: the k_conv3x3_isp instantiation launch_conv3x3_is picks (profile labels of egr_flashsr.cpp)
const char* conv3x3_is_name(int cout, bool gn, bool silu);

}  // namespace egr
