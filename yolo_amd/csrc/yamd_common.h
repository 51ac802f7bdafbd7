// Shared host/device helpers for libyolo_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/yolo_amd.h"

namespace yamd {

// thread-local error text returned by yamd_last_error()
void set_error(const char *fmt, ...);

#define YAMD_REQUIRE(cond, ...)                      \
    do {                                             \
        if (!(cond)) {                               \
            yamd::set_error(__VA_ARGS__);            \
            return YAMD_EINVAL;                      \
        }                                            \
    } while (0)

#define YAMD_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            yamd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return YAMD_EHIP;                                                            \
        }                                                                                \
    } while (0)

// kernel launches do not return an error themselves; pick it up right after the launch
#define YAMD_LAUNCH_CHECK() YAMD_HIP(hipGetLastError())

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace yamd
