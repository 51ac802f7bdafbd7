// Shared host/device helpers for libyolo_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <atomic>

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

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised, once per
// (kernel, device).  A DeviceOnce holds one bit per device ordinal.  Every such kernel is also listed in a registry so
// that yamd_init() can raise them all for the current device up front (nothing is then left to do on the first launch,
// e.g. under stream capture).  Raising is idempotent, so two host threads racing on the same kernel are harmless.
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
};
bool register_kernel(const void *kernel, DeviceOnce *once, int lds_bytes);
int raise_dynamic_lds(DeviceOnce &once, const void *kernel, int lds_bytes);   // YAMD_OK / YAMD_EHIP, for the current device

// KernelAttr<&kernel, bytes>::ensure(): registers the kernel when the library is loaded and raises its limit on first
// use on a device
template <auto KERNEL, int LDS_BYTES>
struct KernelAttr {
    static inline DeviceOnce once;
    static inline const bool registered = register_kernel((const void *)KERNEL, &once, LDS_BYTES);
    static int ensure()
    {
        (void)registered;     // odr-use: the registration runs at load time
        return raise_dynamic_lds(once, (const void *)KERNEL, LDS_BYTES);
    }
};

// The confidence filter of the NMS front end run inside another kernel (the decode of the predict path filters the rows
// it has just decoded, while they are in LDS): where its survivors go inside an NMS workspace.
struct FilterTargets {
    unsigned *img_count, *cls_count;              // per image / per (image, class) candidate counters, zero at launch
    unsigned long long *tmp_key, *tmp_meta;       // unordered candidate lists, `a` slots per image
    int a, nc, n_extra;
    float conf_thr;
};
// yamd_nms.hip: checks the shape, lays out `workspace`, enqueues the clearing of its counters and fills `ft`
int nms_filter_targets(int batch, int a, int c, int n_extra, float conf_thr, void *workspace, size_t workspace_bytes, void *stream,
                       FilterTargets *ft);
// yamd_decode.hip: yamd_decode_anchor_levels that also filters: rows that pass are written to `out` and entered into `ft`'s
// lists, the others are not written at all.  YAMD_ENOSPC when the class count does not fit the kernel's LDS tile
int decode_anchor_levels_filtered(int kind, const float *const *raws, const int *hs, const int *ws, const int *strides,
                                  const float *anchors_wh_host, int n_levels, int batch, int na, int c, int point_num, float img_size,
                                  int clamp_points, float *out, const FilterTargets &ft, void *stream);

// per-device set-up of the NMS kernels and their static scratch array (yamd_nms.hip), called by yamd_init
int nms_init_current_device(int device);
// resolves the address of the fused stem's tile counters on the current device (yamd_stem.hip), called by yamd_init
int stem_init_current_device(int device);

}  // namespace yamd
