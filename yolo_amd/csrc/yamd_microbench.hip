// Two measurement kernels for the benchmark's roofline denominators: a dense bf16 MFMA loop with operands in
// registers (what the matrix pipes sustain on this device, at the clock it holds under that load) and a 16-byte
// streaming copy (what HBM sustains).  Not part of the detection path.
#include "yamd_common.h"

using namespace yamd;

namespace {

// Every wave issues 4 * iters v_mfma_f32_32x32x16_bf16 on four independent accumulators; operands come from a caller
// buffer (random data: the clock a chip holds on zeros is higher and would flatter the peak).
__global__ __launch_bounds__(256) void mfma_peak_kernel(const bf16x8 *__restrict__ operands, float *__restrict__ sink, int iters)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a = operands[lane], b = operands[64 + lane], c = operands[128 + lane], d = operands[192 + lane];
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, d, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, b, acc3, 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j] + acc2[j] + acc3[j];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// Every workgroup copies contiguous 16 KB pieces (one 16-byte access per lane and piece, UNROLL pieces in flight per
// lane before the first store); `mode` selects plain or non-temporal accesses.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
    const size_t chunk = (size_t)256 * UNROLL;                    // 16-byte elements per workgroup step
    for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n16) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n16) {
                if (NT) __builtin_nontemporal_store(v[u], dst + i);
                else dst[i] = v[u];
            }
        }
    }
}

// One lane samples the shader clock counter of its XCD and the constant-rate counter, then sleeps, n times: the ratio of
// the two differences is the shader clock the device holds while other streams keep it busy.  Bounded: the kernel ends
// after n * naps sleeps of 127 * 64 cycles.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *samples, int n, int naps)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        samples[2 * i] = __builtin_readcyclecounter();
        samples[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(127);
    }
}

}  // namespace

extern "C" int yamd_microbench_clock_probe(unsigned long long *samples, int n_samples, int naps, void *stream)
{
    YAMD_REQUIRE(samples && n_samples > 0 && n_samples <= (1 << 16) && naps > 0 && naps <= 64, "bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, samples, n_samples, naps);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

extern "C" int yamd_microbench_mfma_bf16(const void *operands, float *sink, int blocks, int iters, void *stream)
{
    YAMD_REQUIRE(operands && sink && blocks > 0 && blocks <= 65535 && iters > 0, "bad argument");
    YAMD_REQUIRE(((uintptr_t)operands % 16) == 0, "operands must be 16-byte aligned");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8 *)operands, sink, iters);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

extern "C" int yamd_microbench_copy(const void *src, void *dst, size_t bytes, int variant, void *stream)
{
    YAMD_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0, "bad argument");
    YAMD_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "buffers must be 16-byte aligned");
    YAMD_REQUIRE(variant >= 0 && variant < 4, "variant %d: 0 = 4 pieces in flight, 1 = 8, 2 / 3 = the same non-temporal", variant);
    const size_t n16 = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(256 * 16), block(256);
    if (variant == 0) hipLaunchKernelGGL((stream_copy_kernel<4, false>), grid, block, 0, st, (const u32x4 *)src, (u32x4 *)dst, n16);
    else if (variant == 1) hipLaunchKernelGGL((stream_copy_kernel<8, false>), grid, block, 0, st, (const u32x4 *)src, (u32x4 *)dst, n16);
    else if (variant == 2) hipLaunchKernelGGL((stream_copy_kernel<4, true>), grid, block, 0, st, (const u32x4 *)src, (u32x4 *)dst, n16);
    else hipLaunchKernelGGL((stream_copy_kernel<8, true>), grid, block, 0, st, (const u32x4 *)src, (u32x4 *)dst, n16);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}
