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

// four independent 16-byte loads in flight per lane before the first store
__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

extern "C" int yamd_microbench_mfma_bf16(const void *operands, float *sink, int blocks, int iters, void *stream)
{
    YAMD_REQUIRE(operands && sink && blocks > 0 && blocks <= 65535 && iters > 0, "bad argument");
    YAMD_REQUIRE(((uintptr_t)operands % 16) == 0, "operands must be 16-byte aligned");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8 *)operands, sink, iters);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

extern "C" int yamd_microbench_copy(const void *src, void *dst, size_t bytes, void *stream)
{
    YAMD_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0, "bad argument");
    YAMD_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "buffers must be 16-byte aligned");
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, (uint4 *)dst,
                       bytes / 16);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}
