// Which bf16 MFMA shape sustains more FLOP/s on this device, on random and on zero operands: the same 64x64 output
// tile per wave as 2x2 fragments of 32x32x16 or 4x4 fragments of 16x16x32, operands in registers or re-read from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(256) void k32(const bf16x8 *ops, float *sink, int iters)
{
    extern __shared__ bf16x8 sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 4 * 64; i += 256) sm[i] = ops[i & 255];
    __syncthreads();
    bf16x8 a[2], b[2];
    for (int j = 0; j < 2; ++j) { a[j] = ops[j * 64 + lane]; b[j] = ops[128 + j * 64 + lane]; }
    f32x16 acc[2][2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {      // K = 32 per iteration
            if (LDS) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { a[j] = sm[(wave * 4 + j) * 64 + lane]; b[j] = sm[(wave * 4 + 2 + j) * 64 + lane]; }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int j = 0; j < 16; ++j) s += acc[m][n][j];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool LDS>
__global__ __launch_bounds__(256) void k16(const bf16x8 *ops, float *sink, int iters)
{
    extern __shared__ bf16x8 sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 8 * 64; i += 256) sm[i] = ops[i & 255];
    __syncthreads();
    bf16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j) { a[j] = ops[((j & 1) * 64 + lane + j * 7) & 255]; b[j] = ops[(128 + (j & 1) * 64 + lane + j * 5) & 255]; }
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = sm[(wave * 8 + j) * 64 + lane]; b[j] = sm[(wave * 8 + 4 + j) * 64 + lane]; }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int j = 0; j < 4; ++j) s += acc[m][n][j];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv)
{
    const int blocks = 256 * 2, iters = 20000;
    std::vector<unsigned short> h(256 * 8);
    float *sink; void *ops;
    hipMalloc(&sink, blocks * 256 * 4); hipMalloc(&ops, 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        srand(1);
        for (auto &v : h) { float f = zero ? 0.f : (rand() / (float)RAND_MAX * 2 - 1); unsigned u; std::memcpy(&u, &f, 4); v = u >> 16; }
        hipMemcpy(ops, h.data(), 256 * 16, hipMemcpyHostToDevice);
        const double flop = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 32;
        auto run = [&](const char *name, auto kern, size_t lds) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, (const bf16x8 *)ops, sink, iters);
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, (const bf16x8 *)ops, sink, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("%-8s %-28s %8.3f ms  %7.0f TFLOP/s\n", zero ? "zeros" : "random", name, ms, flop / ms / 1e9);
            fflush(stdout);
        };
        run("32x32x16 registers", k32<false>, 16 * 1024);
        run("16x16x32 registers", k16<false>, 32 * 1024);
        run("32x32x16 operands from LDS", k32<true>, 16 * 1024);
        run("16x16x32 operands from LDS", k16<true>, 32 * 1024);
    }
    return 0;
}
