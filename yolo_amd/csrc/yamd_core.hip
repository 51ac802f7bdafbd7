// Error channel + layout / resampling kernels of the conv graph (all HBM-bound, 16-byte accesses).
#include "yamd_common.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace yamd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
struct KernelEntry {
    const void *kernel;
    DeviceOnce *once;
    int lds_bytes;
};
std::mutex &registry_mutex()
{
    static std::mutex m;
    return m;
}
std::vector<KernelEntry> &registry()
{
    static std::vector<KernelEntry> r;     // function-local: usable from other files' static initialisers
    return r;
}
}  // namespace

bool register_kernel(const void *kernel, DeviceOnce *once, int lds_bytes)
{
    std::lock_guard<std::mutex> g(registry_mutex());
    registry().push_back({kernel, once, lds_bytes});
    return true;
}

int raise_dynamic_lds(DeviceOnce &once, const void *kernel, int lds_bytes)
{
    int dev = 0;
    YAMD_HIP(hipGetDevice(&dev));
    YAMD_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d outside [0, 64)", dev);
    const unsigned long long bit = 1ull << dev;
    if (once.done.load(std::memory_order_acquire) & bit) return YAMD_OK;
    YAMD_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    once.done.fetch_or(bit, std::memory_order_release);
    return YAMD_OK;
}
}  // namespace yamd

using namespace yamd;

extern "C" const char *yamd_last_error(void) { return g_err; }
extern "C" int yamd_version(void) { return 200; }

extern "C" int yamd_init(void)
{
    int dev = 0;
    YAMD_HIP(hipGetDevice(&dev));
    std::vector<KernelEntry> entries;
    {
        std::lock_guard<std::mutex> g(registry_mutex());
        entries = registry();
    }
    for (const KernelEntry &e : entries) {
        const int rc = raise_dynamic_lds(*e.once, e.kernel, e.lds_bytes);
        if (rc != YAMD_OK) return rc;
    }
    const int rc = nms_init_current_device(dev);
    return rc != YAMD_OK ? rc : stem_init_current_device(dev);
}

namespace {

// one thread per output pixel: 2x2 input patch of 3 planes -> 16 bf16 channels (32 bytes)
__global__ void focus_pack_kernel(const float *__restrict__ img, __bf16 *__restrict__ y, int y_cs, int B, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        const int oy = (int)((i / Wo) % Ho);
        const int b = (int)(i / ((long long)Wo * Ho));
        float v[2][2][3];   // [row parity][col parity][channel]
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float2 t = *(const float2 *)(img + (((long long)b * 3 + c) * H + (2 * oy + r)) * W + 2 * ox);
                v[r][0][c] = t.x;
                v[r][1][c] = t.y;
            }
        bf16x8 lo, hi;
        // channel order: (even row, even col), (odd row, even col), (even row, odd col), (odd, odd)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = (__bf16)v[0][0][c];
            lo[3 + c] = (__bf16)v[1][0][c];
        }
        lo[6] = (__bf16)v[0][1][0];
        lo[7] = (__bf16)v[0][1][1];
        hi[0] = (__bf16)v[0][1][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) hi[1 + c] = (__bf16)v[1][1][c];
        hi[4] = hi[5] = hi[6] = hi[7] = (__bf16)0.0f;
        __bf16 *dst = y + i * y_cs;
        *(bf16x8 *)dst = lo;
        *(bf16x8 *)(dst + 8) = hi;
    }
}

__global__ void nchw_to_nhwc16_kernel(const float *__restrict__ img, __bf16 *__restrict__ y, int y_cs, int B, int C, int H,
                                      int W)
{
    const long long hw = (long long)H * W, total = (long long)B * hw;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw, s = i % hw;
        bf16x8 lo, hi;
#pragma unroll
        for (int c = 0; c < 8; ++c) lo[c] = (__bf16)(c < C ? img[(b * C + c) * hw + s] : 0.0f);
#pragma unroll
        for (int c = 0; c < 8; ++c) hi[c] = (__bf16)(8 + c < C ? img[(b * C + 8 + c) * hw + s] : 0.0f);
        __bf16 *dst = y + i * y_cs;
        *(bf16x8 *)dst = lo;
        *(bf16x8 *)(dst + 8) = hi;
    }
}

// one thread per (output pixel, 8-channel piece)
__global__ void upsample2x_kernel(const __bf16 *__restrict__ x, int x_cs, __bf16 *__restrict__ y, int y_cs, int B, int H,
                                  int W, int C8)
{
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)B * Ho * Wo * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const long long pix = i / C8;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const long long b = pix / ((long long)Wo * Ho);
        const uint4 v = *(const uint4 *)(x + ((b * H + (oy >> 1)) * W + (ox >> 1)) * x_cs + c8 * 8);
        *(uint4 *)(y + pix * y_cs + c8 * 8) = v;
    }
}

// y[b, 2i+di, 2j+dj, c] = x[b, i, j, (2*di+dj)*C + c]
__global__ void depth_to_space2_kernel(const __bf16 *__restrict__ x, int x_cs, __bf16 *__restrict__ y, int y_cs, int B,
                                       int H, int W, int C8)
{
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)B * Ho * Wo * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const long long pix = i / C8;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const long long b = pix / ((long long)Wo * Ho);
        const int blk = 2 * (oy & 1) + (ox & 1);
        const uint4 v = *(const uint4 *)(x + ((b * H + (oy >> 1)) * W + (ox >> 1)) * x_cs + (blk * C8 + c8) * 8);
        *(uint4 *)(y + pix * y_cs + c8 * 8) = v;
    }
}

__global__ void copy_channels_kernel(const __bf16 *__restrict__ x, int x_cs, __bf16 *__restrict__ y, int y_cs,
                                     long long pixels, int C8)
{
    const long long total = pixels * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const long long pix = i / C8;
        *(uint4 *)(y + pix * y_cs + c8 * 8) = *(const uint4 *)(x + pix * x_cs + c8 * 8);
    }
}

// Order-preserving 16-bit integer keys of bf16 values, two per 32-bit word: flipping the magnitude bits of the negative
// values makes signed 16-bit integer order equal to numeric order (-0 sorts just below +0), and the map is its own
// inverse.  A max over such keys is one packed integer instruction per two values, where comparing as floats costs a
// conversion pair, a compare and a select per value.
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4v bf16_keys(u32x4v w) { return w ^ (((w >> 15) & 0x00010001u) * 0x7FFFu); }
__device__ __forceinline__ u32x4v key_max8(u32x4v a, u32x4v b)
{
    return __builtin_bit_cast(u32x4v, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b)));
}

__global__ void maxpool2_kernel(const __bf16 *__restrict__ x, int x_cs, __bf16 *__restrict__ y, int y_cs, int B, int H,
                                int W, int C8)
{
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const long long pix = i / C8;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const long long b = pix / ((long long)Wo * Ho);
        const __bf16 *s = x + ((b * H + 2 * oy) * W + 2 * ox) * x_cs + c8 * 8;
        const u32x4v a = bf16_keys(*(const u32x4v *)s), c = bf16_keys(*(const u32x4v *)(s + x_cs));
        const u32x4v d = bf16_keys(*(const u32x4v *)(s + (long long)W * x_cs)), e = bf16_keys(*(const u32x4v *)(s + (long long)(W + 1) * x_cs));
        *(u32x4v *)(y + pix * y_cs + c8 * 8) = bf16_keys(key_max8(key_max8(a, c), key_max8(d, e)));
    }
}

// One workgroup per (image, group of G 8-channel pieces): the whole h*w plane of the group (16*G bytes per pixel) sits in
// LDS; a (2R+1)x(2R+1) max is a row pass followed by a column pass, applied three times (R = 2: 5x5, R = 1: 3x3).
// Consecutive lanes take consecutive 16-byte pieces of a pixel, so with G = 8 every global access moves whole 128-byte
// lines (one piece per workgroup would fetch every line once per piece, from eight different L2s).
template <int R>
__global__ void maxpool_cascade_kernel(const __bf16 *__restrict__ x, int x_cs, __bf16 *__restrict__ y1,
                                        __bf16 *__restrict__ y2, __bf16 *__restrict__ y3, int y_cs, int H, int W, int groups,
                                        int G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4v *cur = (u32x4v *)smem;                          // the plane as integer keys (bf16_keys)
    const int n = H * W, total = n * G;
    u32x4v *tmp = cur + total;
    const int c0 = (blockIdx.x % groups) * G * 8;          // first channel of the group
    const long long b = blockIdx.x / groups;
    const int gshift = G == 8 ? 3 : (G == 4 ? 2 : (G == 2 ? 1 : 0));
    for (int i = threadIdx.x; i < total; i += blockDim.x)
        cur[i] = bf16_keys(*(const u32x4v *)(x + (b * n + (i >> gshift)) * x_cs + c0 + (i & (G - 1)) * 8));
    __syncthreads();
    __bf16 *outs[3] = {y1, y2, y3};
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int pix = i >> gshift, g = i & (G - 1);
            const int r = pix / W, c = pix - r * W;
            u32x4v m = cur[i];
#pragma unroll
            for (int d = -R; d <= R; ++d) {
                const int cc = c + d;
                if (d != 0 && cc >= 0 && cc < W) m = key_max8(m, cur[((r * W + cc) << gshift) + g]);
            }
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int pix = i >> gshift, g = i & (G - 1);
            const int r = pix / W, c = pix - r * W;
            u32x4v m = tmp[i];
#pragma unroll
            for (int d = -R; d <= R; ++d) {
                const int rr = r + d;
                if (d != 0 && rr >= 0 && rr < H) m = key_max8(m, tmp[((rr * W + c) << gshift) + g]);
            }
            *(u32x4v *)(outs[pass] + (b * n + pix) * y_cs + c0 + g * 8) = bf16_keys(m);
            // the column pass reads tmp only, so cur can be overwritten for the next pass
            cur[i] = m;
        }
        __syncthreads();
    }
}

int grid_for(long long total, int block) { long long g = (total + block - 1) / block; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int yamd_focus_pack(const float *images, void *y, int y_cstride, int batch, int h, int w, void *stream)
{
    YAMD_REQUIRE(images && y, "null argument");
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "focus needs even h,w (got %dx%d)", h, w);
    YAMD_REQUIRE(y_cstride >= 16 && y_cstride % 8 == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)images % 8) == 0,
                 "focus output needs 16-byte aligned pixels");
    const long long total = (long long)batch * (h / 2) * (w / 2);
    hipLaunchKernelGGL(focus_pack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, images,
                       (__bf16 *)y, y_cstride, batch, h, w);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

extern "C" int yamd_nchw_to_nhwc16(const float *images, void *y, int y_cstride, int batch, int cin, int h, int w,
                                   void *stream)
{
    YAMD_REQUIRE(images && y, "null argument");
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0 && cin > 0 && cin <= 16, "bad shape");
    YAMD_REQUIRE(y_cstride >= 16 && y_cstride % 8 == 0 && ((uintptr_t)y % 16) == 0, "output needs 16-byte aligned pixels");
    const long long total = (long long)batch * h * w;
    hipLaunchKernelGGL(nchw_to_nhwc16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, images,
                       (__bf16 *)y, y_cstride, batch, cin, h, w);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

extern "C" int yamd_upsample2x(const void *x, int x_cstride, void *y, int y_cstride, int batch, int h, int w, int c,
                               void *stream)
{
    YAMD_REQUIRE(x && y, "null argument");
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "bad shape (c=%d must be a multiple of 8)", c);
    YAMD_REQUIRE(x_cstride % 8 == 0 && y_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
                 "upsample needs 16-byte aligned pixels");
    const long long total = (long long)batch * 4 * h * w * (c / 8);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)x, x_cstride, (__bf16 *)y, y_cstride, batch, h, w, c / 8);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

extern "C" int yamd_depth_to_space2(const void *x, int x_cstride, void *y, int y_cstride, int batch, int h, int w, int c,
                                    void *stream)
{
    YAMD_REQUIRE(x && y, "null argument");
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "bad shape (c=%d must be a multiple of 8)", c);
    YAMD_REQUIRE(x_cstride >= 4 * c && x_cstride % 8 == 0 && y_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0 &&
                     ((uintptr_t)y % 16) == 0, "depth_to_space needs 16-byte aligned pixels");
    const long long total = (long long)batch * 4 * h * w * (c / 8);
    hipLaunchKernelGGL(depth_to_space2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)x, x_cstride, (__bf16 *)y, y_cstride, batch, h, w, c / 8);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

extern "C" int yamd_copy_channels(const void *x, int x_cstride, void *y, int y_cstride, long long pixels, int c,
                                  void *stream)
{
    YAMD_REQUIRE(x && y, "null argument");
    YAMD_REQUIRE(pixels > 0 && c > 0 && c % 8 == 0, "bad shape");
    YAMD_REQUIRE(x_cstride % 8 == 0 && y_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
                 "copy needs 16-byte aligned pixels");
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(pixels * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)x, x_cstride, (__bf16 *)y, y_cstride, pixels, c / 8);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

extern "C" int yamd_maxpool2(const void *x, int x_cstride, void *y, int y_cstride, int batch, int h, int w, int c,
                             void *stream)
{
    YAMD_REQUIRE(x && y, "null argument");
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && c > 0 && c % 8 == 0,
                 "maxpool2: bad shape %dx%dx%d", h, w, c);
    YAMD_REQUIRE(x_cstride % 8 == 0 && y_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
                 "maxpool2 needs 16-byte aligned pixels");
    const long long total = (long long)batch * (h / 2) * (w / 2) * (c / 8);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x,
                       x_cstride, (__bf16 *)y, y_cstride, batch, h, w, c / 8);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

namespace {
int maxpool_cascade(int radius, const void *x, int x_cstride, void *y1, void *y2, void *y3, int y_cstride, int batch, int h, int w,
                    int c, void *stream)
{
    YAMD_REQUIRE(x && y1 && y2 && y3, "null argument");
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0 && h * w <= 4096 && c > 0 && c % 8 == 0,
                 "maxpool cascade: bad shape %dx%dx%d (planes up to 4096 pixels, c %% 8 == 0)", h, w, c);
    YAMD_REQUIRE(x_cstride % 8 == 0 && y_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y1 % 16) == 0 &&
                     ((uintptr_t)y2 % 16) == 0 && ((uintptr_t)y3 % 16) == 0,
                 "maxpool needs 16-byte aligned pixels");
    // pieces per workgroup: as many as fit 128 KB of LDS (two planes), at most 8 (= whole 128-byte lines), dividing c / 8
    int G = 8;
    while (G > 1 && ((size_t)h * w * 32 * G > 128 * 1024 || (c / 8) % G != 0)) G >>= 1;
    const size_t lds = (size_t)h * w * 16 * 2 * G;
    auto kern = radius == 2 ? maxpool_cascade_kernel<2> : maxpool_cascade_kernel<1>;
    {
        const int rc = radius == 2 ? KernelAttr<&maxpool_cascade_kernel<2>, 128 * 1024>::ensure()
                                   : KernelAttr<&maxpool_cascade_kernel<1>, 128 * 1024>::ensure();
        if (rc != YAMD_OK) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(batch * (c / 8 / G)), dim3(G >= 4 ? 512 : 256), lds, (hipStream_t)stream, (const __bf16 *)x,
                       x_cstride, (__bf16 *)y1, (__bf16 *)y2, (__bf16 *)y3, y_cstride, h, w, c / 8 / G, G);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}
}  // namespace

extern "C" int yamd_maxpool5_cascade(const void *x, int x_cstride, void *y1, void *y2, void *y3, int y_cstride, int batch,
                                     int h, int w, int c, void *stream)
{
    return maxpool_cascade(2, x, x_cstride, y1, y2, y3, y_cstride, batch, h, w, c, stream);
}

extern "C" int yamd_maxpool3_cascade(const void *x, int x_cstride, void *y1, void *y2, void *y3, int y_cstride, int batch,
                                     int h, int w, int c, void *stream)
{
    return maxpool_cascade(1, x, x_cstride, y1, y2, y3, y_cstride, batch, h, w, c, stream);
}
