// Inference pre-processing on the device: aspect-preserving bicubic resize (Pillow's algorithm, bit for bit),
// centred on a (128,128,128) canvas, /255, HWC uint8 -> CHW float32.
//
// Pillow resamples separably, horizontal pass first, each pass rounded and clipped to uint8.  The coefficients
// are computed in double and rounded to 22-bit fixed point; they are produced here by a small kernel that uses the
// same double-precision operations in the same order (build with -ffp-contract=off), so no host buffer or
// synchronisation is involved.
#include "yamd_common.h"

#include <cmath>

using namespace yamd;

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double bicubic_filter(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// one thread per output index: bounds[xx] = (first input index, tap count), kk[xx][0..ksize) fixed-point taps
__global__ void coef_kernel(int in_size, int out_size, int ksize, int *bounds, int *kk)
{
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
    int *k = kk + (size_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
        int v = 0;
        if (x < xmax) {
            double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w = w / ww;
            v = w >= 0 ? (int)(w * (double)(1 << PRECISION_BITS) + 0.5) : (int)(-0.5 + w * (double)(1 << PRECISION_BITS));
        }
        k[x] = v;
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ int clip8(int acc)
{
    const int v = acc >> PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: img (h, w, 3) uint8 with a row stride -> tmp (h, rw, 3) uint8
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char *img, int h, int row_stride, int rw, int ksize,
                                                         const int *bounds, const int *kk, unsigned char *tmp)
{
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)h * rw) return;
    const int xx = (int)(i % rw), y = (int)(i / rw);
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int *k = kk + (size_t)xx * ksize;
    const unsigned char *src = img + (size_t)y * row_stride + (size_t)xmin * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        const int c = k[x];
        s0 += src[3 * x] * c;
        s1 += src[3 * x + 1] * c;
        s2 += src[3 * x + 2] * c;
    }
    unsigned char *d = tmp + ((size_t)y * rw + xx) * 3;
    d[0] = (unsigned char)clip8(s0);
    d[1] = (unsigned char)clip8(s1);
    d[2] = (unsigned char)clip8(s2);
}

// vertical pass + paste on the grey canvas + /255 + CHW float
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char *tmp, int rw, int rh, int dw, int dh, int th,
                                                              int tw, int ksize, const int *bounds, const int *kk, float *out)
{
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= (long long)th * tw) return;
    const int ox = (int)(i % tw), oy = (int)(i / tw);
    int v0 = 128, v1 = 128, v2 = 128;
    const int yy = oy - dh, xx = ox - dw;
    if (yy >= 0 && yy < rh && xx >= 0 && xx < rw) {
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const int *k = kk + (size_t)yy * ksize;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < cnt; ++y) {
            const unsigned char *s = tmp + ((size_t)(ymin + y) * rw + xx) * 3;
            const int c = k[y];
            s0 += s[0] * c;
            s1 += s[1] * c;
            s2 += s[2] * c;
        }
        v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
    }
    const size_t plane = (size_t)th * tw;
    out[i] = (float)((double)v0 / 255.0);
    out[plane + i] = (float)((double)v1 / 255.0);
    out[2 * plane + i] = (float)((double)v2 / 255.0);
}

struct Geometry {
    double ratio;
    int rw, rh, dw, dh, kx, ky;
    size_t off_tmp, off_bx, off_kx, off_by, off_ky, total;
};

int ksize_for(int in_size, int out_size)
{
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}

size_t up256(size_t v) { return (v + 255) / 256 * 256; }

bool geometry(int h, int w, int th, int tw, Geometry &g)
{
    if (h <= 0 || w <= 0 || th <= 0 || tw <= 0) return false;
    const double rx = 1.0 * tw / w, ry = 1.0 * th / h;
    g.ratio = rx < ry ? rx : ry;
    g.rw = (int)(g.ratio * w);
    g.rh = (int)(g.ratio * h);
    if (g.rw <= 0 || g.rh <= 0) return false;
    g.dw = (tw - g.rw) / 2;
    g.dh = (th - g.rh) / 2;
    g.kx = ksize_for(w, g.rw);
    g.ky = ksize_for(h, g.rh);
    size_t o = 0;
    g.off_tmp = o; o += up256((size_t)h * g.rw * 3);
    g.off_bx = o; o += up256((size_t)g.rw * 2 * sizeof(int));
    g.off_kx = o; o += up256((size_t)g.rw * g.kx * sizeof(int));
    g.off_by = o; o += up256((size_t)g.rh * 2 * sizeof(int));
    g.off_ky = o; o += up256((size_t)g.rh * g.ky * sizeof(int));
    g.total = o;
    return true;
}

}  // namespace

extern "C" size_t yamd_letterbox_workspace_bytes(int h, int w, int target_h, int target_w)
{
    Geometry g;
    return geometry(h, w, target_h, target_w, g) ? g.total : 0;
}

extern "C" int yamd_letterbox_u8(const uint8_t *img, int h, int w, int row_stride, int target_h, int target_w, float *out,
                                 int *geom_host, double *ratio_host, void *workspace, size_t workspace_bytes, void *stream)
{
    YAMD_REQUIRE(img && out && workspace, "null argument");
    Geometry g;
    YAMD_REQUIRE(geometry(h, w, target_h, target_w, g), "bad image size %dx%d for a %dx%d target", h, w, target_h, target_w);
    YAMD_REQUIRE(row_stride >= 3 * w, "row stride %d is smaller than a row of %d RGB pixels", row_stride, w);
    if (workspace_bytes < g.total) {
        set_error("letterbox workspace too small: %zu < %zu", workspace_bytes, g.total);
        return YAMD_ENOSPC;
    }
    YAMD_REQUIRE((long long)h * g.rw < (1ll << 31) && (long long)target_h * target_w < (1ll << 31), "image too large");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    unsigned char *tmp = (unsigned char *)(ws + g.off_tmp);
    int *bx = (int *)(ws + g.off_bx), *kx = (int *)(ws + g.off_kx), *by = (int *)(ws + g.off_by), *ky = (int *)(ws + g.off_ky);
    hipLaunchKernelGGL(coef_kernel, dim3(ceil_div(g.rw, 128)), dim3(128), 0, st, w, g.rw, g.kx, bx, kx);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(coef_kernel, dim3(ceil_div(g.rh, 128)), dim3(128), 0, st, h, g.rh, g.ky, by, ky);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_h_kernel, dim3(ceil_div((long long)h * g.rw, 256)), dim3(256), 0, st, img, h, row_stride, g.rw,
                       g.kx, bx, kx, tmp);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_v_norm_kernel, dim3(ceil_div((long long)target_h * target_w, 256)), dim3(256), 0, st, tmp, g.rw,
                       g.rh, g.dw, g.dh, target_h, target_w, g.ky, by, ky, out);
    YAMD_LAUNCH_CHECK();
    if (geom_host) {
        geom_host[0] = g.rw;
        geom_host[1] = g.rh;
        geom_host[2] = g.dw;
        geom_host[3] = g.dh;
    }
    if (ratio_host) *ratio_host = g.ratio;
    return YAMD_OK;
}
