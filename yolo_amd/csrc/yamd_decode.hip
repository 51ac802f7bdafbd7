// Head decode kernels (HBM-bound).  Build with -ffp-contract=off: the arithmetic follows the
// reference's fp32 operation order one rounding at a time.
//
// Anchor heads: a workgroup takes PIX consecutive pixels of one image, copies their raw logits
// (PIX x na*C floats, contiguous in NHWC) into LDS with coalesced loads, and writes one contiguous
// run of PIX rows x C floats per anchor (rows of one anchor are consecutive in the output), so both
// sides of the kernel stream whole cache lines.
#include "yamd_common.h"

using namespace yamd;

namespace {

constexpr int PIX = 64;

struct AnchorDecodeParams {
    const float *raw;
    float *out;
    int H, W, na, C, kind, P, stride, merge, clamp_points;
    unsigned c_magic;         // ceil(2^22 / C): e / C == (e * c_magic) >> 22 for e * C < 2^22
    float hi;                 // img_size - 1
    float aw[8], ah[8];       // anchor / stride, fp32 division as torch does it
    long long rows_per_image, row_offset;
};

__device__ __forceinline__ float sigmoidf_exact(float t) { return 1.0f / (1.0f + expf(-t)); }

// Hardware exp2 / reciprocal (about 1 ulp each): sigmoid and exp within a few 1e-7 relative of the correctly
// rounded functions for the logit range of a detector head, far inside the 1e-5 the decode is held to.
__device__ __forceinline__ float fast_exp(float t) { return __builtin_amdgcn_exp2f(t * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_sigmoid(float t) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-t)); }

__global__ __launch_bounds__(256) void decode_anchor_kernel(AnchorDecodeParams p)
{
    extern __shared__ __attribute__((aligned(16))) float sraw[];     // PIX*na*C logits, then the grid x / y of the pixels
    const int HW = p.H * p.W;
    const int naC = p.na * p.C;
    float *sgx = sraw + PIX * naC, *sgy = sgx + PIX;
    const int p0 = blockIdx.x * PIX;
    const int np = min(PIX, HW - p0);
    const long long b = blockIdx.y;
    const float *src = p.raw + (b * HW + p0) * (long long)naC;
    const int total = np * naC;
    if (((uintptr_t)src & 15) == 0) {
        const int t4 = total >> 2;
        for (int i = threadIdx.x; i < t4; i += 256) ((float4 *)sraw)[i] = ((const float4 *)src)[i];
        for (int i = (t4 << 2) + threadIdx.x; i < total; i += 256) sraw[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < total; i += 256) sraw[i] = src[i];
    }
    if (threadIdx.x < PIX) {
        const int gp = p0 + threadIdx.x;
        sgx[threadIdx.x] = (float)(gp % p.W);
        sgy[threadIdx.x] = (float)(gp / p.W);
    }
    __syncthreads();

    // The np rows of one anchor are one contiguous run of np*C floats in the output: thread t writes elements
    // t, t+256, ... of that run, so every store instruction covers whole cache lines and no lane idles.
    const float fs = (float)p.stride;
    const int pe = 5 + 2 * p.P;
    const int run = np * p.C;
    for (int a = 0; a < p.na; ++a) {
        const float aw = p.aw[a], ah = p.ah[a];
        float *dst = p.out + ((b * p.rows_per_image + p.row_offset + (long long)a * HW + p0) * p.C);
        for (int e = threadIdx.x; e < run; e += 256) {
            const int pix = (int)(((unsigned)e * p.c_magic) >> 22);      // e / C (exact for e*C < 2^22)
            const int c = e - pix * p.C;
            const float t = sraw[pix * naC + a * p.C + c];
            const float gx = sgx[pix], gy = sgy[pix];
            float v;
            const bool is_box = c < 4;
            bool is_pt = false;
            if (p.kind == YAMD_DECODE_V3) {
                const float s = fast_sigmoid(t), ex = fast_exp(t);
                v = c == 0 ? s + gx : (c == 1 ? s + gy : (c == 2 ? ex * aw : (c == 3 ? ex * ah : s)));
            } else {
                if (c >= 5 && c < pe && p.kind != YAMD_DECODE_V5) {
                    is_pt = true;
                    const bool isx = ((c - 5) & 1) == 0;
                    if (p.kind == YAMD_DECODE_POSE) v = (2.0f * t - 0.5f) + (isx ? gx : gy);
                    else v = t * (isx ? aw : ah) + (isx ? gx : gy);
                } else {
                    const float s = fast_sigmoid(t);
                    const float s2 = 2.0f * s;
                    v = c == 0 ? s2 - 0.5f + gx : (c == 1 ? s2 - 0.5f + gy : (c == 2 ? (s2 * s2) * aw : (c == 3 ? (s2 * s2) * ah : s)));
                }
            }
            if (p.merge) {
                if (is_box) {
                    v = v * fs;
                    v = fminf(fmaxf(v, 0.0f), p.hi);
                } else if (is_pt) {
                    v = v * fs;
                    if (p.clamp_points) v = fminf(fmaxf(v, 0.0f), p.hi);
                }
            }
            dst[e] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// DFL: one workgroup per DPIX pixels of one level.
constexpr int DPIX = 32;

struct DflParams {
    const float *raw;
    float *out;
    int H, W, reg_max, nc, stride, out_cxcywh;
    int unit_conf;                         // column 4 is the constant 1 instead of the best class score
    long long rows_per_image, row_offset;
};

__global__ __launch_bounds__(256) void decode_dfl_kernel(DflParams p)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = p.H * p.W;
    const int CI = 4 * p.reg_max + p.nc;   // input channels per pixel
    const int CO = 5 + p.nc;               // output columns
    float *sraw = sm;                      // DPIX * CI
    float *sout = sm + DPIX * CI;          // DPIX * CO
    const int p0 = blockIdx.x * DPIX;
    const int np = min(DPIX, HW - p0);
    const long long b = blockIdx.y;
    const float *src = p.raw + (b * HW + p0) * (long long)CI;
    const int total = np * CI;
    if (((uintptr_t)src & 15) == 0) {
        const int t4 = total >> 2;
        for (int i = threadIdx.x; i < t4; i += 256) ((float4 *)sraw)[i] = ((const float4 *)src)[i];
        for (int i = (t4 << 2) + threadIdx.x; i < total; i += 256) sraw[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < total; i += 256) sraw[i] = src[i];
    }
    __syncthreads();
    // box sides: thread -> (pixel, side)
    if (threadIdx.x < 4 * DPIX) {
        const int pix = threadIdx.x >> 2, side = threadIdx.x & 3;
        if (pix < np) {
            const float *d = sraw + pix * CI + side * p.reg_max;
            float mx = d[0];
            for (int i = 1; i < p.reg_max; ++i) mx = fmaxf(mx, d[i]);
            float sum = 0.0f;
            for (int i = 0; i < p.reg_max; ++i) sum += expf(d[i] - mx);
            float e = 0.0f;
            for (int i = 0; i < p.reg_max; ++i) e += (expf(d[i] - mx) / sum) * (float)i;
            sout[pix * CO + side] = e;   // ltrb distance in grid units
        }
    }
    // class scores
    for (int i = threadIdx.x; i < np * p.nc; i += 256) {
        const int pix = i / p.nc, c = i % p.nc;
        sout[pix * CO + 5 + c] = 1.0f / (1.0f + expf(-sraw[pix * CI + 4 * p.reg_max + c]));
    }
    __syncthreads();
    if (threadIdx.x < np) {
        const int pix = threadIdx.x, gp = p0 + pix;
        float *o = sout + pix * CO;
        const float ax = (float)(gp % p.W) + 0.5f, ay = (float)(gp / p.W) + 0.5f;
        const float fs = (float)p.stride;
        const float x1 = (ax - o[0]) * fs, y1 = (ay - o[1]) * fs, x2 = (ax + o[2]) * fs, y2 = (ay + o[3]) * fs;
        if (p.out_cxcywh) {
            o[0] = (x1 + x2) / 2.0f;
            o[1] = (y1 + y2) / 2.0f;
            o[2] = x2 - x1;
            o[3] = y2 - y1;
        } else {
            o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
        }
        float best = o[5];
        for (int c = 1; c < p.nc; ++c) best = fmaxf(best, o[5 + c]);
        o[4] = p.unit_conf ? 1.0f : best;
    }
    __syncthreads();
    float *dst = p.out + (b * p.rows_per_image + p.row_offset + p0) * (long long)CO;
    for (int i = threadIdx.x; i < np * CO; i += 256) dst[i] = sout[i];
}

// ---------------------------------------------------------------------------------------------
// Anchor-free ltrb + landmark decode: one workgroup per DPIX pixels of one level.  bins == 1: the four box distances
// are the raw values; bins > 1: each is the expectation of softmax(bins logits) over 0..bins-1 (DFL).
struct LtrbParams {
    const float *raw;
    float *out;
    int H, W, P, nc, stride, out_cxcywh, bins;
    long long rows_per_image, row_offset;
};

__global__ __launch_bounds__(256) void decode_ltrb_kernel(LtrbParams p)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = p.H * p.W;
    const int NB = 4 * p.bins;             // box channels per pixel
    const int CI = NB + 2 * p.P + p.nc;    // input channels per pixel
    const int CO = 5 + 2 * p.P + p.nc;     // output columns
    float *sraw = sm;                      // DPIX * CI
    float *sout = sm + DPIX * CI;          // DPIX * CO
    float *sd = sout + DPIX * CO;          // DPIX * 4 box distances in grid units
    const int p0 = blockIdx.x * DPIX;
    const int np = min(DPIX, HW - p0);
    const long long b = blockIdx.y;
    const float *src = p.raw + (b * HW + p0) * (long long)CI;
    for (int i = threadIdx.x; i < np * CI; i += 256) sraw[i] = src[i];
    __syncthreads();
    if (threadIdx.x < 4 * DPIX) {
        const int pix = threadIdx.x >> 2, side = threadIdx.x & 3;
        if (pix < np) {
            const float *d = sraw + pix * CI + side * p.bins;
            float e = d[0];
            if (p.bins > 1) {
                float mx = d[0];
                for (int i = 1; i < p.bins; ++i) mx = fmaxf(mx, d[i]);
                float sum = 0.0f;
                for (int i = 0; i < p.bins; ++i) sum += expf(d[i] - mx);
                e = 0.0f;
                for (int i = 0; i < p.bins; ++i) e += (expf(d[i] - mx) / sum) * (float)i;
            }
            sd[pix * 4 + side] = e;
        }
    }
    __syncthreads();
    const float fs = (float)p.stride;
    for (int i = threadIdx.x; i < np * CO; i += 256) {
        const int pix = i / CO, c = i % CO, gp = p0 + pix;
        // cell centre in grid units, computed like the reference: ((index + 0.5) * stride) / stride
        const float ax = (((float)(gp % p.W) + 0.5f) * fs) / fs, ay = (((float)(gp / p.W) + 0.5f) * fs) / fs;
        const float *r = sraw + pix * CI + NB;         // [landmarks | cls]
        const float *q = sd + pix * 4;
        float v;
        if (c < 4) {
            const float x1 = (ax - q[0]) * fs, y1 = (ay - q[1]) * fs, x2 = (ax + q[2]) * fs, y2 = (ay + q[3]) * fs;
            if (p.out_cxcywh)
                v = c == 0 ? (x1 + x2) / 2.0f : (c == 1 ? (y1 + y2) / 2.0f : (c == 2 ? x2 - x1 : y2 - y1));
            else
                v = c == 0 ? x1 : (c == 1 ? y1 : (c == 2 ? x2 : y2));
        } else if (c == 4) {
            v = 1.0f;
        } else if (c < 5 + 2 * p.P) {
            const int k = c - 5;
            v = (r[k] + ((k & 1) ? ay : ax)) * fs;
        } else {
            v = 1.0f / (1.0f + expf(-r[c - 5]));
        }
        sout[i] = v;
    }
    __syncthreads();
    float *dst = p.out + (b * p.rows_per_image + p.row_offset + p0) * (long long)CO;
    for (int i = threadIdx.x; i < np * CO; i += 256) dst[i] = sout[i];
}

}  // namespace

extern "C" int yamd_decode_anchor(int kind, const float *raw, int batch, int h, int w, int na, int c,
                                  const float *anchors_wh_host, int stride, int point_num, float img_size, int merge,
                                  int clamp_points, float *out, long long rows_per_image, long long row_offset,
                                  void *stream)
{
    YAMD_REQUIRE(raw && out && anchors_wh_host, "null argument");
    YAMD_REQUIRE(kind >= YAMD_DECODE_V5 && kind <= YAMD_DECODE_FACE, "unknown decode kind %d", kind);
    YAMD_REQUIRE(batch > 0 && batch <= 65535 && h > 0 && w > 0 && na > 0 && na <= 8 && stride > 0, "bad shape");
    const int need = (kind == YAMD_DECODE_POSE) ? 5 + 3 * point_num + 1 : (kind == YAMD_DECODE_FACE ? 5 + 2 * point_num + 1 : 6);
    YAMD_REQUIRE(c >= need && point_num >= 0, "c=%d too small for kind %d with %d points", c, kind, point_num);
    const size_t lds = (size_t)PIX * (na * c + 2) * sizeof(float);
    YAMD_REQUIRE(lds <= 160 * 1024 && (long long)PIX * c * c < (1ll << 22), "na*c=%d too large for the decode tile", na * c);
    {
        const int rc = KernelAttr<&decode_anchor_kernel, 160 * 1024>::ensure();
        if (rc != YAMD_OK) return rc;
    }
    AnchorDecodeParams p;
    p.raw = raw;
    p.out = out;
    p.H = h;
    p.W = w;
    p.na = na;
    p.C = c;
    p.kind = kind;
    p.P = point_num;
    p.c_magic = (unsigned)(((1u << 22) + (unsigned)c - 1u) / (unsigned)c);
    p.stride = stride;
    p.merge = merge ? 1 : 0;
    p.clamp_points = clamp_points ? 1 : 0;
    p.hi = img_size - 1.0f;
    for (int a = 0; a < 8; ++a) {
        p.aw[a] = a < na ? anchors_wh_host[2 * a] / (float)stride : 0.0f;
        p.ah[a] = a < na ? anchors_wh_host[2 * a + 1] / (float)stride : 0.0f;
    }
    if (merge) {
        YAMD_REQUIRE(rows_per_image >= row_offset + (long long)na * h * w && row_offset >= 0, "merged rows out of range");
        p.rows_per_image = rows_per_image;
        p.row_offset = row_offset;
    } else {
        p.rows_per_image = (long long)na * h * w;
        p.row_offset = 0;
    }
    hipLaunchKernelGGL(decode_anchor_kernel, dim3(ceil_div((long long)h * w, PIX), batch), dim3(256), lds,
                       (hipStream_t)stream, p);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

static int decode_dfl_impl(const float *const *raw_host_ptrs, const int *hs, const int *ws, const int *strides,
                           int n_levels, int batch, int reg_max, int nc, int out_cxcywh, int unit_conf, float *out,
                           void *stream)
{
    YAMD_REQUIRE(raw_host_ptrs && hs && ws && strides && out, "null argument");
    YAMD_REQUIRE(n_levels > 0 && n_levels <= 8 && batch > 0 && batch <= 65535 && reg_max > 0 && nc > 0, "bad shape");
    long long rows = 0;
    for (int i = 0; i < n_levels; ++i) rows += (long long)hs[i] * ws[i];
    const size_t lds = (size_t)DPIX * (4 * reg_max + nc + 5 + nc) * sizeof(float);
    YAMD_REQUIRE(lds <= 64 * 1024, "reg_max/nc too large for the DFL tile");
    long long off = 0;
    for (int i = 0; i < n_levels; ++i) {
        YAMD_REQUIRE(raw_host_ptrs[i] != nullptr && hs[i] > 0 && ws[i] > 0, "bad level %d", i);
        DflParams p;
        p.raw = raw_host_ptrs[i];
        p.out = out;
        p.H = hs[i];
        p.W = ws[i];
        p.reg_max = reg_max;
        p.nc = nc;
        p.stride = strides[i];
        p.out_cxcywh = out_cxcywh ? 1 : 0;
        p.unit_conf = unit_conf;
        p.rows_per_image = rows;
        p.row_offset = off;
        hipLaunchKernelGGL(decode_dfl_kernel, dim3(ceil_div((long long)hs[i] * ws[i], DPIX), batch), dim3(256), lds,
                           (hipStream_t)stream, p);
        YAMD_LAUNCH_CHECK();
        off += (long long)hs[i] * ws[i];
    }
    return YAMD_OK;
}

extern "C" int yamd_decode_dfl(const float *const *raw_host_ptrs, const int *hs, const int *ws, const int *strides,
                               int n_levels, int batch, int reg_max, int nc, int out_cxcywh, float *out, void *stream)
{
    return decode_dfl_impl(raw_host_ptrs, hs, ws, strides, n_levels, batch, reg_max, nc, out_cxcywh, 0, out, stream);
}

extern "C" int yamd_decode_dfl_unit_conf(const float *const *raw_host_ptrs, const int *hs, const int *ws,
                                         const int *strides, int n_levels, int batch, int bins, int nc, int out_cxcywh,
                                         float *out, void *stream)
{
    return decode_dfl_impl(raw_host_ptrs, hs, ws, strides, n_levels, batch, bins, nc, out_cxcywh, 1, out, stream);
}

static int decode_ltrb_impl(const float *const *raw_host_ptrs, const int *hs, const int *ws, const int *strides,
                            int n_levels, int batch, int bins, int point_num, int nc, int out_cxcywh, float *out,
                            void *stream)
{
    YAMD_REQUIRE(raw_host_ptrs && hs && ws && strides && out, "null argument");
    YAMD_REQUIRE(n_levels > 0 && n_levels <= 8 && batch > 0 && batch <= 65535 && point_num >= 0 && nc > 0 && bins >= 1,
                 "bad shape");
    long long rows = 0;
    for (int i = 0; i < n_levels; ++i) rows += (long long)hs[i] * ws[i];
    const size_t lds = (size_t)DPIX * (4 * bins + 2 * point_num + nc + 5 + 2 * point_num + nc + 4) * sizeof(float);
    YAMD_REQUIRE(lds <= 64 * 1024, "bins/point_num/nc too large for the decode tile");
    long long off = 0;
    for (int i = 0; i < n_levels; ++i) {
        YAMD_REQUIRE(raw_host_ptrs[i] != nullptr && hs[i] > 0 && ws[i] > 0 && strides[i] > 0, "bad level %d", i);
        LtrbParams p;
        p.raw = raw_host_ptrs[i];
        p.out = out;
        p.H = hs[i];
        p.W = ws[i];
        p.P = point_num;
        p.nc = nc;
        p.stride = strides[i];
        p.out_cxcywh = out_cxcywh ? 1 : 0;
        p.bins = bins;
        p.rows_per_image = rows;
        p.row_offset = off;
        hipLaunchKernelGGL(decode_ltrb_kernel, dim3(ceil_div((long long)hs[i] * ws[i], DPIX), batch), dim3(256), lds,
                           (hipStream_t)stream, p);
        YAMD_LAUNCH_CHECK();
        off += (long long)hs[i] * ws[i];
    }
    return YAMD_OK;
}

extern "C" int yamd_decode_ltrb_points(const float *const *raw_host_ptrs, const int *hs, const int *ws, const int *strides,
                                       int n_levels, int batch, int point_num, int nc, int out_cxcywh, float *out,
                                       void *stream)
{
    return decode_ltrb_impl(raw_host_ptrs, hs, ws, strides, n_levels, batch, 1, point_num, nc, out_cxcywh, out, stream);
}

extern "C" int yamd_decode_dfl_points(const float *const *raw_host_ptrs, const int *hs, const int *ws, const int *strides,
                                      int n_levels, int batch, int bins, int point_num, int nc, int out_cxcywh, float *out,
                                      void *stream)
{
    YAMD_REQUIRE(bins >= 2, "bins=%d: a DFL head has at least two bins per side", bins);
    return decode_ltrb_impl(raw_host_ptrs, hs, ws, strides, n_levels, batch, bins, point_num, nc, out_cxcywh, out, stream);
}
