// Head decode kernels (HBM-bound).  Build with -ffp-contract=off: the arithmetic follows the
// reference's fp32 operation order one rounding at a time.
//
// Both kernels take every pyramid level of a model in ONE launch (a level table in the kernel arguments, workgroup ->
// level by a short search) and have the same shape: a workgroup copies the raw logits of a run of consecutive pixels
// of one image (contiguous in NHWC) into LDS with 16-byte loads, then every thread produces elements t, t+256, ... of
// the workgroup's contiguous run of output floats, four at a time with one 16-byte store where the run is 16-byte
// aligned.  Both sides of the kernel stream whole cache lines and no lane idles.  The pixels per workgroup shrink as
// the rows get wider so that a compute unit always holds 6 to 8 workgroups.
#include "yamd_common.h"

#include <cmath>
#include <cstring>

using namespace yamd;

namespace {

constexpr int MAX_LEVELS = 8;

// Hardware exp2 / reciprocal (about 1 ulp each): sigmoid and exp within a few 1e-7 relative of the correctly
// rounded functions for the logit range of a detector head, far inside the 1e-5 the decode is held to.
__device__ __forceinline__ float fast_exp(float t) { return __builtin_amdgcn_exp2f(t * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_sigmoid(float t) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-t)); }

struct Level {
    const float *raw;
    int H, W, stride;
    int first_block;          // first workgroup (blockIdx.x) of this level
    long long row_offset;     // first output row of this level inside an image
    float aw[4], ah[4];       // anchor / stride, fp32 division as torch does it (anchor heads)
};

// copies `total` floats from global to LDS, 16 bytes per lane where the source is aligned.  The logits are read once
// and the decoded rows written once: both go past the caches (non-temporal), which is worth ~5 % on a streaming copy.
__device__ __forceinline__ void stage_rows(float *dst, const float *src, int total)
{
    if (((uintptr_t)src & 15) == 0) {
        const int t4 = total >> 2;
        for (int i = threadIdx.x; i < t4; i += 256) ((f32x4 *)dst)[i] = __builtin_nontemporal_load((const f32x4 *)src + i);
        for (int i = (t4 << 2) + threadIdx.x; i < total; i += 256) dst[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < total; i += 256) dst[i] = src[i];
    }
}

// writes run[0 .. n) = f(e): four elements per lane and one 16-byte store where dst is 16-byte aligned
template <class F>
__device__ __forceinline__ void store_run(float *dst, int n, F &&f)
{
    if (((uintptr_t)dst & 15) == 0) {
        const int n4 = n >> 2;
        for (int q = threadIdx.x; q < n4; q += 256) {
            float a, b, c, d;
            f(4 * q, a, b, c, d);
            const f32x4 v = {a, b, c, d};
            __builtin_nontemporal_store(v, (f32x4 *)dst + q);
        }
        for (int e = (n4 << 2) + threadIdx.x; e < n; e += 256) {
            float a, b, c, d;
            f(e, a, b, c, d);      // only the first value is in range
            dst[e] = a;
        }
    } else {
        for (int e = threadIdx.x; e < n; e += 256) {
            float a, b, c, d;
            f(e, a, b, c, d);
            dst[e] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Anchor heads (v3/v4, v5/v7, pose, face).
struct AnchorDecodeParams {
    float *out;
    int n_levels, na, C, kind, P, merge, clamp_points, pix;
    int rstr;                 // floats per LDS pixel row: na*C, or na*C + 1 where na*C shares a factor > 2 with the 32 banks
    int nchunk, cw;           // the C columns of a row are transformed in nchunk pieces of cw columns
    unsigned c_magic;         // ceil(2^22 / C): e / C == (e * c_magic) >> 22 for e < pix * C (checked on the host)
    unsigned nac_magic;       // the same for na*C
    float hi;                 // img_size - 1
    long long rows_per_image;
    FilterTargets ft;         // FILTER instantiation only
    Level lv[MAX_LEVELS];
};

__device__ __forceinline__ unsigned order_bits(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// FILTER: the predict path's form.  Every decoded row is put through the confidence filter of the NMS front end while it
// is still in LDS (utils/torch_utils.py:128-136: first-maximum class, fp32 product obj * class score >= threshold); rows
// that pass are entered into the NMS candidate lists (key = score bits and anchor index, ticket inside the (image, class)
// bucket) and written to `out`, rows that fail are not written: the decoded tensor is neither written nor read back in
// full.  The values are those of the plain kernel bit for bit (same code), so the lists equal the separate filter's.
template <bool FILTER>
__global__ __launch_bounds__(256) void decode_anchor_kernel(AnchorDecodeParams p)
{
    extern __shared__ __attribute__((aligned(16))) float sraw[];     // pix rows of rstr floats (na*C logits each)
    unsigned *const hist = (unsigned *)(sraw + p.pix * p.rstr);      // FILTER: nc bucket counts of this workgroup,
    unsigned *const cbase = hist + p.ft.nc;                          // nc bucket bases,
    unsigned *const misc = cbase + p.ft.nc;                          // [0] survivors of the workgroup, [1] their base in the image,
    unsigned char *const sflag = (unsigned char *)(misc + 4);        // one pass flag per row (anchor-major)
    if constexpr (FILTER) {
        for (int c = threadIdx.x; c < p.ft.nc; c += 256) hist[c] = 0u;
        if (threadIdx.x == 0) misc[0] = 0u;
    }
    int li = 0;
#pragma unroll
    for (int i = 1; i < MAX_LEVELS; ++i)
        if (i < p.n_levels && (int)blockIdx.x >= p.lv[i].first_block) li = i;
    const Level &L = p.lv[li];
    const int HW = L.H * L.W;
    const int naC = p.na * p.C, RSTR = p.rstr;
    const int p0 = ((int)blockIdx.x - L.first_block) * p.pix;
    const int np = min(p.pix, HW - p0);
    const long long b = blockIdx.y;
    const float *src = L.raw + (b * HW + p0) * (long long)naC;
    if (RSTR == naC) {
        stage_rows(sraw, src, np * naC);
    } else {                                                          // padded rows (row length shares a factor with the banks)
        for (int i = threadIdx.x; i < np * naC; i += 256) {
            const int pix = (int)(((unsigned)i * p.nac_magic) >> 22);
            sraw[pix * RSTR + (i - pix * naC)] = src[i];
        }
    }
    __syncthreads();

    // Phase A, in place: task = (column chunk, anchor, pixel), the pixel varying fastest, so the lanes of a wavefront
    // share the anchor and walk the same columns: the per-column case analysis is uniform, every lane evaluates one
    // sigmoid per value and nothing else, and the grid position of the pixel is computed once per task.
    const float fs = (float)L.stride;
    const int pe = 5 + 2 * p.P;
    const int C = p.C, W = L.W;
    const int ntask = p.nchunk * p.na * np;
    for (int task = threadIdx.x; task < ntask; task += 256) {
        const int rest = task / np, pix = task - rest * np;
        const int chunk = rest / p.na, a = rest - chunk * p.na;
        const int c0 = chunk * p.cw, c1 = min(C, c0 + p.cw);
        const int gp = p0 + pix, gyi = gp / W;
        const float gx = (float)(gp - gyi * W), gy = (float)gyi;
        const float aw = L.aw[a], ah = L.ah[a];
        float *row = sraw + pix * RSTR + a * C;
        for (int c = c0; c < c1; ++c) {
            const float t = row[c];
            float v;
            bool scale = c < 4, clamp = c < 4;
            if (p.kind == YAMD_DECODE_V3) {
                const float s = fast_sigmoid(t);
                if (c >= 2 && c < 4) v = fast_exp(t) * (c == 2 ? aw : ah);
                else v = c == 0 ? s + gx : (c == 1 ? s + gy : s);
            } else if (c >= 5 && c < pe && p.kind != YAMD_DECODE_V5) {
                const bool isx = ((c - 5) & 1) == 0;
                if (p.kind == YAMD_DECODE_POSE) v = (2.0f * t - 0.5f) + (isx ? gx : gy);
                else v = t * (isx ? aw : ah) + (isx ? gx : gy);
                scale = true;
                clamp = p.clamp_points != 0;
            } else {
                const float s = fast_sigmoid(t);
                const float s2 = 2.0f * s;
                v = c == 0 ? s2 - 0.5f + gx : (c == 1 ? s2 - 0.5f + gy : (c == 2 ? (s2 * s2) * aw : (c == 3 ? (s2 * s2) * ah : s)));
            }
            if (p.merge && scale) {
                v = v * fs;
                if (clamp) v = fminf(fmaxf(v, 0.0f), p.hi);
            }
            row[c] = v;
        }
    }
    __syncthreads();

    if constexpr (FILTER) {
        // one row per thread: row r = (anchor a, pixel) with the pixel varying fastest, as the output orders them
        const int nrow = p.na * np, tid = threadIdx.x;
        bool pass = false;
        unsigned cls = 0u, lt = 0u, lp = 0u;
        unsigned long long key = 0ull;
        if (tid < nrow) {
            const int a = tid / np, pix = tid - a * np;
            const float *row = sraw + pix * RSTR + a * C;
            const float *sc = row + 5 + p.ft.n_extra;
            float best = sc[0];
            int bi = 0;
            for (int c = 1; c < p.ft.nc; ++c) {
                const float v = sc[c];
                if (v > best) { best = v; bi = c; }
            }
            const float score = row[4] * best;
            if (score >= p.ft.conf_thr) {
                pass = true;
                cls = (unsigned)bi;
                const unsigned idx = (unsigned)(L.row_offset + (long long)a * HW + p0 + pix);      // row inside the image
                key = ((unsigned long long)order_bits(score) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
                lt = atomicAdd(&hist[cls], 1u);
                lp = atomicAdd(&misc[0], 1u);
            }
            sflag[tid] = pass ? 1 : 0;
        }
        __syncthreads();
        for (int c = tid; c < p.ft.nc; c += 256)
            if (hist[c]) cbase[c] = atomicAdd(&p.ft.cls_count[b * p.ft.nc + c], hist[c]);
        if (tid == 0 && misc[0]) misc[1] = atomicAdd(&p.ft.img_count[b], misc[0]);
        __syncthreads();
        if (pass) {
            const long long pos = b * p.ft.a + misc[1] + lp;
            p.ft.tmp_key[pos] = key;
            p.ft.tmp_meta[pos] = ((unsigned long long)cls << 32) | (unsigned long long)(cbase[cls] + lt);
        }
        if (misc[0] == 0u) return;                                   // nothing of this workgroup's rows survives
    }

    // Phase B: the np rows of one anchor are one contiguous run of np*C floats in the output.
    const int run = np * C;
    for (int a = 0; a < p.na; ++a) {
        const long long first_row = p.merge ? b * p.rows_per_image + L.row_offset + (long long)a * HW + p0
                                            : (b * p.na + a) * (long long)HW + p0;
        float *dst = p.out + first_row * C;
        const float *sa = sraw + a * C;
        if constexpr (FILTER) {
            // as store_run, but only the elements of rows that passed are stored (16 bytes at once where four
            // consecutive elements all belong to such rows)
            const unsigned char *fl = sflag + a * np;
            const bool vec = ((uintptr_t)dst & 15) == 0;
            const int n4 = vec ? (run >> 2) : 0;
            for (int q = threadIdx.x; q < n4; q += 256) {
                const int e = 4 * q;
                int pix = (int)(((unsigned)e * p.c_magic) >> 22), c = e - pix * C;
                float v[4];
                unsigned ok = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = sa[pix * RSTR + c];
                    ok |= (unsigned)fl[pix] << j;
                    if (++c == C) { c = 0; pix = min(pix + 1, np - 1); }
                }
                if (ok == 15u) {
                    const f32x4 vv = {v[0], v[1], v[2], v[3]};
                    *((f32x4 *)dst + q) = vv;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((ok >> j) & 1u) dst[e + j] = v[j];
                }
            }
            for (int e = (n4 << 2) + threadIdx.x; e < run; e += 256) {
                const int pix = (int)(((unsigned)e * p.c_magic) >> 22), c = e - pix * C;
                if (fl[pix]) dst[e] = sa[pix * RSTR + c];
            }
            continue;
        }
        store_run(dst, run, [&](int e, float &v0, float &v1, float &v2, float &v3) {
            int pix = (int)(((unsigned)e * p.c_magic) >> 22);      // e / C
            int c = e - pix * C;
            const float *r = sa + pix * RSTR;
            v0 = r[c];
            // the following three elements (positions past the end of the run read valid LDS and are dropped)
            if (++c == C) { c = 0; r = sa + min(pix + 1, np - 1) * RSTR; ++pix; }
            v1 = r[c];
            if (++c == C) { c = 0; r = sa + min(pix + 1, np - 1) * RSTR; ++pix; }
            v2 = r[c];
            if (++c == C) { c = 0; r = sa + min(pix + 1, np - 1) * RSTR; ++pix; }
            v3 = r[c];
        });
    }
}

// ---------------------------------------------------------------------------------------------
// Anchor-free heads: per pixel [4 x bins box logits | 2P point values | ncs score logits] ->
// [box(4) | conf | 2P points | ncs scores].  bins == 1: the four box distances are the raw values; bins > 1: each is
// the expectation of softmax(bins logits) over 0..bins-1 (DFL).
struct FreeDecodeParams {
    float *out;
    int n_levels, bins, P, ncs, out_cxcywh, pix;
    int unit_conf;            // column 4: 1 = the constant 1.0, 0 = the best score
    int center_via_stride;    // cell centre as ((i + 0.5) * stride) / stride (YOLOv6) instead of i + 0.5 (YOLOv8)
    unsigned co_magic;        // ceil(2^22 / CO)
    int nchunk, cw;           // the point and score channels of a pixel are transformed in nchunk pieces of cw channels
    long long rows_per_image;
    Level lv[MAX_LEVELS];
};

__global__ __launch_bounds__(256) void decode_free_kernel(FreeDecodeParams p)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int li = 0;
#pragma unroll
    for (int i = 1; i < MAX_LEVELS; ++i)
        if (i < p.n_levels && (int)blockIdx.x >= p.lv[i].first_block) li = i;
    const Level &L = p.lv[li];
    const int HW = L.H * L.W;
    const int NB = 4 * p.bins;
    const int NT = 2 * p.P + p.ncs;        // point and score channels: transformed in place
    const int CI = NB + NT;                // input channels per pixel
    const int CO = 5 + NT;                 // output columns
    float *sraw = sm;                      // pix * CI
    float *sbox = sm + p.pix * CI;         // pix * 5: box distances (grid units), then the finished box and confidence
    const int p0 = ((int)blockIdx.x - L.first_block) * p.pix;
    const int np = min(p.pix, HW - p0);
    const long long b = blockIdx.y;
    stage_rows(sraw, L.raw + (b * HW + p0) * (long long)CI, np * CI);
    __syncthreads();

    // phase 1: four lanes per task.  Tasks [0, 4*np): box side (pixel, side) -> distance; tasks [4*np, 5*np): best
    // score logit of a pixel (sigmoid is monotonic and evaluated once per input below, so the best score is the
    // sigmoid of the best logit, bit for bit).  Every lane of a wave runs the same number of rounds (the shuffles
    // need their partners); lanes beyond the last task recompute task 0 and store nothing.
    const int sub = threadIdx.x & 3;
    const int ntask = p.unit_conf ? 4 * np : 5 * np;
    for (int task = threadIdx.x >> 2; task < ((ntask + 63) & ~63); task += 64) {
        const bool live = task < ntask;
        const int tk = live ? task : 0;
        if (tk < 4 * np) {
            const float *d = sraw + (tk >> 2) * CI + (tk & 3) * p.bins;
            float e;
            if (p.bins > 1) {
                float mx = -INFINITY;
                for (int i = sub; i < p.bins; i += 4) mx = fmaxf(mx, d[i]);
                mx = fmaxf(mx, __shfl_xor(mx, 1));
                mx = fmaxf(mx, __shfl_xor(mx, 2));
                float num = 0.0f, den = 0.0f;
                for (int i = sub; i < p.bins; i += 4) {
                    const float w = fast_exp(d[i] - mx);
                    den += w;
                    num += w * (float)i;
                }
                num += __shfl_xor(num, 1);
                den += __shfl_xor(den, 1);
                num += __shfl_xor(num, 2);
                den += __shfl_xor(den, 2);
                e = num / den;
            } else {
                e = d[0];
            }
            if (live && sub == 0) sbox[(tk >> 2) * 5 + (tk & 3)] = e;
        } else {
            const float *s = sraw + (tk - 4 * np) * CI + NB + 2 * p.P;
            float mx = -INFINITY;
            for (int i = sub; i < p.ncs; i += 4) mx = fmaxf(mx, s[i]);
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            if (live && sub == 0) sbox[(tk - 4 * np) * 5 + 4] = mx;
        }
    }
    __syncthreads();

    // phase 2, in place: task = (column chunk, pixel), the pixel varying fastest, so the lanes of a wavefront walk the
    // same columns (uniform case analysis); chunk 0 also finishes the box of its pixel.
    const float fs = (float)L.stride;
    const int W = L.W;
    const int ntask2 = p.nchunk * np;
    for (int task = threadIdx.x; task < ntask2; task += 256) {
        const int chunk = task / np, pix = task - chunk * np;
        const int gp = p0 + pix, gyi = gp / W;
        float ax = (float)(gp - gyi * W) + 0.5f, ay = (float)gyi + 0.5f;
        if (p.center_via_stride) {
            ax = (ax * fs) / fs;
            ay = (ay * fs) / fs;
        }
        if (chunk == 0) {
            float *q = sbox + pix * 5;
            const float x1 = (ax - q[0]) * fs, y1 = (ay - q[1]) * fs, x2 = (ax + q[2]) * fs, y2 = (ay + q[3]) * fs;
            if (p.out_cxcywh) {
                q[0] = (x1 + x2) / 2.0f; q[1] = (y1 + y2) / 2.0f; q[2] = x2 - x1; q[3] = y2 - y1;
            } else {
                q[0] = x1; q[1] = y1; q[2] = x2; q[3] = y2;
            }
            q[4] = p.unit_conf ? 1.0f : fast_sigmoid(q[4]);
        }
        float *r = sraw + pix * CI + NB;              // [points | scores]
        const int k1 = min(NT, (chunk + 1) * p.cw);
        for (int k = chunk * p.cw; k < k1; ++k) {
            const float t = r[k];
            r[k] = k < 2 * p.P ? (t + ((k & 1) ? ay : ax)) * fs : fast_sigmoid(t);
        }
    }
    __syncthreads();

    float *dst = p.out + (b * p.rows_per_image + L.row_offset + p0) * (long long)CO;
    auto at = [&](int pix, int c) -> float { return c < 5 ? sbox[pix * 5 + c] : sraw[pix * CI + NB + (c - 5)]; };
    store_run(dst, np * CO, [&](int e, float &v0, float &v1, float &v2, float &v3) {
        int pix = (int)(((unsigned)e * p.co_magic) >> 22);      // e / CO
        int c = e - pix * CO;
        v0 = at(pix, c);
        if (++c == CO) { c = 0; pix = min(pix + 1, np - 1); }
        v1 = at(pix, c);
        if (++c == CO) { c = 0; pix = min(pix + 1, np - 1); }
        v2 = at(pix, c);
        if (++c == CO) { c = 0; pix = min(pix + 1, np - 1); }
        v3 = at(pix, c);
    });
}

// pixels per workgroup: the widest tile of at most 24 KB of LDS, so that 6 to 8 workgroups share a compute unit
int pick_pix(int floats_per_pixel)
{
    int pix = 64;
    while (pix > 8 && (size_t)pix * floats_per_pixel * sizeof(float) > 24 * 1024) pix >>= 1;
    return pix;
}

int fill_levels(Level *lv, const float *const *raws, const int *hs, const int *ws, const int *strides, int n_levels, int pix,
                long long *total_rows, int *total_blocks, int rows_per_pixel)
{
    long long rows = 0;
    long long blocks = 0;
    for (int i = 0; i < n_levels; ++i) {
        YAMD_REQUIRE(raws[i] != nullptr && hs[i] > 0 && ws[i] > 0 && strides[i] > 0, "bad level %d", i);
        YAMD_REQUIRE((long long)hs[i] * ws[i] < (1ll << 28), "level %d too large", i);
        lv[i].raw = raws[i];
        lv[i].H = hs[i];
        lv[i].W = ws[i];
        lv[i].stride = strides[i];
        lv[i].first_block = (int)blocks;
        lv[i].row_offset = rows;
        rows += (long long)rows_per_pixel * hs[i] * ws[i];
        blocks += ceil_div((long long)hs[i] * ws[i], pix);
        YAMD_REQUIRE(blocks < (1ll << 31), "too many decode workgroups");
    }
    *total_rows = rows;
    *total_blocks = (int)blocks;
    return YAMD_OK;
}

int decode_anchor_impl(int kind, const float *const *raws, const int *hs, const int *ws, const int *strides,
                       const float *anchors_wh_host, int n_levels, int batch, int na, int c, int point_num, float img_size,
                       int merge, int clamp_points, float *out, long long rows_per_image, long long row_offset, void *stream,
                       const FilterTargets *ft = nullptr)
{
    YAMD_REQUIRE(raws && hs && ws && strides && out && anchors_wh_host, "null argument");
    YAMD_REQUIRE(kind >= YAMD_DECODE_V5 && kind <= YAMD_DECODE_FACE, "unknown decode kind %d", kind);
    YAMD_REQUIRE(n_levels > 0 && n_levels <= MAX_LEVELS && batch > 0 && batch <= 65535 && na > 0 && na <= 4, "bad shape");
    const int need = (kind == YAMD_DECODE_POSE) ? 5 + 3 * point_num + 1 : (kind == YAMD_DECODE_FACE ? 5 + 2 * point_num + 1 : 6);
    YAMD_REQUIRE(c >= need && point_num >= 0, "c=%d too small for kind %d with %d points", c, kind, point_num);
    AnchorDecodeParams p;
    p.pix = pick_pix(na * c + 1);
    {
        int g = na * c, t = 32;                       // gcd(na*c, 32)
        while (t) { const int r = g % t; g = t; t = r; }
        p.rstr = g > 2 ? na * c + 1 : na * c;
    }
    p.nchunk = 1;
    while (p.nchunk < 8 && p.nchunk * na * p.pix < 768 && c / (p.nchunk * 2) >= 4) p.nchunk *= 2;
    p.cw = (c + p.nchunk - 1) / p.nchunk;
    size_t lds = (size_t)p.pix * p.rstr * sizeof(float);
    memset(&p.ft, 0, sizeof(p.ft));
    if (ft) {
        YAMD_REQUIRE(merge && ft->nc > 0 && ft->nc == c - 5 - ft->n_extra && na * p.pix <= 256, "filter targets do not match the rows");
        lds += (size_t)(2 * ft->nc + 4) * sizeof(unsigned) + 256;
        if (lds > 64 * 1024) {
            set_error("%d classes do not fit the fused decode + filter tile", ft->nc);
            return YAMD_ENOSPC;
        }
        p.ft = *ft;
    }
    YAMD_REQUIRE(lds <= 64 * 1024 && (long long)(p.pix * c + 4) * c < (1ll << 22) &&
                     (long long)(p.pix * na * c) * (na * c) < (1ll << 22),
                 "na*c=%d too large for the decode tile", na * c);
    p.nac_magic = (unsigned)(((1u << 22) + (unsigned)(na * c) - 1u) / (unsigned)(na * c));
    long long rows = 0;
    int blocks = 0;
    const int rc = fill_levels(p.lv, raws, hs, ws, strides, n_levels, p.pix, &rows, &blocks, na);
    if (rc != YAMD_OK) return rc;
    for (int i = 0; i < n_levels; ++i)
        for (int a = 0; a < 4; ++a) {
            p.lv[i].aw[a] = a < na ? anchors_wh_host[(i * na + a) * 2] / (float)strides[i] : 0.0f;
            p.lv[i].ah[a] = a < na ? anchors_wh_host[(i * na + a) * 2 + 1] / (float)strides[i] : 0.0f;
        }
    p.out = out;
    p.n_levels = n_levels;
    p.na = na;
    p.C = c;
    p.kind = kind;
    p.P = point_num;
    p.c_magic = (unsigned)(((1u << 22) + (unsigned)c - 1u) / (unsigned)c);
    p.merge = merge ? 1 : 0;
    p.clamp_points = clamp_points ? 1 : 0;
    p.hi = img_size - 1.0f;
    if (merge) {
        YAMD_REQUIRE(row_offset >= 0 && rows_per_image >= row_offset + rows, "merged rows out of range");
        p.rows_per_image = rows_per_image;
        for (int i = 0; i < n_levels; ++i) p.lv[i].row_offset += row_offset;
    } else {
        YAMD_REQUIRE(n_levels == 1, "the head-module form takes one level");
        p.rows_per_image = rows;
    }
    if (ft) {
        YAMD_REQUIRE(rows == ft->a, "filter targets sized for %d rows per image, the levels hold %lld", ft->a, rows);
        hipLaunchKernelGGL(decode_anchor_kernel<true>, dim3(blocks, batch), dim3(256), lds, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL(decode_anchor_kernel<false>, dim3(blocks, batch), dim3(256), lds, (hipStream_t)stream, p);
    }
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

int decode_free_impl(const float *const *raws, const int *hs, const int *ws, const int *strides, int n_levels, int batch,
                     int bins, int point_num, int ncs, int out_cxcywh, int unit_conf, int center_via_stride, float *out,
                     void *stream)
{
    YAMD_REQUIRE(raws && hs && ws && strides && out, "null argument");
    YAMD_REQUIRE(n_levels > 0 && n_levels <= MAX_LEVELS && batch > 0 && batch <= 65535 && point_num >= 0 && ncs > 0 && bins >= 1,
                 "bad shape");
    FreeDecodeParams p;
    const int CI = 4 * bins + 2 * point_num + ncs, CO = 5 + 2 * point_num + ncs;
    p.pix = pick_pix(CI + 5);
    const size_t lds = (size_t)p.pix * (CI + 5) * sizeof(float);
    YAMD_REQUIRE(lds <= 64 * 1024 && (long long)(p.pix * CO + 4) * CO < (1ll << 22), "bins/point_num/nc too large for the decode tile");
    long long rows = 0;
    int blocks = 0;
    const int rc = fill_levels(p.lv, raws, hs, ws, strides, n_levels, p.pix, &rows, &blocks, 1);
    if (rc != YAMD_OK) return rc;
    p.out = out;
    p.n_levels = n_levels;
    p.bins = bins;
    p.P = point_num;
    p.ncs = ncs;
    p.out_cxcywh = out_cxcywh ? 1 : 0;
    p.unit_conf = unit_conf;
    p.center_via_stride = center_via_stride;
    p.co_magic = (unsigned)(((1u << 22) + (unsigned)CO - 1u) / (unsigned)CO);
    p.rows_per_image = rows;
    {
        const int nt = 2 * point_num + ncs;
        p.nchunk = 1;
        while (p.nchunk < 16 && p.nchunk * p.pix < 768 && nt / (p.nchunk * 2) >= 2) p.nchunk *= 2;
        p.cw = (nt + p.nchunk - 1) / p.nchunk;
    }
    hipLaunchKernelGGL(decode_free_kernel, dim3(blocks, batch), dim3(256), lds, (hipStream_t)stream, p);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

}  // namespace

extern "C" int yamd_decode_anchor(int kind, const float *raw, int batch, int h, int w, int na, int c,
                                  const float *anchors_wh_host, int stride, int point_num, float img_size, int merge,
                                  int clamp_points, float *out, long long rows_per_image, long long row_offset,
                                  void *stream)
{
    return decode_anchor_impl(kind, &raw, &h, &w, &stride, anchors_wh_host, 1, batch, na, c, point_num, img_size, merge,
                              clamp_points, out, rows_per_image, row_offset, stream);
}

extern "C" int yamd_decode_anchor_levels(int kind, const float *const *raw_host_ptrs, const int *hs_host, const int *ws_host,
                                         const int *strides_host, const float *anchors_wh_host, int n_levels, int batch,
                                         int na, int c, int point_num, float img_size, int clamp_points, float *out,
                                         void *stream)
{
    YAMD_REQUIRE(hs_host && ws_host && n_levels > 0 && n_levels <= MAX_LEVELS && na > 0, "bad shape");
    long long rows = 0;
    for (int i = 0; i < n_levels; ++i) rows += (long long)na * hs_host[i] * ws_host[i];
    return decode_anchor_impl(kind, raw_host_ptrs, hs_host, ws_host, strides_host, anchors_wh_host, n_levels, batch, na, c,
                              point_num, img_size, 1, clamp_points, out, rows, 0, stream);
}

namespace yamd {
int decode_anchor_levels_filtered(int kind, const float *const *raws, const int *hs, const int *ws, const int *strides,
                                  const float *anchors_wh_host, int n_levels, int batch, int na, int c, int point_num, float img_size,
                                  int clamp_points, float *out, const FilterTargets &ft, void *stream)
{
    YAMD_REQUIRE(hs && ws && n_levels > 0 && n_levels <= MAX_LEVELS && na > 0, "bad shape");
    long long rows = 0;
    for (int i = 0; i < n_levels; ++i) rows += (long long)na * hs[i] * ws[i];
    return decode_anchor_impl(kind, raws, hs, ws, strides, anchors_wh_host, n_levels, batch, na, c, point_num, img_size, 1,
                              clamp_points, out, rows, 0, stream, &ft);
}
}  // namespace yamd

extern "C" int yamd_decode_dfl(const float *const *raw_host_ptrs, const int *hs, const int *ws, const int *strides,
                               int n_levels, int batch, int reg_max, int nc, int out_cxcywh, float *out, void *stream)
{
    YAMD_REQUIRE(reg_max >= 2, "reg_max=%d: a DFL head has at least two bins per side", reg_max);
    return decode_free_impl(raw_host_ptrs, hs, ws, strides, n_levels, batch, reg_max, 0, nc, out_cxcywh, 0, 0, out, stream);
}

extern "C" int yamd_decode_dfl_unit_conf(const float *const *raw_host_ptrs, const int *hs, const int *ws,
                                         const int *strides, int n_levels, int batch, int bins, int nc, int out_cxcywh,
                                         float *out, void *stream)
{
    YAMD_REQUIRE(bins >= 2, "bins=%d: a DFL head has at least two bins per side", bins);
    return decode_free_impl(raw_host_ptrs, hs, ws, strides, n_levels, batch, bins, 0, nc, out_cxcywh, 1, 0, out, stream);
}

extern "C" int yamd_decode_ltrb_points(const float *const *raw_host_ptrs, const int *hs, const int *ws, const int *strides,
                                       int n_levels, int batch, int point_num, int nc, int out_cxcywh, float *out,
                                       void *stream)
{
    return decode_free_impl(raw_host_ptrs, hs, ws, strides, n_levels, batch, 1, point_num, nc, out_cxcywh, 1, 1, out, stream);
}

extern "C" int yamd_decode_dfl_points(const float *const *raw_host_ptrs, const int *hs, const int *ws, const int *strides,
                                      int n_levels, int batch, int bins, int point_num, int nc, int out_cxcywh, float *out,
                                      void *stream)
{
    YAMD_REQUIRE(bins >= 2, "bins=%d: a DFL head has at least two bins per side", bins);
    return decode_free_impl(raw_host_ptrs, hs, ws, strides, n_levels, batch, bins, point_num, nc, out_cxcywh, 1, 1, out, stream);
}
