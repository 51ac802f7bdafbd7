// Batched confidence filter + per-class greedy NMS.  Build with -ffp-contract=off.
//
// Pipeline (all images of the batch in every launch, nothing returns to the host):
//   1. filter   one pass over (B,A,C): rows staged through LDS with coalesced loads; per row the
//               box is converted/clamped, the best class found, score = obj*cls compared with the
//               threshold.  Survivors get a 64-bit key (order-preserving score bits << 32 |
//               ~anchor index, so "larger key" == "earlier in a stable descending sort") and a
//               ticket inside their (image, class) bucket: an LDS histogram per workgroup and one
//               global atomic per (workgroup, class).
//   2. scan     exclusive prefix of the class counts per image.
//   3. bucket   scatter keys into their class buckets (order inside a bucket is arbitrary).
//   4. segment  one workgroup per (image, class): rank the bucket by counting (keys are unique, so
//               the rank IS the position in the stable descending sort), write keys and class-offset
//               boxes in sorted order, then run the greedy sweep in tiles of 64 boxes: wave 0
//               resolves the tile with ballots (bit i of the mask = box i still alive), the kept
//               boxes are broadcast through LDS and every wave clears the later boxes they suppress
//               (one ballot -> one 64-bit word of the removed-bitmask in LDS per wave step).
//               Boxes of different classes never interact: their offset boxes are disjoint because
//               img_size <= 4096, which is what adding cls*4096 achieves in the reference.
//   5. gather   kept rows of an image are merged across classes into global descending-score order
//               by binary-searching each class's (already sorted) kept list, and written out.
//
// The IoU arithmetic is the torchvision CPU kernel's, one fp32 rounding per operation, on the
// offset boxes; 'ovr > thr(double)' is evaluated as 'ovr > largest float <= thr', which is the same
// predicate for every float ovr.
#include "yamd_common.h"

#include <cmath>

using namespace yamd;

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

struct NmsParams {
    const float *pred;
    int B, A, C, in_cxcywh, n_extra, nc, W, max_out;
    float hi, conf_thr, iou_thr;
    u64 *tmp_key, *tmp_meta, *bkey, *skey, *kseg;
    float4 *sbox;
    u32 *img_count, *cls_count, *cls_off, *kcnt;
    float *out_rows;
    int *out_index, *out_counts;
};

__device__ __forceinline__ u32 order_bits(float f)
{
    const u32 u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float clamp_box(float v, float hi) { return fminf(fmaxf(v, 0.0f), hi); }

// box columns of one row -> clamped xyxy (utils/torch_utils.py:117-121)
__device__ __forceinline__ void row_box(const float *row, int in_cxcywh, float hi, float &x1, float &y1, float &x2,
                                        float &y2)
{
    if (in_cxcywh) {
        const float hw = 0.5f * row[2], hh = 0.5f * row[3];
        x1 = row[0] - hw; y1 = row[1] - hh; x2 = row[0] + hw; y2 = row[1] + hh;
    } else {
        x1 = row[0]; y1 = row[1]; x2 = row[2]; y2 = row[3];
    }
    x1 = clamp_box(x1, hi); y1 = clamp_box(y1, hi); x2 = clamp_box(x2, hi); y2 = clamp_box(y2, hi);
}

// first maximal class (torch.max)
__device__ __forceinline__ void row_best(const float *cls, int nc, float &best, int &bi)
{
    best = cls[0];
    bi = 0;
    for (int c = 1; c < nc; ++c) {
        const float v = cls[c];
        if (v > best) { best = v; bi = c; }
    }
}

constexpr int FROWS = 256;

__global__ __launch_bounds__(256) void nms_filter_kernel(NmsParams p)
{
    extern __shared__ __attribute__((aligned(16))) float srow[];
    const int RSTR = p.C | 1;
    u32 *hist = (u32 *)(srow + FROWS * RSTR);   // nc
    u32 *cbase = hist + p.nc;                   // nc
    u32 *misc = cbase + p.nc;                   // [0] block total, [1] image base
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * FROWS;
    const int nrows = min(FROWS, p.A - r0);
    const long long b = blockIdx.y;
    for (int c = tid; c < p.nc; c += 256) hist[c] = 0;
    if (tid == 0) misc[0] = 0;
    {
        const float *src = p.pred + (b * p.A + r0) * (long long)p.C;
        const int total = nrows * p.C;
        const float inv = 1.0f / (float)p.C;
        for (int i = tid; i < total; i += 256) {
            int q = (int)((float)i * inv);
            int r = i - q * p.C;
            if (r >= p.C) { ++q; r -= p.C; }
            if (r < 0) { --q; r += p.C; }
            srow[q * RSTR + r] = src[i];
        }
    }
    __syncthreads();
    bool pass = false;
    u32 cls = 0, lt = 0, lp = 0;
    u64 key = 0;
    if (tid < nrows) {
        const float *row = srow + tid * RSTR;
        float best;
        int bi;
        row_best(row + 5 + p.n_extra, p.nc, best, bi);
        const float sc = row[4] * best;
        if (sc >= p.conf_thr) {
            pass = true;
            cls = (u32)bi;
            key = ((u64)order_bits(sc) << 32) | (u64)(0xFFFFFFFFu - (u32)(r0 + tid));
            lt = atomicAdd(&hist[cls], 1u);
            lp = atomicAdd(&misc[0], 1u);
        }
    }
    __syncthreads();
    for (int c = tid; c < p.nc; c += 256)
        if (hist[c]) cbase[c] = atomicAdd(&p.cls_count[b * p.nc + c], hist[c]);
    if (tid == 0 && misc[0]) misc[1] = atomicAdd(&p.img_count[b], misc[0]);
    __syncthreads();
    if (pass) {
        const long long pos = b * p.A + misc[1] + lp;
        p.tmp_key[pos] = key;
        p.tmp_meta[pos] = ((u64)cls << 32) | (u64)(cbase[cls] + lt);
    }
}

__global__ void nms_scan_kernel(NmsParams p)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        u32 s = 0;
        for (int c = 0; c < p.nc; ++c) {
            p.cls_off[b * (p.nc + 1) + c] = s;
            s += p.cls_count[b * p.nc + c];
        }
        p.cls_off[b * (p.nc + 1) + p.nc] = s;
    }
}

__global__ __launch_bounds__(256) void nms_bucket_kernel(NmsParams p)
{
    const long long b = blockIdx.y;
    const u32 n = p.img_count[b];
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 key = p.tmp_key[b * p.A + i];
    const u64 meta = p.tmp_meta[b * p.A + i];
    const u32 cls = (u32)(meta >> 32), ticket = (u32)meta;
    p.bkey[b * p.A + p.cls_off[b * (p.nc + 1) + cls] + ticket] = key;
}

// suppression test of the torchvision CPU kernel: i = kept (higher score) box, j = candidate
__device__ __forceinline__ bool iou_gt(float ix1, float iy1, float ix2, float iy2, float ia, float jx1, float jy1,
                                       float jx2, float jy2, float ja, float thr)
{
    const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
    const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    float uni = ia + ja;
    uni = uni - inter;
    const float ovr = inter / uni;
    return ovr > thr;
}

constexpr int RANK_TILE = 1024;
constexpr int RANK_R = 4;

__global__ __launch_bounds__(256) void nms_segment_kernel(NmsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_seg[];
    u64 *tk = (u64 *)smem_seg;                         // RANK_TILE keys (8 KB); reused by the sweep
    float4 *kb = (float4 *)smem_seg;                   // 64 kept boxes (aliases tk)
    float *ka = (float *)(smem_seg + 64 * sizeof(float4));   // 64 kept areas
    u32 *s_nk = (u32 *)(smem_seg + 64 * sizeof(float4) + 64 * sizeof(float));
    u64 *rem = (u64 *)(smem_seg + RANK_TILE * sizeof(u64));  // removed bitmask, ceil(n/64) words

    const int cls = blockIdx.x;
    const long long b = blockIdx.y;
    const u32 n = p.cls_count[b * p.nc + cls];
    if (n == 0) return;
    const u32 off = p.cls_off[b * (p.nc + 1) + cls];
    const u64 *keys = p.bkey + b * p.A + off;
    u64 *skey = p.skey + b * p.A + off;
    float4 *sbox = p.sbox + b * p.A + off;
    u64 *kseg = p.kseg + b * p.A + off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float coff = (float)cls * 4096.0f;

    // ---- rank by counting
    for (u32 base = 0; base < n; base += 256 * RANK_R) {
        u64 mk[RANK_R];
        u32 cnt[RANK_R];
#pragma unroll
        for (int r = 0; r < RANK_R; ++r) {
            const u32 i = base + r * 256 + tid;
            mk[r] = i < n ? keys[i] : ~0ull;   // ~0 is larger than any real key: counts nothing
            cnt[r] = 0;
        }
        for (u32 t0 = 0; t0 < n; t0 += RANK_TILE) {
            __syncthreads();
            const u32 tn = min((u32)RANK_TILE, n - t0);
            for (u32 j = tid; j < tn; j += 256) tk[j] = keys[t0 + j];
            __syncthreads();
            for (u32 j = 0; j < tn; ++j) {
                const u64 k = tk[j];
#pragma unroll
                for (int r = 0; r < RANK_R; ++r) cnt[r] += (k > mk[r]) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int r = 0; r < RANK_R; ++r) {
            const u32 i = base + r * 256 + tid;
            if (i < n) {
                const u32 idx = 0xFFFFFFFFu - (u32)mk[r];
                float x1, y1, x2, y2;
                row_box(p.pred + (b * p.A + idx) * (long long)p.C, p.in_cxcywh, p.hi, x1, y1, x2, y2);
                skey[cnt[r]] = mk[r];
                sbox[cnt[r]] = make_float4(x1 + coff, y1 + coff, x2 + coff, y2 + coff);
            }
        }
    }
    const u32 nwords = (n + 63) / 64;
    for (u32 w = tid; w < nwords; w += 256) rem[w] = 0ull;
    __syncthreads();   // sorted keys/boxes (global) and the cleared mask (LDS) are visible to the block

    // ---- greedy sweep, 64 boxes per step
    u32 nkept = 0;     // running count of kept boxes of this segment (uniform)
    for (u32 t = 0; t < nwords; ++t) {
        if (wave == 0) {
            const u32 j = t * 64 + lane;
            const bool inr = j < n;
            float4 bx = inr ? sbox[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float ar = (bx.z - bx.x) * (bx.w - bx.y);
            const bool alive = inr && !((rem[t] >> lane) & 1ull);
            u64 m = __ballot(alive);
            u64 kept = 0ull;
            while (m) {
                const int i = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                kept |= 1ull << i;
                const float ix1 = __shfl(bx.x, i), iy1 = __shfl(bx.y, i), ix2 = __shfl(bx.z, i), iy2 = __shfl(bx.w, i);
                const float ia = __shfl(ar, i);
                const bool sup = (lane > i) && iou_gt(ix1, iy1, ix2, iy2, ia, bx.x, bx.y, bx.z, bx.w, ar, p.iou_thr);
                m &= ~__ballot(sup);
            }
            const u32 nk = (u32)__popcll(kept);
            if ((kept >> lane) & 1ull) {
                const u32 pos = (u32)__popcll(kept & ((1ull << lane) - 1ull));
                kb[pos] = bx;
                ka[pos] = ar;
                kseg[nkept + pos] = skey[j];
            }
            if (lane == 0) *s_nk = nk;
        }
        __syncthreads();
        const u32 nk = *s_nk;
        nkept += nk;
        if (nk > 0) {
            for (u32 j0 = (t + 1 + wave) * 64; j0 < n; j0 += 256) {
                const u32 j = j0 + lane, w = j0 >> 6;
                const bool todo = j < n && !((rem[w] >> lane) & 1ull);
                bool sup = false;
                if (__ballot(todo)) {
                    const float4 bx = (j < n) ? sbox[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float ar = (bx.z - bx.x) * (bx.w - bx.y);
                    for (u32 k = 0; k < nk; ++k) {
                        const float4 kx = kb[k];
                        sup = sup || iou_gt(kx.x, kx.y, kx.z, kx.w, ka[k], bx.x, bx.y, bx.z, bx.w, ar, p.iou_thr);
                    }
                }
                const u64 sm = __ballot(sup && todo);
                if (lane == 0 && sm) rem[w] |= sm;
            }
        }
        __syncthreads();
    }
    if (tid == 0) p.kcnt[b * p.nc + cls] = nkept;
}

// kept rows of class `cls`: global rank = own position + number of larger keys in every other class
__global__ __launch_bounds__(256) void nms_gather_kernel(NmsParams p)
{
    const int cls = blockIdx.x;
    const long long b = blockIdx.y;
    const u32 *kcnt = p.kcnt + b * p.nc;
    const u32 *coff = p.cls_off + b * (p.nc + 1);
    if (cls == 0 && threadIdx.x == 0) {
        u32 s = 0;
        for (int c = 0; c < p.nc; ++c) s += kcnt[c];
        p.out_counts[b] = (int)s;
    }
    const u32 mine = kcnt[cls];
    const u64 *kbase = p.kseg + b * p.A;
    for (u32 i = threadIdx.x; i < mine; i += 256) {
        const u64 key = kbase[coff[cls] + i];
        u32 rank = i;
        for (int c = 0; c < p.nc; ++c) {
            const u32 m = kcnt[c];
            if (c == cls || m == 0) continue;
            const u64 *L = kbase + coff[c];   // descending
            u32 lo = 0, hi = m;               // count of elements > key
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (L[mid] > key) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank >= (u32)p.max_out) continue;
        const u32 idx = 0xFFFFFFFFu - (u32)key;
        const float *row = p.pred + (b * p.A + idx) * (long long)p.C;
        float *o = p.out_rows + (b * p.max_out + rank) * (long long)p.W;
        float x1, y1, x2, y2, best;
        int bi;
        row_box(row, p.in_cxcywh, p.hi, x1, y1, x2, y2);
        row_best(row + 5 + p.n_extra, p.nc, best, bi);
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = row[4]; o[5] = best;
        for (int e = 0; e < p.n_extra; ++e) o[6 + e] = row[5 + e];
        o[p.W - 1] = (float)bi;
        if (p.out_index) p.out_index[b * p.max_out + rank] = (int)idx;
    }
}

struct WsLayout {
    size_t tmp_key, tmp_meta, bkey, skey, kseg, sbox, counters, img_count, cls_count, kcnt, cls_off, total, counters_bytes;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

WsLayout ws_layout(int B, int A, int nc)
{
    WsLayout L;
    const size_t n = (size_t)B * A;
    size_t o = 0;
    // counters first: one memset clears them, from the workspace start, a multiple of 16 bytes
    L.counters = o;
    L.img_count = o; o += align_up((size_t)B * 4, 16);
    L.cls_count = o; o += align_up((size_t)B * nc * 4, 16);
    L.kcnt = o; o += align_up((size_t)B * nc * 4, 16);
    L.counters_bytes = o;
    L.cls_off = o; o += align_up((size_t)B * (nc + 1) * 4, 256);
    o = align_up(o, 256);
    L.tmp_key = o; o += align_up(n * 8, 256);
    L.tmp_meta = o; o += align_up(n * 8, 256);
    L.bkey = o; o += align_up(n * 8, 256);
    L.skey = o; o += align_up(n * 8, 256);
    L.kseg = o; o += align_up(n * 8, 256);
    L.sbox = o; o += align_up(n * 16, 256);
    L.total = o;
    return L;
}

}  // namespace

extern "C" size_t yamd_nms_workspace_bytes(int batch, int a, int c, int n_extra)
{
    const int nc = c - 5 - n_extra;
    if (batch <= 0 || a <= 0 || nc <= 0) return 0;
    return ws_layout(batch, a, nc).total;
}

extern "C" int yamd_nms_batched(const float *pred, int batch, int a, int c, int in_cxcywh, int n_extra, float img_size,
                                float conf_thr, double iou_thr, float *out_rows, int *out_index, int max_out,
                                int *out_counts, void *workspace, size_t workspace_bytes, void *stream)
{
    YAMD_REQUIRE(pred && out_rows && out_counts && workspace, "null argument");
    const int nc = c - 5 - n_extra;
    YAMD_REQUIRE(batch > 0 && batch <= 65535 && a > 0 && n_extra >= 0 && nc > 0, "bad shape B=%d A=%d C=%d extra=%d", batch, a,
                 c, n_extra);
    YAMD_REQUIRE(a <= (1 << 20), "A=%d exceeds the 2^20 anchors per image supported", a);
    YAMD_REQUIRE(max_out > 0, "max_out must be positive");
    // classes are swept independently; that equals the reference only while offset boxes of different
    // classes cannot intersect and IoU 0 never suppresses
    YAMD_REQUIRE(img_size - 1.0f < 4096.0f && img_size >= 1.0f, "img_size=%g outside (1, 4096]", (double)img_size);
    YAMD_REQUIRE(iou_thr >= 0.0, "nms threshold must be >= 0");
    YAMD_REQUIRE(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
    const WsLayout L = ws_layout(batch, a, nc);
    if (workspace_bytes < L.total) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, L.total);
        return YAMD_ENOSPC;
    }
    const size_t flds = (size_t)FROWS * (c | 1) * 4 + (size_t)nc * 8 + 16;
    YAMD_REQUIRE(flds <= 150 * 1024, "C=%d too wide for the filter tile", c);
    const size_t slds = (size_t)RANK_TILE * 8 + (size_t)((a + 63) / 64) * 8;
    YAMD_REQUIRE(slds <= 150 * 1024, "A=%d too large for the sweep bitmask", a);
    char *ws = (char *)workspace;
    NmsParams p;
    p.pred = pred;
    p.B = batch; p.A = a; p.C = c; p.in_cxcywh = in_cxcywh ? 1 : 0; p.n_extra = n_extra; p.nc = nc;
    p.W = 7 + n_extra;
    p.max_out = max_out;
    p.hi = img_size - 1.0f;
    p.conf_thr = conf_thr;
    float tf = (float)iou_thr;
    if ((double)tf > iou_thr) tf = nextafterf(tf, -INFINITY);
    p.iou_thr = tf;
    p.tmp_key = (u64 *)(ws + L.tmp_key);
    p.tmp_meta = (u64 *)(ws + L.tmp_meta);
    p.bkey = (u64 *)(ws + L.bkey);
    p.skey = (u64 *)(ws + L.skey);
    p.kseg = (u64 *)(ws + L.kseg);
    p.sbox = (float4 *)(ws + L.sbox);
    p.img_count = (u32 *)(ws + L.img_count);
    p.cls_count = (u32 *)(ws + L.cls_count);
    p.kcnt = (u32 *)(ws + L.kcnt);
    p.cls_off = (u32 *)(ws + L.cls_off);
    p.out_rows = out_rows;
    p.out_index = out_index;
    p.out_counts = out_counts;
    hipStream_t st = (hipStream_t)stream;

    static bool attr_done = false;
    if (!attr_done) {
        YAMD_HIP(hipFuncSetAttribute((const void *)nms_filter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        YAMD_HIP(hipFuncSetAttribute((const void *)nms_segment_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    YAMD_HIP(hipMemsetAsync(ws + L.counters, 0, L.counters_bytes, st));
    const int achunks = ceil_div(a, FROWS);
    hipLaunchKernelGGL(nms_filter_kernel, dim3(achunks, batch), dim3(256), flds, st, p);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(64), 0, st, p);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_bucket_kernel, dim3(achunks, batch), dim3(256), 0, st, p);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_segment_kernel, dim3(nc, batch), dim3(256), slds, st, p);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_gather_kernel, dim3(nc, batch), dim3(256), 0, st, p);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}
