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
//   4. segment  one workgroup per (image, class): bitonic-sort the bucket's keys (keys are unique, so
//               descending key order IS the stable descending score order), write keys and class-offset
//               boxes in sorted order, then run the greedy sweep in tiles of 64 boxes: the tile's 64x64
//               suppression matrix is built by all waves, wave 0 resolves it as a fixed point on 64-bit
//               masks (bit i = box i), the kept boxes are broadcast through LDS and the waves clear the
//               later boxes they suppress (one ballot -> one 64-bit word of the removed-bitmask in LDS per
//               group), skipping whole 64-box groups whose bounding box no kept box touches.
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
#include <atomic>

using namespace yamd;

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

struct NmsParams {
    const float *pred;
    int B, A, C, in_cxcywh, n_extra, nc, W, max_out;
    float hi, conf_thr, iou_thr;
    float iou_hi, iou_lo;        // iou_thr (1 +- 2^-20): see IouBand
    u64 *tmp_key, *tmp_meta, *bkey, *skey, *kseg;
    float4 *sbox, *mbox;
    u32 *mrank, *posr;
    int sort_cap;
    int big_occ8;                // buckets too large for the LDS sort also run in the two-per-compute-unit instantiation
    int cache_off, cache_groups; // LDS box cache of the sweep: byte offset behind the sweep state, capacity in groups of 64 boxes
    int frows;                   // rows per workgroup of the filter (256, fewer when a row is very wide)
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
    u32 *hist = (u32 *)(srow + p.frows * RSTR); // nc
    u32 *cbase = hist + p.nc;                   // nc
    u32 *misc = cbase + p.nc;                   // [0] block total, [1] image base
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * p.frows;
    const int nrows = min(p.frows, p.A - r0);
    const long long b = blockIdx.y;
    for (int c = tid; c < p.nc; c += 256) hist[c] = 0;
    if (tid == 0) misc[0] = 0;
    {
        const float *src = p.pred + (b * p.A + r0) * (long long)p.C;
        const int total = nrows * p.C;
        const float inv = 1.0f / (float)p.C;
        // 16-byte loads over the aligned part of the run; (row, column) of a lane's first element by one division, then
        // advanced by adds (1024 elements per round, one step per element of the quad)
        const int quads = (((uintptr_t)src & 15) == 0) ? (total >> 2) : 0;
        if (quads > 0) {
            int q = (int)((float)(tid * 4) * inv);
            int r = tid * 4 - q * p.C;
            if (r >= p.C) { ++q; r -= p.C; }
            if (r < 0) { --q; r += p.C; }
            const int qstep = 1024 / p.C, rstep = 1024 - qstep * p.C;
            for (int i = tid; i < quads; i += 256) {
                const float4 v = *(const float4 *)(src + 4 * i);
                const float e[4] = {v.x, v.y, v.z, v.w};
                int qq = q, rr = r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    srow[qq * RSTR + rr] = e[j];
                    if (++rr == p.C) { rr = 0; ++qq; }
                }
                q += qstep;
                r += rstep;
                if (r >= p.C) { r -= p.C; ++q; }
            }
        }
        for (int i = quads * 4 + tid; i < total; i += 256) {
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

// zeroes the per-image / per-class counters at the start of the workspace (16-byte pieces)
__global__ __launch_bounds__(256) void nms_clear_kernel(uint4 *counters, int pieces)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < pieces) counters[i] = make_uint4(0u, 0u, 0u, 0u);
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

// Suppression test of the torchvision CPU kernel (i = kept, higher score box, j = candidate):
//   w = max(0, min(ix2, jx2) - max(ix1, jx1)), h likewise, inter = w * h, ovr = inter / (ia + ja - inter), ovr > thr.
// The sweep's boxes are class-offset boxes: clamped to [0, hi], then + cls * 4096 (which also turns a -0 into +0).
// For non-negative finite floats the float order is the order of the bit patterns, so max / min can be integer
// instructions; they return the very floats fmaxf / fminf return (and need no NaN quieting of their operands).
__device__ __forceinline__ float max_pos(float a, float b) { return __uint_as_float(max(__float_as_uint(a), __float_as_uint(b))); }
__device__ __forceinline__ float min_pos(float a, float b) { return __uint_as_float(min(__float_as_uint(a), __float_as_uint(b))); }
__device__ __forceinline__ float iou_inter_pos(float ix1, float iy1, float ix2, float iy2, float jx1, float jy1, float jx2, float jy2)
{
    const float xx1 = max_pos(ix1, jx1), yy1 = max_pos(iy1, jy1);
    const float xx2 = min_pos(ix2, jx2), yy2 = min_pos(iy2, jy2);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    return w * h;
}

// `RN(inter / uni) > thr` without the division in all but borderline cases.  With hi = thr (1 + 2^-20) and
// lo = thr (1 - 2^-20):  inter > RN(uni * hi + 1e-30)  implies  inter / uni > thr (1 + 2^-21), which is beyond the float
// above thr, so the rounded quotient exceeds thr;  inter < RN(uni * lo - 1e-30)  implies  inter / uni < thr (1 - 2^-21),
// so the rounded quotient is at most thr.  (Each fma rounds once, 2^-24 relative; the +-1e-30 keeps both statements
// true where uni * thr would be subnormal: there neither comparison can hold.)  Everything else - quotients within
// 2^-20 of the threshold, NaN, thr = 0, vanishing or negative areas - is `open` and takes the division itself
// (iou_exact_gt), so the predicate is the reference's for every input.
struct IouBand {
    float thr, hi, lo;
};
__device__ __forceinline__ float iou_union(float inter, float ia, float ja)
{
    float uni = ia + ja;
    uni = uni - inter;
    return uni;
}
// lanes whose predicate holds, straight from the compare's mask (__ballot takes an int: a select and a second compare)
__device__ __forceinline__ u64 ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }
// The band test of a whole wave as lane masks (the three compares feed scalar mask arithmetic directly; bools combined
// per lane would go through a select and a second compare before a ballot, and && / || become branches)
__device__ __forceinline__ void iou_band_masks(float inter, float uni, const IouBand &band, u64 &gt, u64 &open)
{
    // both implications divide by uni: they need uni > 0.  Boxes with x2 < x1 or y2 < y1 (an xyxy input may hold them)
    // have negative "areas" and can make uni <= 0 or NaN: such pairs are open as well
    const u64 pos = ballot64(uni > 0.0f);
    const u64 above = ballot64(inter > __fmaf_rn(uni, band.hi, 1e-30f)), below = ballot64(inter < __fmaf_rn(uni, band.lo, -1e-30f));
    gt = pos & above;
    open = ~(pos & (above | below));
}
__device__ __forceinline__ bool iou_exact_gt(float inter, float uni, float thr)
{
    const float ovr = inter / uni;
    return ovr > thr;
}

constexpr int SEG_THREADS = 1024;           // 16 waves per bucket
constexpr int SEG_WAVES = SEG_THREADS / 64;
constexpr int CELLS = 8192;                 // cells of the grouping: 2 size classes x 64 x 64 places
constexpr int CELL_BITS = 13;

// Phase timing of nms_segment_kernel for the profiling build (-DYAMD_NMS_PHASES, scratch/nms_phases.py): thread 0 of
// every workgroup adds the shader-clock ticks between the marks to nms_phase_ticks.  Compiled out of the library.
#ifdef YAMD_NMS_PHASES
__device__ unsigned long long nms_phase_ticks[16];
// (the ticks are summed in registers and reach the global counters once per workgroup: an atomic per mark would queue
// behind those of every other workgroup and be waited for at the next barrier)
#define PH_DECL unsigned int ph_acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}; unsigned long long ph_t = clock64()
#define PH(k) do { if (tid == 0) { const unsigned long long ph_n = clock64(); ph_acc[k] += (unsigned int)(ph_n - ph_t); ph_t = ph_n; } } while (0)
#define PH_FLUSH do { if (tid == 0) for (int ph_k = 0; ph_k < 8; ++ph_k) atomicAdd(&nms_phase_ticks[ph_k], (unsigned long long)ph_acc[ph_k]); } while (0)
#define PH_COUNT(k, v) do { if (tid == 0) atomicAdd(&nms_phase_ticks[k], (unsigned long long)(v)); } while (0)
#if YAMD_NMS_PHASES > 1   // work counters of every wave as well: they slow the kernel down, the ticks of such a run mean nothing
#define PH_WAVE_COUNT(k, v) do { if (lane == 0) atomicAdd(&nms_phase_ticks[k], (unsigned long long)(v)); } while (0)
#else
#define PH_WAVE_COUNT(k, v)
#endif
#else
#define PH_DECL
#define PH(k)
#define PH_FLUSH
#define PH_COUNT(k, v)
#define PH_WAVE_COUNT(k, v)
#endif

// In-place bitonic sort, descending, of n (any n) 64-bit keys by one SEG_THREADS workgroup.  `a` may
// point to LDS or to global memory.  The network only uses comparators that move the larger key to the
// lower index (each merge starts with a mirror step), so the virtual padding "smaller than everything"
// beyond n never has to move and needs no storage.
__device__ void bitonic_sort_desc(u64 *a, u32 n, int tid)
{
    u32 npad = 1;
    while (npad < n) npad <<= 1;
    const u32 half = npad >> 1;
    for (u32 k = 2; k <= npad; k <<= 1) {
        const u32 h = k >> 1;
        for (u32 c = tid; c < half; c += SEG_THREADS) {
            const u32 i = ((c & ~(h - 1)) << 1) | (c & (h - 1));
            const u32 l = i ^ (k - 1);
            if (l < n) {
                const u64 x = a[i], y = a[l];
                if (x < y) { a[i] = y; a[l] = x; }
            }
        }
        __syncthreads();
        for (u32 j = k >> 2; j > 0; j >>= 1) {
            for (u32 c = tid; c < half; c += SEG_THREADS) {
                const u32 i = ((c & ~(j - 1)) << 1) | (c & (j - 1));
                const u32 l = i | j;
                if (l < n) {
                    const u64 x = a[i], y = a[l];
                    if (x < y) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// The same sort for keys that live in LDS, with the comparator network run three steps at a time: a thread loads the
// 8 keys that three consecutive steps (partner distances 4s, 2s, s) connect, exchanges them in registers and stores
// them back, so a 8192-key sort takes 33 passes over LDS (and barriers) instead of 91.  `a` must have room for the
// next power of two of n keys; the tail is filled with zeros, which are smaller than every real key.
__device__ __forceinline__ void cmpx(u64 &x, u64 &y, bool desc)
{
    const bool sw = desc ? (x < y) : (x > y);
    const u64 t = x;
    x = sw ? y : x;
    y = sw ? t : y;
}

// Key i of an LDS-resident bucket lives in slot i + i/8: the spare slot after every 8 keys spreads the 64-byte runs
// that one thread reads over all banks (2-way conflicts instead of 32-way).
__device__ __forceinline__ u32 lds_slot(u32 i) { return i + (i >> 3); }

__device__ void bitonic_sort_desc_blocked(u64 *a, u32 n, u32 npad, int tid)
{
    for (u32 i = n + tid; i < npad; i += SEG_THREADS) a[lds_slot(i)] = 0ull;
    __syncthreads();
    const u32 nthr = npad >> 3;
    // levels k = 2, 4, 8 touch only aligned runs of 8 keys: one pass
    if ((u32)tid < nthr) {
        u64 v[8];
        u64 *q = a + (size_t)tid * 9;
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = q[m];
#pragma unroll
        for (int m = 0; m < 8; m += 2) cmpx(v[m], v[m + 1], !(m & 2));                    // k=2
#pragma unroll
        for (int m = 0; m < 8; ++m) if (!(m & 2)) cmpx(v[m], v[m | 2], !(m & 4));         // k=4, j=2
#pragma unroll
        for (int m = 0; m < 8; m += 2) cmpx(v[m], v[m + 1], !(m & 4));                    // k=4, j=1
        const bool desc8 = !(tid & 1);                                                    // bit 3 of the key index 8*tid+m
#pragma unroll
        for (int m = 0; m < 4; ++m) cmpx(v[m], v[m | 4], desc8);                          // k=8, j=4
#pragma unroll
        for (int m = 0; m < 8; ++m) if (!(m & 2)) cmpx(v[m], v[m | 2], desc8);            // k=8, j=2
#pragma unroll
        for (int m = 0; m < 8; m += 2) cmpx(v[m], v[m + 1], desc8);                       // k=8, j=1
#pragma unroll
        for (int m = 0; m < 8; ++m) q[m] = v[m];
    }
    __syncthreads();
    for (u32 k = 16; k <= npad; k <<= 1) {
        u32 jh = k >> 1;
        while (jh >= 1) {
            const u32 sft = jh >= 4 ? (u32)(31 - __clz((int)jh)) - 2 : 0;       // log2(s)
            const int a_top = jh >= 4 ? 2 : (jh == 2 ? 1 : 0);
            const int nsteps = a_top + 1;
            if ((u32)tid < nthr) {
                const u32 blk = (u32)tid >> sft, r = (u32)tid & ((1u << sft) - 1u);
                const u32 e0 = ((blk << 3) << sft) + r;
                u64 v[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = a[lds_slot(e0 + ((u32)m << sft))];
#pragma unroll
                for (int st = 2; st >= 0; --st) {
                    if (st > a_top) continue;
                    const int dm = 1 << st;
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        if (m & dm) continue;
                        const bool desc = !((e0 + ((u32)m << sft)) & k);
                        cmpx(v[m], v[m | dm], desc);
                    }
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) a[lds_slot(e0 + ((u32)m << sft))] = v[m];
            }
            __syncthreads();
            jh >>= nsteps;
        }
    }
}

__device__ __forceinline__ u32 spread10(u32 v)   // 10 bits -> every second bit
{
    v &= 0x3ffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

__device__ __forceinline__ float rl(float v, int lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); }

// Element `i` of a wave-uniform array through a 32-bit byte offset (i * sizeof(T) < 2^32 is the caller's promise): the
// access then takes the base from scalar registers and one vector register for the offset, instead of a 64-bit
// address in two vector registers per access that is in flight.
template <typename T>
__device__ __forceinline__ T &at32(T *base, u32 i) { return *(T *)((char *)base + i * (u32)sizeof(T)); }

// One 16-wave workgroup per (class, image) bucket.
//  1. sort the bucket's keys (descending = the stable descending score order),
//  2. write keys / class-offset boxes in that order, then group the same boxes by place (counting sort into
//     grid cells numbered along a Morton curve) so that 64 consecutive boxes form a compact GROUP with a small
//     bounding box,
//  3. greedy sweep in tiles of 64 score-consecutive boxes, two barriers per tile:
//       B(t)  wave 0 resolves tile t on bit masks: lane j holds row j of the tile's 64x64 suppression matrix (which
//             earlier boxes of the tile remove box j) and the kept set is iterated to its fixed point; the kept
//             boxes go to LDS;
//       C(t)  the waves draw groups by ticket; the kept boxes (one per lane) are first tested against the group's
//             bounding box with one ballot, and only the few that can intersect it are applied to the group's
//             boxes; results land in the removed-bitmask (one 64-bit word per group);
//       A(t+1) in the same interval every wave computes 4 columns of the next tile's suppression matrix.
//     The bounding-box test only skips pairs whose intersection is exactly empty, so the kept set is
//     the one the plain greedy loop produces.
template <int MIN_WAVES>
__global__ __launch_bounds__(SEG_THREADS, MIN_WAVES) void nms_segment_kernel(NmsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_seg[];
    // image index fastest: workgroups are dealt round-robin over the 8 XCDs in linear block order, and the
    // heavy buckets are those of a few classes, so consecutive blocks must be different images of one class
    const int cls = blockIdx.y;
    const long long b = blockIdx.x;
    // (loaded by vector instructions, but the same for every lane: in scalar registers, and the bucket's array bases
    // below with them, they cost no vector registers)
    const u32 n = (u32)__builtin_amdgcn_readfirstlane((int)p.cls_count[b * p.nc + cls]);
    if (n == 0) return;
    const u32 off = (u32)__builtin_amdgcn_readfirstlane((int)p.cls_off[b * (p.nc + 1) + cls]);
    const long long base = b * p.A + off;
    const u64 *keys = p.bkey + base;
    u64 *skey = p.skey + base;
    float4 *sbox = p.sbox + base;
    u64 *kseg = p.kseg + base;
    float4 *mbox = p.mbox + base;
    u32 *mrank = p.mrank + base;
    u32 *posr = p.posr + base;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float coff = (float)cls * 4096.0f;
    PH_DECL;

    // ---- 1. sort by key: in LDS when the bucket fits, else in place in global memory (skey)
    u32 npad = 8;
    while (npad < n) npad <<= 1;
    const bool in_lds = npad <= (u32)p.sort_cap;
    // Two instantiations share the grid: the one limited to 64 registers runs two workgroups per compute unit (one
    // fills the other's serial phases) and takes the buckets whose sort fits LDS; a bucket beyond that is sorted in
    // global memory and, unless there are more such buckets than compute units (p.big_occ8), is alone on its
    // compute unit for most of its time: it takes the instantiation without the register limit.
    if ((in_lds || p.big_occ8) != (MIN_WAVES == 8)) return;
    u64 *S = in_lds ? (u64 *)smem_seg : skey;
    for (u32 i = tid; i < n; i += SEG_THREADS) S[in_lds ? lds_slot(i) : i] = keys[i];
    __syncthreads();
    if (in_lds) bitonic_sort_desc_blocked(S, n, npad, tid);
    else bitonic_sort_desc(S, n, tid);
    PH(0);

    // ---- 2. sorted keys + offset boxes out, then the same boxes grouped by place: a counting sort into the 4096
    // cells of a 64x64 grid, cells numbered along a Morton curve, so that 64 consecutive boxes (a GROUP) lie close
    // together.  The order inside a cell is whatever the tickets give; it only affects how tight the group bounding
    // boxes are, never the result.  Buckets of at most one group skip the grouping.
    const float qs = 1024.0f / (p.hi + 1.0f);
    const bool grouped = n > 64u;
    u32 *hist = (u32 *)smem_seg;                        // CELLS counters, then their exclusive prefix
    u32 *wsum = hist + CELLS;                           // SEG_WAVES wave totals
    auto cell_of = [&](float x1, float y1, float x2, float y2) __attribute__((always_inline)) {
        // one bit of size class above 12 bits of place: boxes of very different size cannot overlap much (the IoU is at
        // most the ratio of the areas), and a group of small boxes has a small bounding box that few kept boxes touch
        const u32 qx = (u32)fminf(fmaxf((x1 + x2) * 0.5f * qs, 0.0f), 1023.0f) >> 4;
        const u32 qy = (u32)fminf(fmaxf((y1 + y2) * 0.5f * qs, 0.0f), 1023.0f) >> 4;
        const float side2 = (x2 - x1) * (y2 - y1) * (qs * qs);          // area in units of (image / 1024)^2
        const u32 sc = side2 > 96.0f * 96.0f;                           // sides beyond about a tenth of the image
        return spread10(qx) | (spread10(qy) << 1) | (sc << 12);
    };
    auto scan_cells = [&]() __attribute__((always_inline)) {
        constexpr int PER = CELLS / SEG_THREADS;
        u32 sum = 0u;                                   // (the counts are read again below: no register array)
#pragma unroll
        for (int i = 0; i < PER; ++i) sum += hist[tid * PER + i];
        u32 inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 o = (u32)__shfl_up((int)inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 run = inc - sum;
        for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const u32 c = hist[tid * PER + i];
            hist[tid * PER + i] = run;
            run += c;
        }
    };
    if (in_lds) {
        // at most RPT ranks per thread (r = tid + 1024 q): cells and tickets stay in registers, one word for both
        // (cell < 2^13, ticket < n <= 2^13)
        constexpr int RPT = 8192 / SEG_THREADS, RH = 1;
        u32 cq[RPT];
        // (a copy of the thread index the compiler cannot see through: addresses formed from it here are not shared with
        // the later phases and do not stay in registers, or spill, across the loops in between)
        u32 tl = (u32)tid;
        asm volatile("" : "+v"(tl));
        // one rank at a time: the registers of more rows in flight would cost the second workgroup per compute unit
#pragma unroll
        for (int h = 0; h < RPT / RH; ++h) {
            u64 kq[RH];
            float raw[RH][4];
#pragma unroll
            for (int q = 0; q < RH; ++q) {
                const u32 r = tl + SEG_THREADS * (h * RH + q);
                kq[q] = r < n ? S[lds_slot(r)] : 0ull;
            }
#pragma unroll
            for (int q = 0; q < RH; ++q) {
                const u32 r = tl + SEG_THREADS * (h * RH + q);
                if (r < n) {
                    const float *row = p.pred + (b * p.A + (0xFFFFFFFFu - (u32)kq[q])) * (long long)p.C;
                    raw[q][0] = row[0]; raw[q][1] = row[1]; raw[q][2] = row[2]; raw[q][3] = row[3];
                } else {
                    raw[q][0] = raw[q][1] = raw[q][2] = raw[q][3] = 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < RH; ++q) {
                const u32 r = tl + SEG_THREADS * (h * RH + q);
                float x1, y1, x2, y2;
                row_box(raw[q], p.in_cxcywh, p.hi, x1, y1, x2, y2);
                const float4 ob = make_float4(x1 + coff, y1 + coff, x2 + coff, y2 + coff);
                cq[h * RH + q] = cell_of(x1, y1, x2, y2);
                if (r < n) {                            // r < 8192
                    at32(skey, r) = kq[q];
                    at32(sbox, r) = ob;
                    if (!grouped && h == 0) { at32(mrank, r) = r; at32(posr, r) = r; at32(mbox, r) = ob; }      // n <= 64
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // (rounds interleaved by the scheduler need more than 64 registers)
        }
        __syncthreads();                                // the sorted keys in LDS are dead from here
        PH(1);
        if (grouped) {
            for (u32 c = tid; c < (u32)CELLS; c += SEG_THREADS) hist[c] = 0u;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < RPT; ++q)
                if (tl + SEG_THREADS * q < n) cq[q] |= atomicAdd(&hist[cq[q]], 1u) << CELL_BITS;
            __syncthreads();
            scan_cells();
            __syncthreads();
            // (a second copy: with the index of the stores above the compiler would keep all eight boxes in registers
            // for the load below instead of reading them back)
            u32 tr = (u32)tid;
            asm volatile("" : "+v"(tr));
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const u32 r = tr + SEG_THREADS * q;
                if (r < n) {
                    const u32 sp = hist[cq[q] & (u32)(CELLS - 1)] + (cq[q] >> CELL_BITS);
                    at32(mrank, sp) = r;
                    at32(posr, r) = sp;
                    at32(mbox, sp) = at32(sbox, r);     // this thread's own store above
                }
            }
        }
    } else {
        // any bucket size: the keys were sorted in place in global memory; cells and tickets pass through posr
        for (u32 r = tid; r < n; r += SEG_THREADS) {
            const u64 key = S[r];
            const u32 idx = 0xFFFFFFFFu - (u32)key;
            float x1, y1, x2, y2;
            row_box(p.pred + (b * p.A + idx) * (long long)p.C, p.in_cxcywh, p.hi, x1, y1, x2, y2);
            sbox[r] = make_float4(x1 + coff, y1 + coff, x2 + coff, y2 + coff);
            posr[r] = cell_of(x1, y1, x2, y2);
        }
        for (u32 c = tid; c < (u32)CELLS; c += SEG_THREADS) hist[c] = 0u;
        __syncthreads();
        for (u32 r = tid; r < n; r += SEG_THREADS) {
            const u32 c = posr[r];
            const u32 t = atomicAdd(&hist[c], 1u);
            posr[r] = c | (t << CELL_BITS);             // cell < 8192, ticket < n <= 2^18
        }
        __syncthreads();
        scan_cells();
        __syncthreads();
        for (u32 r = tid; r < n; r += SEG_THREADS) {
            const u32 v = posr[r];
            const u32 sp = hist[v & (u32)(CELLS - 1)] + (v >> CELL_BITS);
            mrank[sp] = r;
            posr[r] = sp;
            mbox[sp] = sbox[r];
        }
    }
    __syncthreads();
    PH(2);

    // ---- sweep state in LDS (aliases the sort array)
    IouBand band;
    band.thr = p.iou_thr; band.hi = p.iou_hi; band.lo = p.iou_lo;
    const u32 ng = (n + 63) / 64;                       // groups == tiles == mask words
    float4 *gbb = (float4 *)smem_seg;                   // ng group bounding boxes
    float4 *kb = gbb + ng;                              // 64 kept boxes of the current tile
    unsigned char *rowb = (unsigned char *)(kb + 64);   // 64 x 16: the tile's suppression matrix, row j = 16 nibbles
    u64 *rem = (u64 *)(rowb + 64 * SEG_WAVES);          // ng removed/decided-bit words (by Morton position)
    u32 *gmax = (u32 *)(rem + ng);                      // ng: largest score rank inside the group
    float *ka = (float *)(gmax + ng);                   // 64 areas
    u32 *s_nk = (u32 *)(ka + 64);
    for (u32 g = wave; g < ng; g += SEG_WAVES) {
        const u32 s = g * 64 + lane;
        const bool v = s < n;
        const float4 bx = v ? mbox[s] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
        float mnx = bx.x, mny = bx.y, mxx = bx.z, mxy = bx.w;
        u32 mr = v ? mrank[s] : 0u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            mnx = fminf(mnx, __shfl_xor(mnx, d));
            mny = fminf(mny, __shfl_xor(mny, d));
            mxx = fmaxf(mxx, __shfl_xor(mxx, d));
            mxy = fmaxf(mxy, __shfl_xor(mxy, d));
            mr = max(mr, (u32)__shfl_xor((int)mr, d));
        }
        if (lane == 0) {
            gbb[g] = make_float4(mnx, mny, mxx, mxy);
            gmax[g] = mr;
            rem[g] = 0ull;
        }
    }

    // The first p.cache_groups groups' boxes wait in LDS behind the sweep state (the sort array is dead), so that phase C
    // touches global memory only for what lies beyond; the barrier before the first tile orders these stores.
    float4 *lbox = (float4 *)(smem_seg + p.cache_off);
    const u32 ncache = min((u32)p.cache_groups, ng);
    for (u32 g = wave; g < ncache; g += SEG_WAVES) {
        const u32 s = g * 64 + lane;
        lbox[g * 64 + lane] = s < n ? mbox[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Phase C hands groups to waves by ticket (an LDS counter that wave 0 resets in phase B): the groups' costs differ
    // by what the kept boxes of a tile touch, and a fixed assignment left the average wave waiting at the tile's second
    // barrier for as long as it had worked.  Ticket k is group ng - 1 - (k >> tshift) - the groups of large boxes, which
    // most kept boxes touch, lie last in the Morton order and go first - and share k & (2^tshift - 1) of its kept boxes:
    // a bucket of fewer than 16 groups splits every group between 2^tshift waves.
    u32 *s_tk = s_nk + 1;
    const u32 tshift = ng >= (u32)SEG_WAVES ? 0u : (u32)(31 - __clz((int)((u32)SEG_WAVES / ng)));
    const u32 ntk = ng << tshift;

    // columns [4*wave, 4*wave+4) of tile `t`'s suppression matrix (box i removes box j > i), stored by ROW: lane j
    // writes the four bits "i removes me" of this wave's columns as byte `wave` of its 16-byte row, so that bit i of
    // the 64-bit row j is nibble i / 4, bit i % 4.
    constexpr int CPW = 64 / SEG_WAVES;
    static_assert(CPW == 4, "a row is 16 nibbles");
    // `dead`: boxes of the tile that are already removed; a removed box suppresses nothing, its column stays zero
    auto phase_a = [&](u32 t, const float4 bx, const float ar, const u64 dead) __attribute__((always_inline)) {
        const u32 tn = min(64u, n - t * 64);
        const u64 valid = tn >= 64u ? ~0ull : (1ull << tn) - 1ull;
        u32 nib = 0u;
#pragma unroll
        for (int ii = 0; ii < CPW; ++ii) {
            const int i = wave * CPW + ii;
            if ((u32)i < tn && !((dead >> i) & 1ull)) {
                const float inter = iou_inter_pos(rl(bx.x, i), rl(bx.y, i), rl(bx.z, i), rl(bx.w, i), bx.x, bx.y, bx.z, bx.w);
                const float uni = iou_union(inter, rl(ar, i), ar);
                u64 gt, open;
                iou_band_masks(inter, uni, band, gt, open);
                const u64 later = valid & ~((2ull << i) - 1ull);          // lanes j > i that hold a box
                bool sup = (gt >> lane) & 1ull;
                if (open & later) sup = iou_exact_gt(inter, uni, band.thr);
                nib |= (sup && ((later >> lane) & 1ull)) ? (1u << ii) : 0u;
            }
        }
        rowb[lane * SEG_WAVES + wave] = (unsigned char)nib;
    };
    // 16 bytes with one nibble each -> 64 bits: four bytes of a word fold to 16 bits in two steps
    auto fold4 = [](u32 x) __attribute__((always_inline)) {
        x = (x | (x >> 4)) & 0x00ff00ffu;
        return (x | (x >> 8)) & 0x0000ffffu;
    };
    auto tile_box = [&](u32 t) __attribute__((always_inline)) {
        const u32 j = t * 64 + lane;
        return j < n ? sbox[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    float4 bx = tile_box(0);                            // this lane's box of the current tile (by score rank)
    float ar = (bx.z - bx.x) * (bx.w - bx.y);
    __syncthreads();                                    // gbb/gmax/rem complete
    phase_a(0, bx, ar, 0ull);
    u32 sp_cur = lane < (int)min(64u, n) ? posr[lane] : 0u;   // Morton position of this lane's tile box
    u64 key_cur = lane < (int)min(64u, n) ? skey[lane] : 0ull;
    __syncthreads();
    PH(3);

    u32 nkept = 0;
    for (u32 t = 0; t < ng; ++t) {
        const u32 tn = min(64u, n - t * 64);
        // ---- B(t): tile resolve on wave 0
        if (wave == 0) {
            const bool inr = (u32)lane < tn;
            const bool alive = inr && !((rem[sp_cur >> 6] >> (sp_cur & 63)) & 1ull);
            const uint4 rw = *(const uint4 *)(rowb + lane * SEG_WAVES);
            const u32 rlo = fold4(rw.x) | (fold4(rw.y) << 16), rhi = fold4(rw.z) | (fold4(rw.w) << 16);
            // greedy pass as a fixed point: box j is kept iff it is alive and no KEPT box i < j removes it.  A row only
            // holds bits i < j, so position j is final once positions < j are: every round settles at least one more
            // position, the fixed point is unique, and it is what the serial loop over i produces.
            u64 kept = ballot64(alive), prev;
            do {
                prev = kept;
                kept = ballot64(alive && ((rlo & (u32)prev) | (rhi & (u32)(prev >> 32))) == 0u);
            } while (kept != prev);
            if ((kept >> lane) & 1ull) {
                const u32 pos = (u32)__popcll(kept & ((1ull << lane) - 1ull));
                kb[pos] = bx;
                ka[pos] = ar;
                kseg[nkept + pos] = key_cur;
            }
            if (lane == 0) {
                *s_nk = (u32)__popcll(kept);
                *s_tk = 0u;
            }
            // every box of this tile is decided now: flag it so that later phases skip it
            if (inr) atomicOr(&rem[sp_cur >> 6], 1ull << (sp_cur & 63));
        }
        PH(4);
        __syncthreads();
        const u32 nk = *s_nk;
        PH(5);
        nkept += nk;
        // ---- next tile's data first (latency hides under C), then C(t), then A(t+1)
        const bool have_next = t + 1 < ng;
        float4 nbx = make_float4(0.f, 0.f, 0.f, 0.f);
        u32 nsp = 0;
        u64 nkey = 0ull;
        if (have_next) {
            const u32 j = (t + 1) * 64 + lane;
            if (j < n) {
                nbx = sbox[j];
                nsp = posr[j];
                if (wave == 0) nkey = skey[j];
            }
        }
        if (nk > 0 && have_next) {
            const u32 first_later = (t + 1) * 64;
            float4 kx = make_float4(0.f, 0.f, 0.f, 0.f);
            float kar = 0.f;
            if ((u32)lane < nk) {
                kx = kb[lane];
                kar = ka[lane];
            }
            // kept boxes of this tile against one group's boxes (fetched by `boxes` once the group is known to need them)
            auto apply = [&](const u32 g, const u32 part, const u32 parts, auto boxes) __attribute__((always_inline)) {
                const float4 bb = gbb[g];
                const bool ov = (u32)lane < nk && kx.z > bb.x && kx.x < bb.z && kx.w > bb.y && kx.y < bb.w;
                u64 rel = ballot64(ov);
                if (parts > 1u) {
                    // this wave's share of the relevant kept boxes: number part, part + parts, ... (picked here, on the
                    // scalar unit, so that the loop over them below stays as it is)
                    u64 r = rel, m = 0ull;
                    u32 ord = 0u, mine = part;
                    while (r) {
                        const u64 low = r & (0ull - r);
                        r ^= low;
                        if (ord++ == mine) { m |= low; mine += parts; }
                    }
                    rel = m;
                }
                PH_WAVE_COUNT(11, 1);
                if (rel == 0ull) return;
                const bool todo = g * 64 + lane < n && !((rem[g] >> lane) & 1ull);
                if (ballot64(todo) == 0ull) return;
                const float4 jx = boxes();
                const float ja = (jx.z - jx.x) * (jx.w - jx.y);
                PH_WAVE_COUNT(12, 1);
                PH_WAVE_COUNT(13, __popcll(rel));
                // the kept boxes come out of the lanes' registers (lane k holds kept box k); pairs the band test leaves
                // open are rare and wait in `redo` for the division, so that the loop has no branch but its own
                u64 supm = 0ull, redo = 0ull;
                const u64 todom = ballot64(todo);
                while (rel) {
                    const int k = __ffsll((long long)rel) - 1;
                    rel &= rel - 1ull;
                    const float inter = iou_inter_pos(rl(kx.x, k), rl(kx.y, k), rl(kx.z, k), rl(kx.w, k), jx.x, jx.y, jx.z, jx.w);
                    const float uni = iou_union(inter, rl(kar, k), ja);
                    u64 gt, open;
                    iou_band_masks(inter, uni, band, gt, open);
                    supm |= gt;
                    redo |= (open & todom) != 0ull ? 1ull << k : 0ull;
                }
                while (redo) {
                    const int k = __ffsll((long long)redo) - 1;
                    redo &= redo - 1ull;
                    const float inter = iou_inter_pos(rl(kx.x, k), rl(kx.y, k), rl(kx.z, k), rl(kx.w, k), jx.x, jx.y, jx.z, jx.w);
                    supm |= ballot64(iou_exact_gt(inter, iou_union(inter, rl(kar, k), ja), band.thr));
                }
                const u64 sm = supm & todom;
                if (lane == 0 && sm) atomicOr(&rem[g], sm);
            };
            auto ticket = [&]() __attribute__((always_inline)) {
                u32 v = 0u;
                if (lane == 0) v = atomicAdd(s_tk, 1u);
                return v;
            };
            u32 tv = ticket();
            for (;;) {
                const u32 tk = (u32)__builtin_amdgcn_readfirstlane((int)tv);
                if (tk >= ntk) break;
                tv = ticket();                          // the next one is on its way while this one is worked on
                const u32 g = ng - 1u - (tk >> tshift);
                if (gmax[g] >= first_later)
                    apply(g, tk & ((1u << tshift) - 1u), 1u << tshift, [&]() __attribute__((always_inline)) {
                        const u32 s = g * 64 + lane;
                        if (g < ncache) return lbox[g * 64 + lane];
                        return s < n ? mbox[s] : make_float4(0.f, 0.f, 0.f, 0.f);
                    });
            }
        }
        if (have_next) {
            bx = nbx;
            ar = (bx.z - bx.x) * (bx.w - bx.y);
            sp_cur = nsp;
            key_cur = nkey;
            // what earlier tiles (and the part of this one's phase C that has landed) removed of the next tile
            const bool gone = (t + 1) * 64 + lane < n && ((rem[nsp >> 6] >> (nsp & 63)) & 1ull);
            phase_a(t + 1, bx, ar, ballot64(gone));
        }
        PH(6);
        __syncthreads();
        PH(7);
    }
    if (tid == 0) p.kcnt[b * p.nc + cls] = nkept;
    PH_FLUSH; PH_COUNT(8, 1); PH_COUNT(9, n); PH_COUNT(10, nkept);
}

// kept rows of class `cls`: global rank = own position + number of larger keys in every other class.
// The other classes' kept lists (each already descending) are streamed through LDS in chunks and every
// thread binary-searches its own key there.
constexpr int GATHER_THREADS = 1024;
constexpr int GATHER_CHUNK = 2048;
constexpr int GSORT_CAP = 16384;     // kept rows per image the LDS gather handles (128 KB of keys)
constexpr int GSORT_MAXC = 256;      // ... and classes
__global__ __launch_bounds__(GATHER_THREADS) void nms_gather_kernel(NmsParams p)
{
    __shared__ u64 lk[GATHER_CHUNK];
    const int cls = blockIdx.y;
    const long long b = blockIdx.x;
    const u32 *kcnt = p.kcnt + b * p.nc;
    const u32 *coff = p.cls_off + b * (p.nc + 1);
    const int tid = threadIdx.x;
    if (cls == 0 && tid == 0) {
        u32 s = 0;
        for (int c = 0; c < p.nc; ++c) s += kcnt[c];
        p.out_counts[b] = (int)s;
    }
    const u32 mine = kcnt[cls];
    if (mine == 0) return;
    {
        u32 s = 0;
        for (int c = 0; c < p.nc; ++c) s += kcnt[c];
        if (s <= (u32)GSORT_CAP && p.nc <= GSORT_MAXC) return;     // nms_gather_sort_kernel writes this image
    }
    const u64 *kbase = p.kseg + b * p.A;
    for (u32 i0 = 0; i0 < mine; i0 += GATHER_THREADS) {
        const u32 i = i0 + tid;
        const bool have = i < mine;
        const u64 key = have ? kbase[coff[cls] + i] : 0ull;
        u32 rank = i;
        for (int c = 0; c < p.nc; ++c) {
            const u32 m = kcnt[c];
            if (c == cls || m == 0) continue;
            const u64 *L = kbase + coff[c];
            for (u32 c0 = 0; c0 < m; c0 += GATHER_CHUNK) {
                const u32 cn = min((u32)GATHER_CHUNK, m - c0);
                __syncthreads();
                for (u32 j = tid; j < cn; j += GATHER_THREADS) lk[j] = L[c0 + j];
                __syncthreads();
                if (have) {
                    u32 lo = 0, hi = cn;               // count of elements > key (list is descending)
                    while (lo < hi) {
                        const u32 mid = (lo + hi) >> 1;
                        if (lk[mid] > key) lo = mid + 1; else hi = mid;
                    }
                    rank += lo;
                }
            }
        }
        if (!have || rank >= (u32)p.max_out) continue;
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

// Gather for images with at most GSORT_CAP kept rows (the common case): one workgroup per image holds the kept keys of
// all classes in LDS (each class list is already in descending order).  The global rank of a key is its index in
// its own list plus, for every other class, the number of keys greater than it (binary search in LDS, no chunking).
// The keys are stored in rank order for nms_rows_kernel.
__global__ __launch_bounds__(GATHER_THREADS) void nms_gather_sort_kernel(NmsParams p)
{
    extern __shared__ __attribute__((aligned(16))) u64 gk[];             // the kept keys of the image, class after class
    __shared__ u32 cpos[GSORT_MAXC + 1];
    const long long b = blockIdx.x;
    const u32 *kcnt = p.kcnt + b * p.nc;
    const u32 *coff = p.cls_off + b * (p.nc + 1);
    const int tid = threadIdx.x;
    if (p.nc > GSORT_MAXC) return;                      // uniform: nms_gather_kernel takes over
    u32 total = 0;
    for (int c = 0; c < p.nc; ++c) total += kcnt[c];
    if (total > (u32)GSORT_CAP) return;                 // uniform: nms_gather_kernel takes this image
    if (tid == 0) {
        p.out_counts[b] = (int)total;
        u32 s = 0;
        for (int c = 0; c < p.nc; ++c) { cpos[c] = s; s += kcnt[c]; }
        cpos[p.nc] = s;
    }
    if (total == 0) return;
    const u64 *kbase = p.kseg + b * p.A;
    {
        u32 pos = 0;
        for (int c = 0; c < p.nc; ++c) {
            const u32 m = kcnt[c];
            const u64 *L = kbase + coff[c];
            for (u32 j = tid; j < m; j += GATHER_THREADS) gk[pos + j] = L[j];
            pos += m;
        }
    }
    __syncthreads();
    for (u32 e = tid; e < total; e += GATHER_THREADS) {
        const u64 key = gk[e];
        u32 rank = 0;
        for (int c = 0; c < p.nc; ++c) {
            const u32 lo0 = cpos[c], hi0 = cpos[c + 1];
            if (lo0 == hi0) continue;
            if (e >= lo0 && e < hi0) { rank += e - lo0; continue; }         // own list
            u32 lo = lo0, hi = hi0;                                         // count of keys > key (descending list)
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (gk[mid] > key) lo = mid + 1; else hi = mid;
            }
            rank += lo - lo0;
        }
        p.bkey[b * p.A + rank] = key;                   // the bucket keys are dead after the sweep: kept keys in rank order
    }
}

// Output rows of the images ranked by nms_gather_sort_kernel: 32 lanes per row, many workgroups per image so the
// dependent source-row loads of different rows overlap.  Lane c loads column c of the source row, the first-maximum
// class is reduced across the lanes, and lanes 0..W-1 write the output columns of row `rank`.
constexpr int ROWS_PER_BLOCK = 64;
__global__ __launch_bounds__(256) void nms_rows_kernel(NmsParams p)
{
    const long long b = blockIdx.y;
    const u32 *kcnt = p.kcnt + b * p.nc;
    if (p.nc > GSORT_MAXC) return;
    u32 total = 0;
    for (int c = 0; c < p.nc; ++c) total += kcnt[c];
    if (total > (u32)GSORT_CAP) return;
    const u32 nout = min(total, (u32)p.max_out);
    const u32 rbase = blockIdx.x * ROWS_PER_BLOCK;
    if (rbase >= nout) return;
    const int tid = threadIdx.x;
    const int sub = tid & 31, grp = tid >> 5;
    // inputs and outputs never overlap: with that known, the loads of all eight rows of a lane group are issued before
    // the first store, so the group pays the key -> row dependency once, not eight times
    const float *__restrict__ pred = p.pred;
    float *__restrict__ out_rows = p.out_rows;
    constexpr int NIT = ROWS_PER_BLOCK / 8;
    u32 idxs[NIT];
    float heads[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const u32 r = rbase + it * 8 + grp;
        idxs[it] = 0xFFFFFFFFu - (u32)p.bkey[b * p.A + (r < nout ? r : rbase)];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) heads[it] = (sub < 5) ? pred[(b * p.A + idxs[it]) * (long long)p.C + sub] : 0.0f;
    // ... and the lane's first class score of every row (all of them for up to 32 classes)
    float cls_first[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
        cls_first[it] = (sub < p.nc) ? pred[(b * p.A + idxs[it]) * (long long)p.C + 5 + p.n_extra + sub] : 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const u32 r = rbase + it * 8 + grp;
        const bool have = r < nout;                                    // uniform in the 32-lane group
        const u32 idx = idxs[it];
        const float *row = pred + (b * p.A + idx) * (long long)p.C;
        float *o = out_rows + (b * p.max_out + r) * (long long)p.W;
        const float head = heads[it];
        const float c0 = __shfl(head, 0, 32), c1 = __shfl(head, 1, 32), c2 = __shfl(head, 2, 32), c3 = __shfl(head, 3, 32);
        const float obj = __shfl(head, 4, 32);
        // first maximum over the classes: strided per lane, then across lanes (ties: smaller index); a NaN only
        // wins as class 0, exactly like the sequential scan that starts from class 0
        const float *cls0 = row + 5 + p.n_extra;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        if (sub < p.nc) {
            float v = cls_first[it];
            if (sub > 0 && v != v) v = -INFINITY;
            bv = v;
            bi = sub;
        }
        for (int c = sub + 32; c < p.nc; c += 32) {
            float v = cls0[c];
            if (v != v) v = -INFINITY;
            if (v > bv) { bv = v; bi = c; }
        }
#pragma unroll
        for (int m = 16; m > 0; m >>= 1) {
            const float ov = __shfl_xor(bv, m, 32);
            const int oi = __shfl_xor(bi, m, 32);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        bv = __shfl(bv, 0, 32);
        bi = __shfl(bi, 0, 32);
        float x1, y1, x2, y2;
        if (p.in_cxcywh) {
            const float hw = 0.5f * c2, hh = 0.5f * c3;
            x1 = c0 - hw; y1 = c1 - hh; x2 = c0 + hw; y2 = c1 + hh;
        } else {
            x1 = c0; y1 = c1; x2 = c2; y2 = c3;
        }
        x1 = clamp_box(x1, p.hi); y1 = clamp_box(y1, p.hi); x2 = clamp_box(x2, p.hi); y2 = clamp_box(y2, p.hi);
        if (!have) continue;
        const float ov = sub == 0 ? x1 : (sub == 1 ? y1 : (sub == 2 ? x2 : (sub == 3 ? y2 : (sub == 4 ? obj : bv))));
        if (sub < 6) o[sub] = ov;
        for (int e = sub; e < p.n_extra; e += 32) o[6 + e] = row[5 + e];
        if (sub == 6) o[p.W - 1] = (float)bi;
        if (sub == 7 && p.out_index) p.out_index[b * p.max_out + r] = (int)idx;
    }
}

__global__ void compact_scan_kernel(const int *counts, int batch, int max_out, long long *offsets)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long long s = 0;
        for (int b = 0; b < batch; ++b) {
            offsets[b] = s;
            s += min(counts[b], max_out);
        }
        offsets[batch] = s;
    }
}

__global__ __launch_bounds__(256) void compact_copy_kernel(const float *rows, const int *counts, int max_out, int width,
                                                           float *compact, const long long *offsets)
{
    const long long b = blockIdx.y;
    const long long nfl = (long long)min(counts[b], max_out) * width;
    const float *src = rows + b * max_out * width;
    float *dst = compact + offsets[b] * width;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nfl; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

struct WsLayout {
    size_t tmp_key, tmp_meta, bkey, skey, kseg, sbox, mbox, counters, img_count, cls_count, kcnt, cls_off, total, counters_bytes;
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
    L.mbox = o; o += align_up(n * 16, 256);
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

// Static scratch of yamd_nms_batched_static: one copy per GPU for the life of the process.  The name is a host-side
// handle only; the per-device address comes from hipGetSymbolAddress (scratch_for_current_device).
#define YAMD_NMS_STATIC_BYTES (16u << 20)
static __device__ __align__(256) unsigned char scratch[YAMD_NMS_STATIC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::atomic<void *> g_scratch_ptr[MAX_DEVICES];

int scratch_for_current_device(void **out)
{
    int dev = 0;
    YAMD_HIP(hipGetDevice(&dev));
    YAMD_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device ordinal %d outside [0, %d)", dev, MAX_DEVICES);
    void *ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (ptr == nullptr) {
        YAMD_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));   // this device's copy (the device is current)
        YAMD_REQUIRE(ptr != nullptr && ((uintptr_t)ptr % 256) == 0, "static scratch has no usable device address");
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);    // every thread resolves the same address
    }
    *out = ptr;
    return YAMD_OK;
}

int nms_batched_impl(const float *pred, int batch, int a, int c, int in_cxcywh, int n_extra, float img_size, float conf_thr,
                     double iou_thr, float *out_rows, int *out_index, int max_out, int *out_counts, void *workspace,
                     size_t workspace_bytes, void *stream, bool prefiltered = false);
}  // namespace

namespace yamd {
int nms_init_current_device(int device)
{
    (void)device;
    int rc = KernelAttr<&nms_filter_kernel, 160 * 1024>::ensure();
    if (rc == YAMD_OK) rc = KernelAttr<&nms_segment_kernel<8>, 160 * 1024>::ensure();
        if (rc == YAMD_OK) rc = KernelAttr<&nms_segment_kernel<4>, 160 * 1024>::ensure();
    if (rc == YAMD_OK) rc = KernelAttr<&nms_gather_sort_kernel, 150 * 1024>::ensure();
    if (rc != YAMD_OK) return rc;
    void *ptr = nullptr;
    return scratch_for_current_device(&ptr);
}
}  // namespace yamd

extern "C" int yamd_nms_batched(const float *pred, int batch, int a, int c, int in_cxcywh, int n_extra, float img_size,
                                float conf_thr, double iou_thr, float *out_rows, int *out_index, int max_out,
                                int *out_counts, void *workspace, size_t workspace_bytes, void *stream)
{
    YAMD_REQUIRE(workspace, "null argument");
    return nms_batched_impl(pred, batch, a, c, in_cxcywh, n_extra, img_size, conf_thr, iou_thr, out_rows, out_index, max_out,
                            out_counts, workspace, workspace_bytes, stream);
}

extern "C" size_t yamd_nms_static_capacity_bytes(void) { return YAMD_NMS_STATIC_BYTES; }

extern "C" int yamd_nms_batched_static(const float *pred, int batch, int a, int c, int in_cxcywh, int n_extra, float img_size,
                                       float conf_thr, double iou_thr, float *out_rows, int *out_index, int max_out,
                                       int *out_counts, void *stream)
{
    const int nc = c - 5 - n_extra;
    YAMD_REQUIRE(batch > 0 && a > 0 && n_extra >= 0 && nc > 0, "bad shape B=%d A=%d C=%d extra=%d", batch, a, c, n_extra);
    const size_t need = ws_layout(batch, a, nc).total;
    if (need > YAMD_NMS_STATIC_BYTES) {
        set_error("yamd_nms_batched_static: batch %d x %d anchors x %d classes needs %zu bytes of scratch, the static array "
                  "holds %u; use yamd_nms_batched with a caller-owned workspace",
                  batch, a, nc, need, (unsigned)YAMD_NMS_STATIC_BYTES);
        return YAMD_ENOSPC;
    }
    void *ptr = nullptr;
    const int rc = scratch_for_current_device(&ptr);
    if (rc != YAMD_OK) return rc;
    return nms_batched_impl(pred, batch, a, c, in_cxcywh, n_extra, img_size, conf_thr, iou_thr, out_rows, out_index, max_out,
                            out_counts, ptr, YAMD_NMS_STATIC_BYTES, stream);
}

namespace yamd {
int nms_filter_targets(int batch, int a, int c, int n_extra, float conf_thr, void *workspace, size_t workspace_bytes, void *stream,
                       FilterTargets *ft)
{
    YAMD_REQUIRE(workspace && ft, "null argument");
    const int nc = c - 5 - n_extra;
    YAMD_REQUIRE(batch > 0 && a > 0 && n_extra >= 0 && nc > 0 && nc <= 65535 && batch <= 65535, "bad shape B=%d A=%d C=%d extra=%d", batch, a,
                 c, n_extra);
    YAMD_REQUIRE(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
    const WsLayout L = ws_layout(batch, a, nc);
    if (workspace_bytes < L.total) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, L.total);
        return YAMD_ENOSPC;
    }
    char *ws = (char *)workspace;
    const int pieces = (int)(L.counters_bytes / 16);
    hipLaunchKernelGGL(nms_clear_kernel, dim3(ceil_div(pieces, 256)), dim3(256), 0, (hipStream_t)stream, (uint4 *)(ws + L.counters), pieces);
    YAMD_LAUNCH_CHECK();
    ft->img_count = (u32 *)(ws + L.img_count);
    ft->cls_count = (u32 *)(ws + L.cls_count);
    ft->tmp_key = (u64 *)(ws + L.tmp_key);
    ft->tmp_meta = (u64 *)(ws + L.tmp_meta);
    ft->a = a;
    ft->nc = nc;
    ft->n_extra = n_extra;
    ft->conf_thr = conf_thr;
    return YAMD_OK;
}
}  // namespace yamd

extern "C" int yamd_decode_nms_anchor(int kind, const float *const *raw_host_ptrs, const int *hs_host, const int *ws_host,
                                      const int *strides_host, const float *anchors_wh_host, int n_levels, int batch, int na, int c,
                                      int point_num, float img_size, int clamp_points, float *pred, float conf_thr, double iou_thr,
                                      float *out_rows, int *out_index, int max_out, int *out_counts, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    YAMD_REQUIRE(hs_host && ws_host && pred && n_levels > 0 && n_levels <= 8 && na > 0, "bad argument");
    YAMD_REQUIRE(kind >= YAMD_DECODE_V5 && kind <= YAMD_DECODE_FACE && point_num >= 0, "unknown decode kind %d", kind);
    long long rows = 0;
    for (int i = 0; i < n_levels; ++i) rows += (long long)na * hs_host[i] * ws_host[i];
    YAMD_REQUIRE(rows > 0 && rows < (1ll << 31), "bad level sizes");
    const int a = (int)rows;
    const int n_extra = kind == YAMD_DECODE_POSE ? 3 * point_num : (kind == YAMD_DECODE_FACE ? 2 * point_num : 0);
    FilterTargets ft;
    int rc = nms_filter_targets(batch, a, c, n_extra, conf_thr, workspace, workspace_bytes, stream, &ft);
    if (rc != YAMD_OK) return rc;
    rc = decode_anchor_levels_filtered(kind, raw_host_ptrs, hs_host, ws_host, strides_host, anchors_wh_host, n_levels, batch, na, c,
                                       point_num, img_size, clamp_points, pred, ft, stream);
    if (rc == YAMD_ENOSPC) {
        // more classes than the fused tile holds: the two steps one after the other (the same result)
        rc = yamd_decode_anchor_levels(kind, raw_host_ptrs, hs_host, ws_host, strides_host, anchors_wh_host, n_levels, batch, na, c,
                                       point_num, img_size, clamp_points, pred, stream);
        if (rc != YAMD_OK) return rc;
        return nms_batched_impl(pred, batch, a, c, 1, n_extra, img_size, conf_thr, iou_thr, out_rows, out_index, max_out, out_counts,
                                workspace, workspace_bytes, stream);
    }
    if (rc != YAMD_OK) return rc;
    return nms_batched_impl(pred, batch, a, c, 1, n_extra, img_size, conf_thr, iou_thr, out_rows, out_index, max_out, out_counts,
                            workspace, workspace_bytes, stream, true);
}

namespace {
int nms_batched_impl(const float *pred, int batch, int a, int c, int in_cxcywh, int n_extra, float img_size, float conf_thr,
                     double iou_thr, float *out_rows, int *out_index, int max_out, int *out_counts, void *workspace,
                     size_t workspace_bytes, void *stream, bool prefiltered)
{
    YAMD_REQUIRE(pred && out_rows && out_counts && workspace, "null argument");
    const int nc = c - 5 - n_extra;
    YAMD_REQUIRE(batch > 0 && a > 0 && n_extra >= 0 && nc > 0 && nc <= 65535 && batch <= 65535, "bad shape B=%d A=%d C=%d extra=%d", batch, a,
                 c, n_extra);
    YAMD_REQUIRE(max_out > 0, "max_out must be positive");
    YAMD_REQUIRE((long long)batch * nc <= (1ll << 26), "batch x classes = %lld exceeds 2^26", (long long)batch * nc);
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
    // the filter stages FROWS rows per workgroup in LDS, fewer when a row is so wide that they would not fit
    int frows = FROWS;
    while (frows > 8 && (size_t)frows * (c | 1) * 4 + (size_t)nc * 8 + 16 > 150 * 1024) frows >>= 1;
    const size_t flds = (size_t)frows * (c | 1) * 4 + (size_t)nc * 8 + 16;
    YAMD_REQUIRE(flds <= 150 * 1024, "C=%d too wide for the filter tile", c);
    // 20 bits of posr hold a candidate's ticket inside its grid cell, and the sweep keeps 28 bytes per group of 64
    // candidates in LDS
    YAMD_REQUIRE(a <= (1 << 18), "A=%d exceeds the 262144 anchors per image supported", a);
    int sort_cap = 8;
    while (sort_cap < a && sort_cap < 8192) sort_cap <<= 1;
    const size_t ngmax = (size_t)(a + 63) / 64;
    const size_t sweep_lds = ngmax * (16 + 8 + 4) + 64 * (16 + SEG_WAVES + 4) + 16;
    const size_t sort_lds = (size_t)sort_cap * 9;          // 8 bytes per key + one spare slot per 8 keys
    size_t slds = sort_lds > sweep_lds ? sort_lds : sweep_lds;
    const size_t group_lds = (size_t)(CELLS + SEG_WAVES) * 4;
    if (slds < group_lds) slds = group_lds;
    YAMD_REQUIRE(slds <= 160 * 1024, "A=%d needs %zu bytes of LDS for the sweep", a, slds);
    // the sweep's box cache fills what two workgroups per compute unit leave behind the sweep state
    const size_t cache_off = (sweep_lds + 15) & ~(size_t)15;
    size_t cache_groups = 0;
    if (cache_off + 1024 <= 80 * 1024) {
        cache_groups = (80 * 1024 - cache_off) / 1024;
        if (cache_groups > ngmax) cache_groups = ngmax;
        if (slds < cache_off + cache_groups * 1024) slds = cache_off + cache_groups * 1024;
    }
    char *ws = (char *)workspace;
    NmsParams p;
    p.pred = pred;
    p.B = batch; p.A = a; p.C = c; p.in_cxcywh = in_cxcywh ? 1 : 0; p.n_extra = n_extra; p.nc = nc;
    p.W = 7 + n_extra;
    p.sort_cap = sort_cap;
    {
        // such buckets only come one or a few per image (most candidates of an image in one class): with more images
        // than compute units they queue up, and two per compute unit beat one faster workgroup
        static std::atomic<int> cus[64];
        int dev = 0;
        YAMD_HIP(hipGetDevice(&dev));
        int ncu = dev >= 0 && dev < 64 ? cus[dev].load(std::memory_order_relaxed) : 0;
        if (ncu == 0) {
            YAMD_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
            if (dev >= 0 && dev < 64) cus[dev].store(ncu, std::memory_order_relaxed);
        }
        p.big_occ8 = batch > ncu ? 1 : 0;
    }
    p.cache_off = (int)cache_off;
    p.cache_groups = (int)cache_groups;
    p.max_out = max_out;
    p.hi = img_size - 1.0f;
    p.conf_thr = conf_thr;
    float tf = (float)iou_thr;
    if ((double)tf > iou_thr) tf = nextafterf(tf, -INFINITY);
    p.iou_thr = tf;
    p.iou_hi = tf * (1.0f + 0x1p-20f);
    p.iou_lo = tf * (1.0f - 0x1p-20f);
    p.tmp_key = (u64 *)(ws + L.tmp_key);
    p.tmp_meta = (u64 *)(ws + L.tmp_meta);
    p.bkey = (u64 *)(ws + L.bkey);
    p.skey = (u64 *)(ws + L.skey);
    p.kseg = (u64 *)(ws + L.kseg);
    p.sbox = (float4 *)(ws + L.sbox);
    p.mbox = (float4 *)(ws + L.mbox);
    // the unordered candidate lists are dead once the buckets are filled: their space holds the sweep's
    // Morton-order rank / position tables (4 bytes per candidate each)
    p.mrank = (u32 *)(ws + L.tmp_key);
    p.posr = (u32 *)(ws + L.tmp_meta);
    p.img_count = (u32 *)(ws + L.img_count);
    p.cls_count = (u32 *)(ws + L.cls_count);
    p.kcnt = (u32 *)(ws + L.kcnt);
    p.cls_off = (u32 *)(ws + L.cls_off);
    p.out_rows = out_rows;
    p.out_index = out_index;
    p.out_counts = out_counts;
    hipStream_t st = (hipStream_t)stream;

    {
        int rc = KernelAttr<&nms_filter_kernel, 160 * 1024>::ensure();
        if (rc == YAMD_OK) rc = KernelAttr<&nms_segment_kernel<8>, 160 * 1024>::ensure();
        if (rc == YAMD_OK) rc = KernelAttr<&nms_segment_kernel<4>, 160 * 1024>::ensure();
        if (rc == YAMD_OK) rc = KernelAttr<&nms_gather_sort_kernel, 150 * 1024>::ensure();
        if (rc != YAMD_OK) return rc;
    }
    const int achunks = ceil_div(a, 256);
    p.frows = frows;
    if (!prefiltered) {      // (prefiltered: the kernel that produced `pred` has filled the counters and candidate lists)
        // the counters are cleared by a kernel, not a memset, so that a captured step consists of kernel nodes only
        const int pieces = (int)(L.counters_bytes / 16);
        hipLaunchKernelGGL(nms_clear_kernel, dim3(ceil_div(pieces, 256)), dim3(256), 0, st, (uint4 *)(ws + L.counters), pieces);
        YAMD_LAUNCH_CHECK();
        hipLaunchKernelGGL(nms_filter_kernel, dim3(ceil_div(a, frows), batch), dim3(256), flds, st, p);
        YAMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(64), 0, st, p);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_bucket_kernel, dim3(achunks, batch), dim3(256), 0, st, p);
    YAMD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_segment_kernel<8>, dim3(batch, nc), dim3(SEG_THREADS), slds, st, p);
    YAMD_LAUNCH_CHECK();
    if (a > sort_cap && !p.big_occ8) {      // buckets larger than the LDS sort can exist
        hipLaunchKernelGGL(nms_segment_kernel<4>, dim3(batch, nc), dim3(SEG_THREADS), slds, st, p);
        YAMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nms_gather_sort_kernel, dim3(batch), dim3(GATHER_THREADS), (size_t)GSORT_CAP * sizeof(u64), st, p);
    YAMD_LAUNCH_CHECK();
    {
        const int cap = max_out < GSORT_CAP ? max_out : GSORT_CAP;
        hipLaunchKernelGGL(nms_rows_kernel, dim3(ceil_div(cap < a ? cap : a, ROWS_PER_BLOCK), batch), dim3(256), 0, st, p);
        YAMD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nms_gather_kernel, dim3(batch, nc), dim3(GATHER_THREADS), 0, st, p);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}
}  // namespace

extern "C" int yamd_compact_rows(const float *rows, const int *counts, int batch, int max_out, int width, float *compact,
                                 long long *offsets, void *stream)
{
    YAMD_REQUIRE(rows && counts && compact && offsets, "null argument");
    YAMD_REQUIRE(batch > 0 && batch <= 65535 && max_out > 0 && width > 0, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(64), 0, st, counts, batch, max_out, offsets);
    YAMD_LAUNCH_CHECK();
    const int gx = ceil_div((long long)max_out * width, 256 * 8);
    hipLaunchKernelGGL(compact_copy_kernel, dim3(gx > 64 ? 64 : gx, batch), dim3(256), 0, st, rows, counts, max_out, width,
                       compact, offsets);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

#ifdef YAMD_NMS_PHASES
extern "C" int yamd_debug_nms_phases(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(nms_phase_ticks), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nms_phase_ticks), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
