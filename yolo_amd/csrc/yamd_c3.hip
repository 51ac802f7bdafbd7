// A whole C3 block with one bottleneck in ONE launch (luoyang/yolov5/YoloV5Body.py:96-111 with n = 1):
//
//     a = act(cv1(x)); b = act(cv2(x)); t = act(m.cv1(a)); u = act(m.cv2(t)) [3x3]; m = a + u (or u); y = act(cv3(cat(m, b)))
//
// for the narrow, high-resolution stage (64 channels in and out, 32 inside).  Launched layer by layer the block moves
// its tensors through memory five times (x read, a and b written, a read, m written, m and b read, y written); here x
// is read once and y written once, and nothing in between leaves the compute unit.
//
// A persistent workgroup of four waves walks 8 x 16 output tiles.  Per tile:
//   0. the tile's 10 x 18 halo of x (all 64 channels) is staged into LDS (the next tile's travels in registers meanwhile);
//   1. per 32-pixel fragment of the halo, a wave computes a (MFMA, weights in registers), and straight from the
//      accumulator registers -- rounded to bf16 as the layer-by-layer path stores it -- t = act(m.cv1(a)): the accumulator
//      of a 32x32 MFMA holds, per lane, 16 channels of ONE pixel, which is a valid B operand of the next MFMA once the
//      next layer's weights are packed in the matching channel order (pack_chained below).  a and t go to LDS images
//      (t as zeros outside the image: the 3x3 conv pads t).  The wave also computes b for its own 32 output pixels and
//      keeps it in registers;
//   2. u = 3x3 conv over the t image (weights resident in LDS), m = a + u with a from its LDS image, and again from
//      registers y = act(cv3([m | b])); y leaves through a per-wave LDS tile as whole 128-byte pixels.
// Three barriers per tile; two workgroups share a compute unit and fill each other's waits.
#include "yamd_conv_kernel.h"

namespace yamd {

namespace {

constexpr int C3_CI = 64, C3_CH = 32, C3_CO = 64;
constexpr int C3_TW = 16, C3_TH = 8;
constexpr int C3_NPW = C3_TW + 2, C3_NPH = C3_TH + 2, C3_HP = C3_NPW * C3_NPH;      // 180 halo pixels
constexpr int C3_HFRAG = (C3_HP + 31) / 32;                                          // 6 fragments of 32 halo pixels
constexpr int C3_RSX = C3_CI * 2 + 16;       // bytes per halo pixel of the x image (odd multiple of 16: conflict-free reads)
constexpr int C3_RSH = C3_CH * 2 + 16;       // the same for the a and t images
constexpr int C3_RSO = C3_CO * 2 + 16;       // output staging tile
constexpr int C3_XB = C3_HP * C3_RSX;        // 25 920
constexpr int C3_HB = C3_HP * C3_RSH;        // 14 400
constexpr int C3_W2B = 9 * 2 * 1024;         // 3x3 weights: [tap][k-step] fragments
constexpr int C3_NBIAS = 3 * C3_CH + C3_CH + C3_CO;      // cv1 | cv2 | m.cv1 | m.cv2 | cv3
constexpr int C3_LDS = C3_XB + 2 * C3_HB + C3_W2B + C3_NBIAS * 4;
constexpr int C3_XP = (C3_HP * 8 + 255) / 256;            // 16-byte pieces of the x halo per thread
constexpr int C3_NREGFRAG = 4 + 4 + 2 + 8;                // register-resident weight fragments per lane
static_assert(4 * 32 * C3_RSO <= C3_XB, "the output staging tiles reuse the x image");
static_assert(2 * C3_LDS <= 160 * 1024, "two workgroups per compute unit");

struct C3Params {
    const __bf16 *x;
    const bf16x8 *wreg;      // [cv1: 4][cv2: 4][m.cv1: 2][cv3: 2 x 4] fragments of 64 lanes x 8 bf16
    const char *wm2;         // [tap 9][k-step 2] fragments of 1 KB
    const float *bias;       // C3_NBIAS floats
    __bf16 *y;
    int x_cs, y_cs, H, W, tiles_x, tiles_y, total_tiles, shortcut;
};

__device__ __forceinline__ bf16x8 pack8(const f32x16 &v, int k)
{
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[8 * k + e];
    return o;
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void c3_small_kernel(C3Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const xA = smem;
    char *const aA = xA + C3_XB;
    char *const tA = aA + C3_HB;
    char *const w2 = tA + C3_HB;
    float *const sb = (float *)(w2 + C3_W2B);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c32 = lane & 31, g = lane >> 5;

    // every XCD (blockIdx.x mod 8) walks a contiguous range of tiles, its workgroups side by side in it: neighbouring
    // tiles (shared halos) run on one L2 at about the same time
    const int per_xcd = (p.total_tiles + 7) >> 3;
    const int xcd = blockIdx.x & 7, step = gridDim.x >> 3;
    int tl = blockIdx.x >> 3;
    auto tile_at = [&](int t) { return (t < per_xcd && xcd * per_xcd + t < p.total_tiles) ? xcd * per_xcd + t : -1; };
    int tile = tile_at(tl);
    if (tile < 0) return;

    for (int q = tid; q < C3_W2B / 16; q += 256) *(u32x4 *)(w2 + q * 16) = *(const u32x4 *)(p.wm2 + q * 16);
    if (tid < C3_NBIAS) sb[tid] = p.bias[tid];
    bf16x8 wc1[4], wc2[4], wm1[2], w3[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        wc1[k] = p.wreg[k * 64 + lane];
        wc2[k] = p.wreg[(4 + k) * 64 + lane];
        w3[0][k] = p.wreg[(10 + k) * 64 + lane];
        w3[1][k] = p.wreg[(14 + k) * 64 + lane];
    }
    wm1[0] = p.wreg[8 * 64 + lane];
    wm1[1] = p.wreg[9 * 64 + lane];

    // x staging: piece u of this thread = (halo pixel, 16-byte channel group), the same for every tile
    int loff[C3_XP], hrc[C3_XP];
#pragma unroll
    for (int u = 0; u < C3_XP; ++u) {
        const int q = min(tid + u * 256, C3_HP * 8 - 1);
        const int pix = q >> 3, gc = q & 7;
        loff[u] = pix * C3_RSX + gc * 16;
        hrc[u] = ((pix / C3_NPW) << 16) | ((pix % C3_NPW) << 8) | gc;
    }
    u32x4 rx[C3_XP];
    auto load_x = [&](int t, unsigned &mask) __attribute__((always_inline)) {
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y, b = t / p.tiles_y;
        const __bf16 *ximg = p.x + (size_t)b * p.H * p.W * p.x_cs;
        mask = 0u;
#pragma unroll
        for (int u = 0; u < C3_XP; ++u) {
            const int ih = ty * C3_TH - 1 + (hrc[u] >> 16), iw = tx * C3_TW - 1 + ((hrc[u] >> 8) & 255);
            const bool inb = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
            rx[u] = *(const u32x4 *)(ximg + (ihc * p.W + iwc) * p.x_cs + (hrc[u] & 255) * 8);
            mask |= inb ? (1u << u) : 0u;
        }
    };
    auto bias16 = [&](int base) __attribute__((always_inline)) -> f32x16 {
        f32x16 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 bv = *(const f32x4 *)(sb + base + 8 * i + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * i + j] = bv[j];
        }
        return v;
    };
    auto act16 = [&](f32x16 &v) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = apply_act(v[e], ACT);
    };

    unsigned mask;
    load_x(tile, mask);
    // this lane's output pixel: fragment `wave` of the tile = rows 2*wave, 2*wave + 1
    const int pr = wave * 2 + (c32 >> 4), pc = c32 & 15;
    const int hpi = (pr + 1) * C3_NPW + pc + 1;                 // its place in the halo images
    for (;;) {
#pragma unroll
        for (int u = 0; u < C3_XP; ++u) *(u32x4 *)(xA + loff[u]) = ((mask >> u) & 1u) ? rx[u] : u32x4{0u, 0u, 0u, 0u};
        tl += step;
        const int next = tile_at(tl);
        int t = tile;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y, b = t / p.tiles_y;
        const int oh0 = ty * C3_TH, ow0 = tx * C3_TW;
        __syncthreads();                                        // the x image (and, first time round, weights and biases) are in LDS
        if (next >= 0) load_x(next, mask);                      // in flight during this tile's arithmetic

        // ---- a and t on the halo
        for (int hf = wave; hf < C3_HFRAG; hf += 4) {
            const int hp = hf * 32 + c32, hpc = min(hp, C3_HP - 1);
            f32x16 acc = bias16(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc1[k], *(const bf16x8 *)(xA + hpc * C3_RSX + k * 32 + g * 16), acc, 0, 0, 0);
            act16(acc);
            const bf16x8 a0 = pack8(acc, 0), a1 = pack8(acc, 1);
            f32x16 tac = bias16(2 * C3_CH);
            tac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm1[0], a0, tac, 0, 0, 0);
            tac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm1[1], a1, tac, 0, 0, 0);
            act16(tac);
            const int hr = hpc / C3_NPW, hc = hpc - hr * C3_NPW;
            const int ih = oh0 - 1 + hr, iw = ow0 - 1 + hc;
            const bool inimg = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            if (hp < C3_HP) {
                // register quad 4i .. 4i+3 = channels 8i + 4g .. + 3
                const u32x4 av0 = __builtin_bit_cast(u32x4, a0), av1 = __builtin_bit_cast(u32x4, a1);
                char *ad = aA + hp * C3_RSH + g * 8, *td = tA + hp * C3_RSH + g * 8;
                *(uint2 *)(ad) = make_uint2(av0[0], av0[1]);
                *(uint2 *)(ad + 16) = make_uint2(av0[2], av0[3]);
                *(uint2 *)(ad + 32) = make_uint2(av1[0], av1[1]);
                *(uint2 *)(ad + 48) = make_uint2(av1[2], av1[3]);
                const u32x4 tv0 = __builtin_bit_cast(u32x4, pack8(tac, 0)), tv1 = __builtin_bit_cast(u32x4, pack8(tac, 1));
                const unsigned keep = inimg ? 0xFFFFFFFFu : 0u;
                *(uint2 *)(td) = make_uint2(tv0[0] & keep, tv0[1] & keep);
                *(uint2 *)(td + 16) = make_uint2(tv0[2] & keep, tv0[3] & keep);
                *(uint2 *)(td + 32) = make_uint2(tv1[0] & keep, tv1[1] & keep);
                *(uint2 *)(td + 48) = make_uint2(tv1[2] & keep, tv1[3] & keep);
            }
        }
        // ---- b on this wave's output pixels, kept in registers as the B operand of cv3
        bf16x8 b0, b1;
        {
            f32x16 acc = bias16(C3_CH);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc2[k], *(const bf16x8 *)(xA + hpi * C3_RSX + k * 32 + g * 16), acc, 0, 0, 0);
            act16(acc);
            b0 = pack8(acc, 0);
            b1 = pack8(acc, 1);
        }
        __syncthreads();                                        // a and t images complete; the x image is free

        // ---- u = 3x3 conv over t, m = a + u
        f32x16 uac = bias16(3 * C3_CH);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char *tp = tA + ((pr + tap / 3) * C3_NPW + pc + tap % 3) * C3_RSH + g * 16;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                uac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8 *)(w2 + (tap * 2 + k) * 1024 + lane * 16),
                                                              *(const bf16x8 *)(tp + k * 32), uac, 0, 0, 0);
        }
        act16(uac);
        if (p.shortcut) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x4 av = *(const bf16x4 *)(aA + hpi * C3_RSH + (8 * i + 4 * g) * 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) uac[4 * i + j] += (float)av[j];
            }
        }
        const bf16x8 m0 = pack8(uac, 0), m1 = pack8(uac, 1);

        // ---- y = act(cv3([m | b])) and out through this wave's staging tile
        char *const wreg = xA + wave * (32 * C3_RSO);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x16 yac = bias16(4 * C3_CH + n * 32);
            yac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[n][0], m0, yac, 0, 0, 0);
            yac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[n][1], m1, yac, 0, 0, 0);
            yac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[n][2], b0, yac, 0, 0, 0);
            yac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[n][3], b1, yac, 0, 0, 0);
            act16(yac);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)yac[4 * i + j];
                *(bf16x4 *)(wreg + c32 * C3_RSO + (n * 32 + 8 * i + 4 * g) * 2) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
            const int piece = lane & 7, prow = lane >> 3;           // 8 pieces of 16 bytes per pixel, 8 pixels per instruction
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + prow;                        // pixel of the fragment
                const int oh = oh0 + wave * 2 + (r >> 4), ow = ow0 + (r & 15);
                const uint4 v = *(const uint4 *)(wreg + r * C3_RSO + piece * 16);
                if (oh < p.H && ow < p.W)
                    *(uint4 *)(p.y + ((size_t)(b * p.H + oh) * p.W + ow) * p.y_cs + piece * 8) = v;
            }
        }
        if (next < 0) break;
        tile = next;
        __syncthreads();                                        // every wave is done with the images before the next tile overwrites them
    }
}

// fragment of 64 lanes x 8 bf16: lane (c32 = output channel, g), element e <- w[cout0 + c32][ci(g, e)]
template <class F>
void pack_fragment(uint16_t *dst, const float *w, int cin, int cout0, int taps, int tap, F &&ci_of)
{
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
            const int co = cout0 + (lane & 31), ci = ci_of(lane >> 5, e);
            const float v = w[((size_t)co * cin + ci) * taps + tap];
            uint32_t u;
            memcpy(&u, &v, 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) u = (u >> 16) | 0x40;      // keep a NaN a NaN
            else u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
            dst[lane * 8 + e] = (uint16_t)u;
        }
}

bool c3_shape_built(int cin, int ch, int cout) { return cin == C3_CI && ch == C3_CH && cout == C3_CO; }

}  // namespace
}  // namespace yamd

using namespace yamd;

extern "C" int64_t yamd_c3_packed_elems(int cin, int ch, int cout)
{
    return c3_shape_built(cin, ch, cout) ? (int64_t)(C3_NREGFRAG + 18) * 512 : 0;
}

extern "C" int yamd_c3_pack_weights(int cin, int ch, int cout, const float *w_cv1, const float *b_cv1, const float *w_cv2,
                                    const float *b_cv2, const float *w_m1, const float *b_m1, const float *w_m2, const float *b_m2,
                                    const float *w_cv3, const float *b_cv3, void *w_packed, float *bias_packed)
{
    YAMD_REQUIRE(w_cv1 && w_cv2 && w_m1 && w_m2 && w_cv3 && w_packed && bias_packed, "null argument");
    YAMD_REQUIRE(c3_shape_built(cin, ch, cout), "no fused C3 kernel built for %d > %d > %d channels", cin, ch, cout);
    uint16_t *out = (uint16_t *)w_packed;
    // operands read from LDS images keep the natural channel order: k-step k, lane group g, element e = channel 16k + 8g + e
    auto natural = [](int k) { return [k](int g, int e) { return 16 * k + 8 * g + e; }; };
    // operands that come straight from the previous layer's accumulator registers: a lane of group g holds, for k-step k,
    // the channels 16k + 4g + {0..3} and 16k + 8 + 4g + {0..3} of its pixel
    auto chained = [](int k, int base) { return [k, base](int g, int e) { return base + 16 * k + 8 * (e >> 2) + 4 * g + (e & 3); }; };
    int f = 0;
    for (int k = 0; k < 4; ++k) pack_fragment(out + (f++) * 512, w_cv1, cin, 0, 1, 0, natural(k));
    for (int k = 0; k < 4; ++k) pack_fragment(out + (f++) * 512, w_cv2, cin, 0, 1, 0, natural(k));
    for (int k = 0; k < 2; ++k) pack_fragment(out + (f++) * 512, w_m1, ch, 0, 1, 0, chained(k, 0));
    for (int n = 0; n < 2; ++n)
        for (int k = 0; k < 4; ++k) pack_fragment(out + (f++) * 512, w_cv3, 2 * ch, n * 32, 1, 0, chained(k & 1, k < 2 ? 0 : ch));
    for (int tap = 0; tap < 9; ++tap)
        for (int k = 0; k < 2; ++k) pack_fragment(out + (f++) * 512, w_m2, ch, 0, 9, tap, natural(k));
    float *bp = bias_packed;
    const float *bs[5] = {b_cv1, b_cv2, b_m1, b_m2, b_cv3};
    const int bn[5] = {ch, ch, ch, ch, cout};
    for (int s = 0; s < 5; ++s)
        for (int c = 0; c < bn[s]; ++c) *bp++ = bs[s] ? bs[s][c] : 0.0f;
    return YAMD_OK;
}

extern "C" int yamd_c3_fused(int cin, int ch, int cout, int act, const void *x, int x_cstride, const void *w_packed,
                             const float *bias_packed, int shortcut, void *y, int y_cstride, int batch, int h, int w, void *stream)
{
    YAMD_REQUIRE(x && w_packed && bias_packed && y, "null argument");
    YAMD_REQUIRE(c3_shape_built(cin, ch, cout), "no fused C3 kernel built for %d > %d > %d channels", cin, ch, cout);
    YAMD_REQUIRE(act == YAMD_ACT_SILU, "the fused C3 kernel is built for SiLU (got activation %d)", act);
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0, "bad shape %dx%dx%d", batch, h, w);
    YAMD_REQUIRE(x_cstride >= cin && x_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0, "input needs 16-byte aligned pixels (cstride=%d)",
                 x_cstride);
    YAMD_REQUIRE(y_cstride >= cout && y_cstride % 8 == 0 && ((uintptr_t)y % 16) == 0, "output needs 16-byte aligned pixels (cstride=%d)",
                 y_cstride);
    YAMD_REQUIRE(((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)bias_packed % 16) == 0, "packed weights and bias must be 16-byte aligned");
    YAMD_REQUIRE((long long)h * w * x_cstride < (1ll << 31), "one input image must stay below 2^31 elements");
    // tiles read the halo of their neighbours' pixels: the output must not overwrite the input
    {
        const char *xb = (const char *)x, *yb = (const char *)y;
        const size_t npix = (size_t)batch * h * w;
        const size_t xbytes = ((npix - 1) * x_cstride + cin) * 2, ybytes = ((npix - 1) * y_cstride + cout) * 2;
        const long long dist = (long long)(yb - xb), pix = (long long)x_cstride * 2;
        const bool disjoint = yb + ybytes <= xb || xb + xbytes <= yb;
        const bool slices = x_cstride == y_cstride && ((dist >= cin * 2 && dist + cout * 2 <= pix) || (-dist >= cout * 2 && -dist + cin * 2 <= pix));
        YAMD_REQUIRE(disjoint || slices, "fused C3 output overlaps its input");
    }
    C3Params p;
    p.x = (const __bf16 *)x;
    p.wreg = (const bf16x8 *)w_packed;
    p.wm2 = (const char *)w_packed + (size_t)C3_NREGFRAG * 1024;
    p.bias = bias_packed;
    p.y = (__bf16 *)y;
    p.x_cs = x_cstride;
    p.y_cs = y_cstride;
    p.H = h;
    p.W = w;
    p.tiles_x = ceil_div(w, C3_TW);
    p.tiles_y = ceil_div(h, C3_TH);
    const long long total = (long long)batch * p.tiles_x * p.tiles_y;
    YAMD_REQUIRE(total > 0 && total < (1ll << 31) - 8, "fused C3 grid out of range");
    p.total_tiles = (int)total;
    p.shortcut = shortcut ? 1 : 0;
    {
        const int rc = KernelAttr<&c3_small_kernel<YAMD_ACT_SILU>, 160 * 1024>::ensure();
        if (rc != YAMD_OK) return rc;
    }
    static std::atomic<int> cus[64];
    int dev = 0;
    YAMD_HIP(hipGetDevice(&dev));
    YAMD_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d outside [0, 64)", dev);
    int ncu = cus[dev].load(std::memory_order_relaxed);
    if (ncu == 0) {
        YAMD_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        cus[dev].store(ncu, std::memory_order_relaxed);
    }
    // persistent grid: two workgroups per compute unit, a multiple of the 8 XCDs, no more than the tiles need
    long long grid = (long long)ncu * 2 / 8 * 8;
    const long long need = (total + 7) / 8 * 8;
    if (grid > need) grid = need;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(c3_small_kernel<YAMD_ACT_SILU>, dim3((unsigned)grid), dim3(256), (size_t)C3_LDS, (hipStream_t)stream, p);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}
