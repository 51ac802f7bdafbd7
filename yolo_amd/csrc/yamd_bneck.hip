// A residual bottleneck in one launch: y = [x +] act(conv3x3(act(conv1x1(x)))), both convs with bias (folded BatchNorm).
//
// The unfused form writes the 1x1 conv's output t to memory, reads it back with a halo for the 3x3 conv and reads x a
// second time for the shortcut: five tensor passes where two are needed (read x, write y).  On the high-resolution
// stages these layers have few channels and run at the speed of memory, so the passes are their cost.
//
// A workgroup owns a TH x TW tile of output pixels.  Stage 1 computes t on the tile's halo ((TH+2) x (TW+2) pixels, all
// CM channels) as a GEMM over 32-channel chunks of x staged through LDS, and leaves it in LDS as bf16 (the rounding the
// unfused path applies when it stores t), with zeros where the halo lies outside the image: the 3x3 conv's padding pads
// t, not x.  Stage 2 is the implicit GEMM of the general conv kernel with its activation image already resident: only
// the 3x3 weights stream through LDS, 16 input channels per chunk.  Operand roles, accumulator layout, packed weight
// format and the order of the MFMAs over k are those of conv_igemm_kernel, so the result equals the two unfused
// launches with 16-channel chunks bit for bit.
#include "yamd_conv_kernel.h"

namespace yamd {

struct BneckParams {
    const __bf16 *x;
    const char *w1, *w2;         // packed as yamd_conv_pack_weights does for (1x1, ck 32, nrep CM/32) and (3x3, ck 16, nrep CO/32)
    const float *b1, *b2;
    const __bf16 *res;           // shortcut source (x) or nullptr
    __bf16 *y;
    int x_cs, res_cs, y_cs;
    int H, W, TH, tiles_x, tiles_y, total_tiles, act;
};

template <int TW, int CI, int CM, int CO, int MREP, int NW>
struct BneckGeom {
    static constexpr int NT = NW * 64;
    static constexpr int NPW = TW + 2;
    static constexpr int THMAX = (NW * 32 * MREP) / TW;
    static constexpr int HPMAX = (THMAX + 2) * NPW;
    static constexpr int NREP1 = CM / 32, NREP2 = CO / 32;
    static constexpr int RST = CM * 2 + 16;          // bytes per halo pixel of the t image (odd multiple of 16: conflict-free)
    static constexpr int RSX = 32 * 2 + 16;          // bytes per halo pixel of an x chunk
    static constexpr int W1B = 2 * NREP1 * 1024;     // one 32-channel chunk of the 1x1 weights
    static constexpr int W2B = 9 * NREP2 * 1024;     // one 16-channel chunk of the 3x3 weights
    static constexpr int EPI = NW * 32 * (NREP2 * 64 + 16);
    static constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    static __host__ __device__ constexpr int t_bytes(int th) { return (th + 2) * NPW * RST; }
    // the epilogue tiles reuse the whole allocation (t is dead by then)
    static __host__ __device__ constexpr int lds_bytes(int th) { return max3(t_bytes(th) + W1B + (th + 2) * NPW * RSX, t_bytes(th) + W2B, EPI); }
};

template <int TW, int CI, int CM, int CO, int MREP, int NW, int ACTK>
__global__ __launch_bounds__(NW * 64, (NW == 4 || MREP == 1) ? 2 : 1) void bneck_kernel(BneckParams p)
{
    using G = BneckGeom<TW, CI, CM, CO, MREP, NW>;
    constexpr int NT = G::NT, NPW = G::NPW, NREP1 = G::NREP1, NREP2 = G::NREP2, RST = G::RST, RSX = G::RSX;
    constexpr int W1B = G::W1B, W2B = G::W2B;
    constexpr int NCH1 = CI / 32, NCH2 = CM / 16;
    constexpr int FW = ((G::HPMAX + 31) / 32 + NW - 1) / NW;       // stage-1 fragments (32 halo pixels) per wave
    constexpr int AP1 = (G::HPMAX * 4 + NT - 1) / NT;              // 16-byte pieces of an x chunk per thread
    constexpr int WP1 = (W1B / 16 + NT - 1) / NT;
    constexpr int WP2 = (W2B / 16 + NT - 1) / NT;
    static_assert(CI % 32 == 0 && CM % 32 == 0 && CO % 32 == 0 && NREP1 <= 4 && NREP2 <= 4, "channel counts");
    static_assert(AP1 <= 32, "one in-image bit per piece");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // consecutive workgroup ids go round-robin to the 8 XCDs; give each XCD a contiguous range of tiles (shared halos)
    const int per_xcd = gridDim.x >> 3;
    int bid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_tiles) return;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oh0 = ty * p.TH, ow0 = tx * TW;
    const int NPH = p.TH + 2, HP = NPH * NPW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c32 = lane & 31, g = lane >> 5;

    char *const tA = smem;
    char *const buf = smem + G::t_bytes(p.TH);
    char *const xA = buf + W1B;

    const __bf16 *ximg = p.x + (size_t)b * p.H * p.W * p.x_cs;

    // ---- stage 1: t = act(W1 x + b1) on the halo ------------------------------------------------------------------------
    int goff[AP1], loff[AP1];
    unsigned inb_mask = 0u;
    static_for<AP1>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        const int q = min(tid + u * NT, HP * 4 - 1);
        const int pix = q >> 2, gc = q & 3;
        const int hr = pix / NPW, hc = pix % NPW;
        const int ih = oh0 - 1 + hr, iw = ow0 - 1 + hc;
        const bool inb = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
        goff[u] = (ihc * p.W + iwc) * p.x_cs + gc * 8;
        loff[u] = pix * RSX + gc * 16;
        inb_mask |= inb ? (1u << u) : 0u;
    });
    u32x4 ra[AP1], rw1[WP1];
    auto issue1 = [&](int ch) __attribute__((always_inline)) {
        const char *wsrc = p.w1 + (size_t)ch * W1B;
        static_for<WP1>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            rw1[u] = *(const u32x4 *)(wsrc + min(tid + u * NT, W1B / 16 - 1) * 16);
        });
        static_for<AP1>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            ra[u] = *(const u32x4 *)(ximg + ch * 32 + goff[u]);
        });
    };
    auto write1 = [&]() __attribute__((always_inline)) {
        static_for<WP1>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            *(u32x4 *)(buf + min(tid + u * NT, W1B / 16 - 1) * 16) = rw1[u];
        });
        static_for<AP1>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            // the values of out-of-image pixels do not matter (their t is forced to zero below); keep them finite
            *(u32x4 *)(xA + loff[u]) = ((inb_mask >> u) & 1u) ? ra[u] : u32x4{0u, 0u, 0u, 0u};
        });
    };
    issue1(0);
    // this lane's halo pixel in each of the wave's fragments (clamped: surplus slots compute something valid, never stored)
    int hpix[FW];
#pragma unroll
    for (int f = 0; f < FW; ++f) hpix[f] = (wave * FW + f) * 32 + c32;
    f32x16 acc1[FW][NREP1];
#pragma unroll
    for (int n = 0; n < NREP1; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 bv = *(const f32x4 *)(p.b1 + n * 32 + 8 * i + 4 * g);
#pragma unroll
            for (int f = 0; f < FW; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1[f][n][4 * i + j] = bv[j];
        }
    write1();
    __syncthreads();
    u32x4 rw2[WP2];
    auto issue2 = [&](int ch) __attribute__((always_inline)) {
        const char *wsrc = p.w2 + (size_t)ch * W2B;
        static_for<WP2>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            rw2[u] = *(const u32x4 *)(wsrc + min(tid + u * NT, W2B / 16 - 1) * 16);
        });
    };
    auto write2 = [&]() __attribute__((always_inline)) {
        static_for<WP2>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            *(u32x4 *)(buf + min(tid + u * NT, W2B / 16 - 1) * 16) = rw2[u];
        });
    };
#pragma unroll
    for (int ch = 0; ch < NCH1; ++ch) {
        const bool more = ch + 1 < NCH1;
        if (more) issue1(ch + 1);
        else issue2(0);             // the first 3x3 weights travel under the last chunk's MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[FW], wv[NREP1];
#pragma unroll
            for (int f = 0; f < FW; ++f) a[f] = *(const bf16x8 *)(xA + min(hpix[f], HP - 1) * RSX + ks * 32 + g * 16);
#pragma unroll
            for (int n = 0; n < NREP1; ++n) wv[n] = *(const bf16x8 *)(buf + (ks * NREP1 + n) * 1024 + lane * 16);
#pragma unroll
            for (int f = 0; f < FW; ++f)
#pragma unroll
                for (int n = 0; n < NREP1; ++n)
                    acc1[f][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[n], a[f], acc1[f][n], 0, 0, 0);
        }
        __syncthreads();            // every wave is done reading this chunk
        if (more) {
            write1();
            __syncthreads();
        }
    }
    // t into its LDS image: lane holds, for halo pixel hpix[f], channels n*32 + 8i + 4g + j
#pragma unroll
    for (int f = 0; f < FW; ++f) {
        const int hp = hpix[f];
        const int hr = hp / NPW, hc = hp % NPW;
        const int ih = oh0 - 1 + hr, iw = ow0 - 1 + hc;
        const bool inimg = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        if (hp < HP) {
#pragma unroll
            for (int n = 0; n < NREP1; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = apply_act(acc1[f][n][4 * i + j], ACTK >= 0 ? ACTK : p.act);
                        o[j] = inimg ? (__bf16)v : (__bf16)0.0f;
                    }
                    *(bf16x4 *)(tA + hp * RST + (n * 32 + 8 * i + 4 * g) * 2) = o;
                }
        }
    }
    write2();
    __syncthreads();

    // ---- stage 2: 3x3 conv over the resident t image --------------------------------------------------------------------
    int pixoff[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = wave * (32 * MREP) + m * 32 + c32;
        int pr = pp / TW, pc = pp % TW;
        if (pr >= p.TH) { pr = 0; pc = 0; }
        pixoff[m] = (pr * NPW + pc) * RST + g * 16;
    }
    f32x16 acc[MREP][NREP2];
#pragma unroll
    for (int n = 0; n < NREP2; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 bv = *(const f32x4 *)(p.b2 + n * 32 + 8 * i + 4 * g);
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][n][4 * i + j] = bv[j];
        }
    for (int ch = 0; ch < NCH2; ++ch) {
        const bool more = ch + 1 < NCH2;
        if (more) issue2(ch + 1);
        const char *tch = tA + ch * 32;
        static_for<9>([&](auto tc) __attribute__((always_inline)) {
            constexpr int tap = decltype(tc)::value;
            constexpr int tapoff = ((tap / 3) * NPW + (tap % 3)) * RST;
            bf16x8 a[MREP], wv[NREP2];
#pragma unroll
            for (int m = 0; m < MREP; ++m) a[m] = *(const bf16x8 *)(tch + pixoff[m] + tapoff);
#pragma unroll
            for (int n = 0; n < NREP2; ++n) wv[n] = *(const bf16x8 *)(buf + (tap * NREP2 + n) * 1024 + lane * 16);
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int n = 0; n < NREP2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[n], a[m], acc[m][n], 0, 0, 0);
        });
        __syncthreads();
        if (more) {
            write2();
            __syncthreads();
        }
    }

    // ---- epilogue: activation, shortcut, bf16, through a per-wave LDS tile so that stores are 16 bytes per lane -----------
    constexpr int RSO = NREP2 * 64 + 16;
    constexpr int PPR = NREP2 * 4;
    constexpr int PIXPI = 64 / PPR;
    char *wreg = smem + wave * (32 * RSO);      // the last barrier of the chunk loop freed the t image and the weight buffer
    const int piece = lane % PPR, prow = lane / PPR;
    __bf16 *const sdst = p.y + piece * 8;
    const bool has_res = p.res != nullptr;      // uniform
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = wave * (32 * MREP) + m * 32 + c32;
        const int pr = pp / TW, pc = pp % TW;
        const bool valid = pr < p.TH && oh0 + pr < p.H && ow0 + pc < p.W;
        const int opix = valid ? (b * p.H + oh0 + pr) * p.W + ow0 + pc : -1;
        bf16x4 rr[NREP2][4];
        if (has_res) {
#pragma unroll
            for (int n = 0; n < NREP2; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    rr[n][i] = *(const bf16x4 *)(p.res + (opix >= 0 ? (size_t)opix * p.res_cs + n * 32 + 8 * i + 4 * g : (size_t)0));
        }
        *(int *)(wreg + c32 * RSO + NREP2 * 64) = opix;
#pragma unroll
        for (int n = 0; n < NREP2; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(acc[m][n][4 * i + j], ACTK >= 0 ? ACTK : p.act);
                if (has_res) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)rr[n][i][j];
                }
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                *(bf16x4 *)(wreg + c32 * RSO + (n * 32 + 8 * i + 4 * g) * 2) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / PIXPI; ++it) {
            const char *row = wreg + (it * PIXPI + prow) * RSO;
            const uint4 v = *(const uint4 *)(row + piece * 16);
            const int spix = *(const int *)(row + NREP2 * 64);
            if (spix >= 0) *(uint4 *)(sdst + (size_t)spix * p.y_cs) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

namespace {

template <int TW, int CI, int CM, int CO, int MREP, int NW, int ACTK>
int launch_bneck(BneckParams p, int batch, hipStream_t st)
{
    using G = BneckGeom<TW, CI, CM, CO, MREP, NW>;
    p.tiles_x = ceil_div(p.W, TW);
    p.tiles_y = ceil_div(p.H, G::THMAX);
    p.TH = ceil_div(p.H, p.tiles_y);
    const size_t lds = (size_t)G::lds_bytes(p.TH);
    YAMD_REQUIRE(lds <= 160 * 1024, "bottleneck tile needs %zu bytes of LDS", lds);
    {
        const int rc = KernelAttr<&bneck_kernel<TW, CI, CM, CO, MREP, NW, ACTK>, 160 * 1024>::ensure();
        if (rc != YAMD_OK) return rc;
    }
    const long long total = (long long)batch * p.tiles_y * p.tiles_x;
    YAMD_REQUIRE(total > 0 && total < (1ll << 31) - 8, "bottleneck grid out of range");
    p.total_tiles = (int)total;
    const long long grid = (total + 7) / 8 * 8;
    hipLaunchKernelGGL((bneck_kernel<TW, CI, CM, CO, MREP, NW, ACTK>), dim3((unsigned)grid), dim3(NW * 64), lds, st, p);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

template <int CI, int CM, int CO, int MREP, int NW>
int launch_bneck_shape(const BneckParams &p, int batch, hipStream_t st)
{
    const int tw = pick_tw(p.W);
    // the activation kinds the reference's residual units use are compile-time constants of their kernels (looked up per
    // value the activation is a chain of compares and branches: 1 750 branches in the 64-channel form)
    if (p.act == YAMD_ACT_SILU)
        return tw == 16 ? launch_bneck<16, CI, CM, CO, MREP, NW, YAMD_ACT_SILU>(p, batch, st)
                        : launch_bneck<20, CI, CM, CO, MREP, NW, YAMD_ACT_SILU>(p, batch, st);
    if (p.act == YAMD_ACT_LEAKY)
        return tw == 16 ? launch_bneck<16, CI, CM, CO, MREP, NW, YAMD_ACT_LEAKY>(p, batch, st)
                        : launch_bneck<20, CI, CM, CO, MREP, NW, YAMD_ACT_LEAKY>(p, batch, st);
    if (p.act == YAMD_ACT_MISH)
        return tw == 16 ? launch_bneck<16, CI, CM, CO, MREP, NW, YAMD_ACT_MISH>(p, batch, st)
                        : launch_bneck<20, CI, CM, CO, MREP, NW, YAMD_ACT_MISH>(p, batch, st);
    return tw == 16 ? launch_bneck<16, CI, CM, CO, MREP, NW, -1>(p, batch, st) : launch_bneck<20, CI, CM, CO, MREP, NW, -1>(p, batch, st);
}

bool shape_built(int cin, int cmid, int cout)
{
    return (cin == cout) && ((cin == 32 && cmid == 32) || (cin == 64 && (cmid == 64 || cmid == 32)) ||
                             (cin == 128 && (cmid == 128 || cmid == 64)));
}

void fill_desc(yamd_conv_desc *d, int ks, int cin, int cout, int ck)
{
    memset(d, 0, sizeof(*d));
    d->ksize = ks;
    d->stride = 1;
    d->cin = cin;
    d->cout = d->cout_pad = cout;
    d->ck = ck;
    d->nrep = cout / 32;
    d->mrep = 1;
    d->nwaves = 4;
    d->stages = 1;
    d->residual_scale = 1.0f;
    d->packed_w_elems = (int64_t)cout * cin * ks * ks;
}

}  // namespace
}  // namespace yamd

using namespace yamd;

extern "C" int64_t yamd_bottleneck_packed_elems(int cin, int cmid, int cout)
{
    return shape_built(cin, cmid, cout) ? (int64_t)cmid * cin + (int64_t)cout * cmid * 9 : 0;
}

extern "C" int yamd_bottleneck_pack_weights(int cin, int cmid, int cout, const float *w1, const float *b1, const float *w2,
                                            const float *b2, void *w_packed, float *bias_packed)
{
    YAMD_REQUIRE(w1 && w2 && w_packed && bias_packed, "null argument");
    YAMD_REQUIRE(shape_built(cin, cmid, cout), "no bottleneck kernel built for %d > %d > %d channels", cin, cmid, cout);
    yamd_conv_desc d1, d2;
    fill_desc(&d1, 1, cin, cmid, 32);
    fill_desc(&d2, 3, cmid, cout, 16);
    uint16_t *wp = (uint16_t *)w_packed;
    int rc = yamd_conv_pack_weights(&d1, w1, cin, b1, wp, bias_packed);
    if (rc != YAMD_OK) return rc;
    return yamd_conv_pack_weights(&d2, w2, cmid, b2, wp + (size_t)cmid * cin, bias_packed + cmid);
}

extern "C" int yamd_bottleneck(int cin, int cmid, int cout, int act, const void *x, int x_cstride, const void *w_packed,
                               const float *bias_packed, int shortcut, void *y, int y_cstride, int batch, int h, int w, void *stream)
{
    YAMD_REQUIRE(x && w_packed && bias_packed && y, "null argument");
    YAMD_REQUIRE(shape_built(cin, cmid, cout), "no bottleneck kernel built for %d > %d > %d channels", cin, cmid, cout);
    YAMD_REQUIRE(act >= YAMD_ACT_NONE && act <= YAMD_ACT_RELU, "unknown activation %d", act);
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0, "bad shape %dx%dx%d", batch, h, w);
    YAMD_REQUIRE(x_cstride >= cin && x_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0, "input needs 16-byte aligned pixels (cstride=%d)",
                 x_cstride);
    YAMD_REQUIRE(y_cstride >= cout && y_cstride % 8 == 0 && ((uintptr_t)y % 16) == 0, "output needs 16-byte aligned pixels (cstride=%d)",
                 y_cstride);
    YAMD_REQUIRE(((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)bias_packed % 16) == 0, "packed weights and bias must be 16-byte aligned");
    YAMD_REQUIRE((long long)h * w * x_cstride < (1ll << 31), "one input image must stay below 2^31 elements");
    YAMD_REQUIRE((long long)batch * h * w < (1ll << 31), "more than 2^31 output pixels in one launch");
    // tiles read the halo of their neighbours' pixels: the output must not overwrite the input.  Either the tensors are
    // disjoint, or they are different channel ranges of the same pixels (two slices of one buffer)
    const char *xb = (const char *)x, *yb = (const char *)y;
    const size_t npix = (size_t)batch * h * w;      // a tensor ends after the channels of its last pixel, not after that pixel's stride
    const size_t xbytes = ((npix - 1) * x_cstride + cin) * 2, ybytes = ((npix - 1) * y_cstride + cout) * 2;
    const long long dist = (long long)(yb - xb), pix = (long long)x_cstride * 2;
    const bool disjoint = yb + ybytes <= xb || xb + xbytes <= yb;
    const bool slices = x_cstride == y_cstride && ((dist >= cin * 2 && dist + cout * 2 <= pix) || (-dist >= cout * 2 && -dist + cin * 2 <= pix));
    YAMD_REQUIRE(disjoint || slices, "bottleneck output overlaps its input");
    BneckParams p;
    p.x = (const __bf16 *)x;
    p.w1 = (const char *)w_packed;
    p.w2 = p.w1 + (size_t)cmid * cin * 2;
    p.b1 = bias_packed;
    p.b2 = bias_packed + cmid;
    p.res = shortcut ? (const __bf16 *)x : nullptr;
    p.y = (__bf16 *)y;
    p.x_cs = p.res_cs = x_cstride;
    p.y_cs = y_cstride;
    p.H = h;
    p.W = w;
    p.act = act;
    p.TH = p.tiles_x = p.tiles_y = p.total_tiles = 0;
    hipStream_t st = (hipStream_t)stream;
    // (waves, 32-pixel fragments per wave) per shape as measured at batch 256: 4 waves x 2 for the square units, 8 x 1 for 64 > 32 > 64
    if (cin == 32) return launch_bneck_shape<32, 32, 32, 2, 4>(p, batch, st);
    if (cin == 64 && cmid == 64) return launch_bneck_shape<64, 64, 64, 2, 4>(p, batch, st);
    if (cin == 64) return launch_bneck_shape<64, 32, 64, 1, 8>(p, batch, st);
    if (cmid == 128) return launch_bneck_shape<128, 128, 128, 1, 8>(p, batch, st);
    // 128 > 64 > 128: 20-wide tiles of 12 rows keep the t image small enough for two workgroups per compute unit
    if (p.act == YAMD_ACT_SILU) return launch_bneck<20, 128, 64, 128, 2, 4, YAMD_ACT_SILU>(p, batch, st);
    return launch_bneck<20, 128, 64, 128, 2, 4, -1>(p, batch, st);
}
