// Conv + bias (folded BatchNorm) + activation (+ residual) as an NHWC implicit GEMM on bf16 MFMA.
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[pixel][k], k = (tap, cin).  A workgroup owns a
// TH x TW spatial tile of output pixels (BM = 128*MREP slots) times BN = 32*NREP output channels.
// Per input-channel chunk of CK channels it stages
//   - the input halo tile ((TH-1)*S+KS) x ((TW-1)*S+KS) pixels x CK channels into LDS once
//     (so a 3x3 conv reads HBM ~1.3x, not 9x), one padded row per halo pixel, and
//   - the matching packed weight blob (all taps) with a linear copy,
// then walks taps x 16-channel k-steps issuing v_mfma_f32_32x32x16_bf16 with the weights as the
// A operand (rows = cout) and activations as the B operand (cols = pixel).  The fp32 accumulator
// of a lane then holds 4 consecutive couts of one pixel per register quad, which is exactly an
// 8-byte NHWC store.
//
// LDS image of the activations: halo pixel row stride RS = CK*2 + 16 bytes.  RS/16 is odd, and a
// ds_read_b128 of the 32x32x16 operand serves 16 lanes of one k-group with 16 distinct pixels
// (mod 16) per LDS cycle, so the fragment reads are bank-conflict free; the staging writes
// (4..8 consecutive 16-byte pieces per pixel) are too.  For stride 2 the halo columns are stored
// de-interleaved (even columns first) so that the 32 pixels of a fragment are again consecutive.
#include "yamd_common.h"

#include <cstring>
#include <cmath>
#include <mutex>
#include <type_traits>

using namespace yamd;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct ConvParams {
    const __bf16 *x;
    const char *w;
    const float *bias;
    const __bf16 *res;
    void *y;
    int x_cs, res_cs, y_cs;
    int H, W, Hout, Wout, Cout;
    int TH, tiles_x, tiles_y, n_tiles, nchunks;
    int act;
};

__device__ __forceinline__ float apply_act(float v, int act)
{
    switch (act) {
    case YAMD_ACT_SILU:
        return v / (1.0f + __expf(-v));
    case YAMD_ACT_LEAKY:
        return v > 0.0f ? v : 0.1f * v;
    case YAMD_ACT_MISH: {
        float sp = v > 20.0f ? v : log1pf(__expf(v));
        return v * tanhf(sp);
    }
    case YAMD_ACT_RELU:
        return fmaxf(v, 0.0f);
    default:
        return v;
    }
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>); every array index
// inside f is a constant, so staging arrays stay in registers
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// activation of 4 values with ONE branch on the (uniform) kind
__device__ __forceinline__ void act4(float (&v)[4], int act)
{
    if (act == YAMD_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] / (1.0f + __expf(-v[j]));
    } else if (act != YAMD_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], act);
    }
}

template <int KS, int S, int TW, int MREP, int NREP, int CK, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvParams p)
{
    constexpr int PAD = KS / 2;
    constexpr int NPW = (TW - 1) * S + KS;   // halo width in input pixels
    constexpr int NPWH = (NPW + 1) / 2;      // even-column count (stride-2 de-interleave)
    constexpr int RS = CK * 2 + 16;          // LDS bytes per halo pixel
    constexpr int G8 = CK / 8;               // 16-byte pieces per halo pixel
    constexpr int KSTEPS = CK / 16;
    constexpr int TAPS = KS * KS;
    constexpr int WBYTES = TAPS * KSTEPS * NREP * 1024;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ldsW = smem;
    char *ldsA = smem + WBYTES;

    int bid = blockIdx.x;
    const int nt = bid % p.n_tiles;
    bid /= p.n_tiles;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oh0 = ty * p.TH, ow0 = tx * TW;
    const int NPH = (p.TH - 1) * S + KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c32 = lane & 31, g = lane >> 5;

    // LDS byte offset of this lane's pixel (tap 0,0) for each 32-pixel fragment
    int pixoff[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = wave * (32 * MREP) + m * 32 + c32;
        int pr = pp / TW, pc = pp % TW;
        if (pr >= p.TH) { pr = 0; pc = 0; }   // slot beyond the tile: read something valid, never stored
        pixoff[m] = ((pr * S) * NPW + pc) * RS + g * 16;
    }

    f32x16 acc[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // Staging registers: the next chunk's global loads are issued before the current chunk is computed
    // and written to LDS afterwards, so HBM/L2 latency hides under the MFMA work.
    constexpr int THMAX = (128 * MREP) / TW;
    constexpr int NPHMAX = (THMAX - 1) * S + KS;
    constexpr int AP = (NPHMAX * NPW * G8 + 255) / 256;   // activation pieces per thread
    constexpr int WP = (WBYTES / 16 + 255) / 256;         // weight pieces per thread
    u32x4 ra[AP], rw[WP];
    const int total = NPH * NPW * G8;

    // No divergent control flow in the staging: piece indices beyond the end are clamped (the duplicate
    // loads/stores are benign) and out-of-image pixels load a clamped address and are zeroed by a select.
    auto issue_loads = [&](int ch) __attribute__((always_inline)) {
        const int c0 = ch * CK;
        const char *wsrc = p.w + (size_t)(nt * p.nchunks + ch) * WBYTES;
        static_for<WP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const int q = min(tid + u * 256, WBYTES / 16 - 1);
            rw[u] = *(const u32x4 *)(wsrc + q * 16);
        });
        static_for<AP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const int q = min(tid + u * 256, total - 1);
            const int pix = q / G8, gc = q % G8;
            const int hr = pix / NPW, hc = pix % NPW;
            const int ih = oh0 * S - PAD + hr, iw = ow0 * S - PAD + hc;
            const bool inb = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
            const u32x4 v = *(const u32x4 *)(p.x + ((size_t)(b * p.H + ihc) * p.W + iwc) * p.x_cs + c0 + gc * 8);
            ra[u] = inb ? v : u32x4{0u, 0u, 0u, 0u};
        });
    };
    auto write_lds = [&]() __attribute__((always_inline)) {
        static_for<WP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const int q = min(tid + u * 256, WBYTES / 16 - 1);
            *(u32x4 *)(ldsW + q * 16) = rw[u];
        });
        static_for<AP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const int q = min(tid + u * 256, total - 1);
            const int pix = q / G8, gc = q % G8;
            const int hr = pix / NPW, hc = pix % NPW;
            const int lc = (S == 2) ? ((hc & 1) * NPWH + (hc >> 1)) : hc;
            *(u32x4 *)(ldsA + (hr * NPW + lc) * RS + gc * 16) = ra[u];
        });
    };

    issue_loads(0);
    write_lds();
    __syncthreads();
    for (int ch = 0; ch < p.nchunks; ++ch) {
        const bool more = ch + 1 < p.nchunks;
        if (more) issue_loads(ch + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kh = tap / KS, kw = tap % KS;
            const int tapoff = (S == 1) ? (kh * NPW + kw) * RS : (kh * NPW + (kw & 1) * NPWH + (kw >> 1)) * RS;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                bf16x8 a[MREP], wv[NREP];
#pragma unroll
                for (int m = 0; m < MREP; ++m) a[m] = *(const bf16x8 *)(ldsA + pixoff[m] + tapoff + ks * 32);
#pragma unroll
                for (int n = 0; n < NREP; ++n)
                    wv[n] = *(const bf16x8 *)(ldsW + ((tap * KSTEPS + ks) * NREP + n) * 1024 + lane * 16);
#pragma unroll
                for (int m = 0; m < MREP; ++m)
#pragma unroll
                    for (int n = 0; n < NREP; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[n], a[m], acc[m][n], 0, 0, 0);
            }
        }
        __syncthreads();   // every wave is done reading this chunk
        if (more) {
            write_lds();
            __syncthreads();
        }
    }

    // epilogue: lane holds, for pixel c32 of each fragment, couts 8*i + 4*g + j (i,j = 0..3)
    if constexpr (OUT_F32) {
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            const int pp = wave * (32 * MREP) + m * 32 + c32;
            const int pr = pp / TW, pc = pp % TW;
            const int oh = oh0 + pr, ow = ow0 + pc;
            const bool valid = pr < p.TH && oh < p.Hout && ow < p.Wout;
            const size_t pix = (size_t)(b * p.Hout + oh) * p.Wout + ow;
#pragma unroll
            for (int n = 0; n < NREP; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = nt * (NREP * 32) + n * 32 + 8 * i + 4 * g;
                    if (!valid || co >= p.Cout) continue;
                    float *y = (float *)p.y + pix * p.y_cs + co;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (co + j < p.Cout) y[j] = apply_act(acc[m][n][4 * i + j] + p.bias[co + j], p.act);
                }
            }
        }
    } else {
        // bf16 output goes through a per-wave LDS tile [32 pixels][BN couts] so that the global stores are
        // 16 bytes per lane and whole pixel rows (BN*2 contiguous bytes) per group of lanes.
        constexpr int RSO = NREP * 64 + 16;        // bytes per pixel row of the staging tile
        constexpr int PPR = NREP * 4;              // 16-byte pieces per pixel
        constexpr int PIXPI = 64 / PPR;            // pixels per wave store instruction
        char *wreg = smem + wave * (32 * RSO);     // the chunk loop ended with a barrier: LDS is free
        auto epilogue = [&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            const int pp = wave * (32 * MREP) + m * 32 + c32;
            const int pr = pp / TW, pc = pp % TW;
            const bool valid = pr < p.TH && oh0 + pr < p.Hout && ow0 + pc < p.Wout;
            const size_t pix = (size_t)(b * p.Hout + oh0 + pr) * p.Wout + ow0 + pc;
#pragma unroll
            for (int n = 0; n < NREP; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cl = n * 32 + 8 * i + 4 * g;
                    const int co = nt * (NREP * 32) + cl;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[m][n][4 * i + j] + p.bias[co + j];
                    act4(v, p.act);
                    if (p.res && valid && co < p.Cout) {
                        const bf16x4 r = *(const bf16x4 *)(p.res + pix * p.res_cs + co);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
                    }
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                    *(bf16x4 *)(wreg + c32 * RSO + cl * 2) = o;
                }
            }
            // the tile is private to this wave: order the LDS writes before the cross-lane reads
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 32 / PIXPI; ++it) {
                const int r = it * PIXPI + lane / PPR, piece = lane % PPR;
                const uint4 v = *(const uint4 *)(wreg + r * RSO + piece * 16);
                const int pq = wave * (32 * MREP) + m * 32 + r;
                const int qr = pq / TW, qc = pq % TW;
                const int co = nt * (NREP * 32) + piece * 8;
                if (qr < p.TH && oh0 + qr < p.Hout && ow0 + qc < p.Wout && co < p.Cout)
                    *(uint4 *)((__bf16 *)p.y + ((size_t)(b * p.Hout + oh0 + qr) * p.Wout + ow0 + qc) * p.y_cs + co) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        epilogue(std::integral_constant<int, 0>{});
        if constexpr (MREP > 1) epilogue(std::integral_constant<int, 1>{});
    }
}

template <int KS, int S, int TW, int MREP, int NREP, int CK, bool OUT_F32>
int launch_conv(const ConvParams &p, int batch, hipStream_t stream)
{
    constexpr int NPW = (TW - 1) * S + KS;
    constexpr int RS = CK * 2 + 16;
    constexpr int WBYTES = KS * KS * (CK / 16) * NREP * 1024;
    const int NPH = (p.TH - 1) * S + KS;
    size_t lds = (size_t)WBYTES + (size_t)NPH * NPW * RS;
    if (!OUT_F32 && lds < (size_t)4 * 32 * (NREP * 64 + 16)) lds = (size_t)4 * 32 * (NREP * 64 + 16);
    YAMD_REQUIRE(lds <= 160 * 1024, "conv tile needs %zu bytes of LDS", lds);
    auto kern = conv_igemm_kernel<KS, S, TW, MREP, NREP, CK, OUT_F32>;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    YAMD_HIP(attr_err);
    const long long grid = (long long)batch * p.tiles_y * p.tiles_x * p.n_tiles;
    YAMD_REQUIRE(grid > 0 && grid < (1ll << 31), "conv grid out of range");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

// tile width: prefer an exact divisor of the output width
int pick_tw(int wout) { return (wout % 16 == 0) ? 16 : ((wout % 20 == 0 || wout % 10 == 0) ? 20 : 16); }

}  // namespace

extern "C" int yamd_conv_desc_init(yamd_conv_desc *d, int ksize, int stride, int cin, int cout, int act, int out_f32)
{
    YAMD_REQUIRE(d != nullptr, "null descriptor");
    YAMD_REQUIRE((ksize == 1 && stride == 1) || (ksize == 3 && (stride == 1 || stride == 2)),
                 "unsupported conv k=%d s=%d", ksize, stride);
    YAMD_REQUIRE(cin > 0 && cin % 16 == 0, "cin=%d must be a positive multiple of 16", cin);
    YAMD_REQUIRE(cout > 0, "cout=%d", cout);
    YAMD_REQUIRE(act >= YAMD_ACT_NONE && act <= YAMD_ACT_RELU, "unknown activation %d", act);
    YAMD_REQUIRE(out_f32 || cout % 8 == 0, "bf16 output needs cout %% 8 == 0 (got %d)", cout);
    d->ksize = ksize;
    d->stride = stride;
    d->cin = cin;
    d->cout = cout;
    d->act = act;
    d->out_f32 = out_f32 ? 1 : 0;
    const int groups = (cout + 31) / 32;
    d->mrep = (stride == 2) ? 1 : 2;
    if (out_f32) {
        YAMD_REQUIRE(ksize == 1 && cin % 64 == 0, "float output is only built for 1x1 convs with cin %% 64 == 0");
        d->nrep = 4;
        d->ck = 64;
    } else if (ksize == 1) {
        d->nrep = groups >= 4 ? 4 : (groups >= 2 ? 2 : 1);
        d->ck = (cin % 64 == 0) ? 64 : 32;
        YAMD_REQUIRE(cin % 32 == 0, "1x1 conv needs cin %% 32 == 0 (got %d)", cin);
    } else if (stride == 1) {
        if (cin % 32 != 0) {          // 16-channel stem
            d->ck = 16;
            d->nrep = groups >= 4 ? 4 : 1;
            YAMD_REQUIRE(groups == 1 || groups >= 4, "3x3 conv with cin=%d cout=%d not built", cin, cout);
        } else if (groups >= 4) {
            d->ck = 16;
            d->nrep = 4;
        } else {
            d->ck = 32;
            d->nrep = groups >= 2 ? 2 : 1;
        }
    } else {
        if (groups >= 4) {
            d->ck = 16;
            d->nrep = 4;
        } else {
            d->ck = 32;
            d->nrep = 2;
            YAMD_REQUIRE(cin % 32 == 0, "3x3 s2 conv with cin=%d cout=%d not built", cin, cout);
        }
    }
    const int bn = d->nrep * 32;
    d->cout_pad = ((cout + bn - 1) / bn) * bn;
    d->packed_w_elems = (int64_t)d->cout_pad * cin * ksize * ksize;
    return YAMD_OK;
}

static inline uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // keep NaN a NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

extern "C" int yamd_conv_pack_weights(const yamd_conv_desc *d, const float *w, int cin_real, const float *bias,
                                      uint16_t *wp, float *bp)
{
    YAMD_REQUIRE(d && w && wp && bp, "null argument");
    YAMD_REQUIRE(cin_real > 0 && cin_real <= d->cin, "cin_real=%d exceeds stored cin=%d", cin_real, d->cin);
    const int KS = d->ksize, CK = d->ck, NREP = d->nrep, taps = KS * KS, ksteps = CK / 16;
    const int n_tiles = d->cout_pad / (NREP * 32), nchunks = d->cin / CK;
    size_t o = 0;
    for (int nt = 0; nt < n_tiles; ++nt)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int tap = 0; tap < taps; ++tap)
                for (int ks = 0; ks < ksteps; ++ks)
                    for (int n = 0; n < NREP; ++n)
                        for (int g = 0; g < 2; ++g)
                            for (int c = 0; c < 32; ++c)
                                for (int j = 0; j < 8; ++j) {
                                    const int co = (nt * NREP + n) * 32 + c;
                                    const int ci = ch * CK + ks * 16 + g * 8 + j;
                                    float v = 0.0f;
                                    if (co < d->cout && ci < cin_real)
                                        v = w[((size_t)co * cin_real + ci) * taps + tap];
                                    wp[o++] = f32_to_bf16_rne(v);
                                }
    if ((int64_t)o != d->packed_w_elems) {
        set_error("internal: packed %zu elements, expected %lld", o, (long long)d->packed_w_elems);
        return YAMD_EINVAL;
    }
    for (int co = 0; co < d->cout_pad; ++co) bp[co] = (bias && co < d->cout) ? bias[co] : 0.0f;
    return YAMD_OK;
}

extern "C" int yamd_conv2d_bias_act(const yamd_conv_desc *d, const void *x, int x_cstride, const void *w_packed,
                                    const float *bias_packed, const void *residual, int res_cstride, void *y,
                                    int y_cstride, int batch, int h, int w, void *stream)
{
    YAMD_REQUIRE(d && x && w_packed && bias_packed && y, "null argument");
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0, "bad shape %dx%dx%d", batch, h, w);
    YAMD_REQUIRE(x_cstride >= d->cin && x_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0,
                 "input needs 16-byte aligned pixels (cstride=%d)", x_cstride);
    YAMD_REQUIRE(((uintptr_t)w_packed % 16) == 0, "packed weights must be 16-byte aligned");
    if (!d->out_f32) {
        YAMD_REQUIRE(y_cstride >= d->cout && y_cstride % 8 == 0 && ((uintptr_t)y % 16) == 0,
                     "bf16 output needs 16-byte aligned pixels (cstride=%d)", y_cstride);
        if (residual)
            YAMD_REQUIRE(res_cstride >= d->cout && res_cstride % 4 == 0 && ((uintptr_t)residual % 8) == 0,
                         "residual needs 8-byte aligned pixels (cstride=%d)", res_cstride);
    } else {
        YAMD_REQUIRE(y_cstride >= d->cout && residual == nullptr, "float output: cstride=%d, no residual", y_cstride);
    }
    const int KS = d->ksize, S = d->stride, pad = KS / 2;
    ConvParams p;
    p.x = (const __bf16 *)x;
    p.w = (const char *)w_packed;
    p.bias = bias_packed;
    p.res = (const __bf16 *)residual;
    p.y = y;
    p.x_cs = x_cstride;
    p.res_cs = res_cstride;
    p.y_cs = y_cstride;
    p.H = h;
    p.W = w;
    p.Hout = (h + 2 * pad - KS) / S + 1;
    p.Wout = (w + 2 * pad - KS) / S + 1;
    p.Cout = d->cout;
    p.act = d->act;
    const int TW = pick_tw(p.Wout);
    const int th_max = (128 * d->mrep) / TW;
    p.tiles_x = ceil_div(p.Wout, TW);
    p.tiles_y = ceil_div(p.Hout, th_max);
    p.TH = ceil_div(p.Hout, p.tiles_y);
    p.n_tiles = d->cout_pad / (d->nrep * 32);
    p.nchunks = d->cin / d->ck;
    hipStream_t st = (hipStream_t)stream;

#define YAMD_CONV_CASE(KS_, S_, MREP_, NREP_, CK_, F32_)                                                   \
    if (KS == KS_ && S == S_ && d->mrep == MREP_ && d->nrep == NREP_ && d->ck == CK_ && d->out_f32 == F32_) \
        return TW == 16 ? launch_conv<KS_, S_, 16, MREP_, NREP_, CK_, (F32_ != 0)>(p, batch, st)           \
                        : launch_conv<KS_, S_, 20, MREP_, NREP_, CK_, (F32_ != 0)>(p, batch, st);

    YAMD_CONV_CASE(1, 1, 2, 1, 32, 0)
    YAMD_CONV_CASE(1, 1, 2, 2, 32, 0)
    YAMD_CONV_CASE(1, 1, 2, 4, 32, 0)
    YAMD_CONV_CASE(1, 1, 2, 1, 64, 0)
    YAMD_CONV_CASE(1, 1, 2, 2, 64, 0)
    YAMD_CONV_CASE(1, 1, 2, 4, 64, 0)
    YAMD_CONV_CASE(1, 1, 2, 4, 64, 1)
    YAMD_CONV_CASE(3, 1, 2, 1, 16, 0)
    YAMD_CONV_CASE(3, 1, 2, 4, 16, 0)
    YAMD_CONV_CASE(3, 1, 2, 1, 32, 0)
    YAMD_CONV_CASE(3, 1, 2, 2, 32, 0)
    YAMD_CONV_CASE(3, 2, 1, 4, 16, 0)
    YAMD_CONV_CASE(3, 2, 1, 2, 32, 0)
#undef YAMD_CONV_CASE
    set_error("no conv kernel built for k=%d s=%d mrep=%d nrep=%d ck=%d f32=%d", KS, S, d->mrep, d->nrep, d->ck,
              d->out_f32);
    return YAMD_EINVAL;
}
