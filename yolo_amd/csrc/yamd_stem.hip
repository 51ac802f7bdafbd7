// First conv of a network straight from the fp32 NCHW image: 3x3 stride 1 / stride 2 on the 3 colour channels, or the
// YOLOv5 Focus (space-to-depth) + 3x3 conv folded into one 6x6 stride-2 conv.  No padded NHWC copy of the image is
// ever written: a workgroup stages the fp32 patch of its output tile into LDS as bf16 (the rounding the rest of the
// graph sees too), every lane gathers the K = 27 (108) patch values of its output pixel from there as the MFMA B
// operand, and the weights live in registers as the A operand (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// Bias + activation, then the tile leaves through LDS as whole NHWC pixel rows.
#include "yamd_conv_kernel.h"

using namespace yamd;

namespace {

constexpr int TH = 8, TW = 32;       // output tile: 8 rows x 32 columns, wave w owns rows 2w and 2w+1

template <int KS, int S>
struct StemGeom {
    static constexpr int PAD = KS == 6 ? 2 : 1;
    static constexpr int R = (TH - 1) * S + KS;        // patch rows
    static constexpr int CW = ((TW - 1) * S + KS + 2) / 2 * 2;   // patch columns, even: pairs of columns stay 4-byte aligned
    static constexpr int K = 3 * KS * KS;
    static constexpr int KSTEPS = (K + 15) / 16;
    static constexpr int PATCH = 3 * R * CW;           // bf16 elements; element PATCH is a zero for k >= K
    // patch element of reduction index k relative to the pixel's origin; k = (ch * KS + ky) * KS + kx
    static constexpr int off(int k)
    {
        if (k >= K) return -1;
        const int ch = k / (KS * KS), rem = k % (KS * KS);
        return (ch * R + rem / KS) * CW + rem % KS;
    }
};

struct StemParams {
    const float *img;
    const bf16x8 *w;       // [kstep][ncol][lane] fragments
    const float *bias;
    __bf16 *y;
    int H, W, Hout, Wout, Cout, y_cs, act, tiles_x, tiles_y;
};

template <int KS, int S, int NREP>
__global__ __launch_bounds__(256) void stem_conv_kernel(StemParams p)
{
    using G = StemGeom<KS, S>;
    constexpr int RSO = NREP * 64 + 16;                 // bytes per pixel of the output staging tile
    constexpr int LDS_IN = (G::PATCH + 8) * 2;
    constexpr int LDS_OUT = 4 * 32 * RSO;
    __shared__ __attribute__((aligned(16))) char smem[LDS_IN > LDS_OUT ? LDS_IN : LDS_OUT];
    __bf16 *patch = (__bf16 *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c32 = lane & 31, g = lane >> 5;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const long long b = bid / p.tiles_y;
    const int oh0 = ty * TH, ow0 = tx * TW;
    const int ih0 = oh0 * S - G::PAD, iw0 = ow0 * S - G::PAD;

    // weights: one A fragment per (k step, 32-cout column)
    bf16x8 wf[G::KSTEPS][NREP];
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks)
#pragma unroll
        for (int n = 0; n < NREP; ++n) wf[ks][n] = p.w[(ks * NREP + n) * 64 + lane];

    // stage the patch: fp32 -> bf16, zeros outside the image.  All loads of a thread are issued before the first
    // LDS write (clamped address + select instead of a branch), so the tile pays one memory latency, not one per element.
    const float *img = p.img + b * 3 * (long long)p.H * p.W;
    constexpr int NE = (G::PATCH + 255) / 256;
    float pv[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = min(tid + u * 256, G::PATCH - 1);
        const int c = e % G::CW, rr = e / G::CW;
        const int r = rr % G::R, ch = rr / G::R;
        const int ih = ih0 + r, iw = iw0 + c;
        const bool inb = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const float v = img[((long long)ch * p.H + min(max(ih, 0), p.H - 1)) * p.W + min(max(iw, 0), p.W - 1)];
        pv[u] = inb ? v : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = tid + u * 256;
        if (e < G::PATCH) patch[e] = (__bf16)pv[u];
    }
    if (tid < 8) patch[G::PATCH + tid] = (__bf16)0.0f;
    __syncthreads();

    f32x16 acc[2][NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 bv = *(const f32x4 *)(p.bias + n * 32 + 8 * i + 4 * g);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][n][4 * i + j] = bv[j];
        }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int row = wave * 2 + m;
        const int base = (row * S) * G::CW + c32 * S;          // patch origin of this lane's output pixel
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks) {
            bf16x8 a;
            if constexpr (KS == 6) {
                // k = (ch*6 + ky)*6 + kx with kx fastest: (k, k+1) for even k are neighbours in a patch row, and the
                // row pitch, the stride and the pixel origin are all even, so each pair is one aligned 32-bit read
                u32x4 av;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const int o0 = G::off(ks * 16 + j), o1 = G::off(ks * 16 + 8 + j);
                    const int idx = g ? (o1 < 0 ? G::PATCH : base + o1) : (o0 < 0 ? G::PATCH : base + o0);
                    av[j / 2] = *(const unsigned *)(patch + idx);
                }
                a = __builtin_bit_cast(bf16x8, av);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int o0 = G::off(ks * 16 + j), o1 = G::off(ks * 16 + 8 + j);
                    const int idx = g ? (o1 < 0 ? G::PATCH : base + o1) : (o0 < 0 ? G::PATCH : base + o0);
                    a[j] = patch[idx];
                }
            }
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][n], a, acc[m][n], 0, 0, 0);
        }
    }
    __syncthreads();                                            // the patch is dead: LDS becomes the output tile

    char *wreg = smem + wave * (32 * RSO);
    constexpr int PPR = NREP * 4;                               // 16-byte pieces per pixel
    constexpr int PIXPI = 64 / PPR;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)apply_act(acc[m][n][4 * i + j], p.act);
                *(bf16x4 *)(wreg + c32 * RSO + (n * 32 + 8 * i + 4 * g) * 2) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int oh = oh0 + wave * 2 + m;
#pragma unroll
        for (int it = 0; it < 32 / PIXPI; ++it) {
            const int r = it * PIXPI + lane / PPR, piece = lane % PPR;
            const uint4 v = *(const uint4 *)(wreg + r * RSO + piece * 16);
            const int ow = ow0 + r, co = piece * 8;
            if (oh < p.Hout && ow < p.Wout && co < p.Cout)
                *(uint4 *)(p.y + ((b * p.Hout + oh) * p.Wout + ow) * p.y_cs + co) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int KS, int S>
int launch_stem(const StemParams &p, int batch, hipStream_t st)
{
    const long long grid = (long long)batch * p.tiles_y * p.tiles_x;
    YAMD_REQUIRE(grid > 0 && grid < (1ll << 31), "stem grid out of range");
    if (p.Cout <= 32) hipLaunchKernelGGL((stem_conv_kernel<KS, S, 1>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((stem_conv_kernel<KS, S, 2>), dim3((unsigned)grid), dim3(256), 0, st, p);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

bool stem_mode(int mode, int &ks, int &s)
{
    switch (mode) {
    case YAMD_STEM_K3S1: ks = 3; s = 1; return true;
    case YAMD_STEM_K3S2: ks = 3; s = 2; return true;
    case YAMD_STEM_FOCUS: ks = 6; s = 2; return true;
    default: return false;
    }
}

}  // namespace

extern "C" int64_t yamd_stem_packed_elems(int mode, int cout)
{
    int ks, s;
    if (!stem_mode(mode, ks, s) || cout <= 0 || cout > 64) return 0;
    const int ksteps = (3 * ks * ks + 15) / 16, nrep = cout <= 32 ? 1 : 2;
    return (int64_t)ksteps * nrep * 64 * 8;
}

extern "C" int yamd_stem_pack_weights(int mode, const float *w, const float *bias, int cout, void *w_packed, float *bias_packed)
{
    YAMD_REQUIRE(w && w_packed && bias_packed, "null argument");
    int ks, s;
    YAMD_REQUIRE(stem_mode(mode, ks, s), "unknown stem mode %d", mode);
    YAMD_REQUIRE(cout > 0 && cout <= 64 && cout % 8 == 0, "stem cout=%d (1..64, multiple of 8)", cout);
    const int K = 3 * ks * ks, ksteps = (K + 15) / 16, nrep = cout <= 32 ? 1 : 2;
    __bf16 *out = (__bf16 *)w_packed;
    for (int kk = 0; kk < ksteps; ++kk)
        for (int n = 0; n < nrep; ++n)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = n * 32 + (lane & 31), k = kk * 16 + (lane >> 5) * 8 + j;
                    float v = 0.0f;
                    if (co < cout && k < K) {
                        const int ch = k / (ks * ks), rem = k % (ks * ks), ky = rem / ks, kx = rem % ks;
                        if (mode == YAMD_STEM_FOCUS) {
                            // Focus channel q*3+ch holds image pixel (2Y + dy, 2X + dx), q = dy + 2*dx; tap (ky', kx') of
                            // the 3x3 conv on it is tap (2ky' + dy, 2kx' + dx) of the 6x6 stride-2 conv on the image
                            const int q = (ky & 1) + 2 * (kx & 1);
                            v = w[((size_t)co * 12 + q * 3 + ch) * 9 + (ky >> 1) * 3 + (kx >> 1)];
                        } else {
                            v = w[((size_t)co * 3 + ch) * 9 + ky * 3 + kx];
                        }
                    }
                    out[(((size_t)kk * nrep + n) * 64 + lane) * 8 + j] = (__bf16)v;
                }
    for (int c = 0; c < nrep * 32; ++c) bias_packed[c] = (bias && c < cout) ? bias[c] : 0.0f;
    return YAMD_OK;
}

extern "C" int yamd_stem_conv(int mode, const float *images, int batch, int h, int w, const void *w_packed,
                              const float *bias_packed, int cout, int act, void *y, int y_cstride, void *stream)
{
    YAMD_REQUIRE(images && w_packed && bias_packed && y, "null argument");
    int ks, s;
    YAMD_REQUIRE(stem_mode(mode, ks, s), "unknown stem mode %d", mode);
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0 && (s == 1 || (h % 2 == 0 && w % 2 == 0)), "bad image shape %dx%dx%d", batch, h, w);
    YAMD_REQUIRE(cout > 0 && cout <= 64 && cout % 8 == 0, "stem cout=%d (1..64, multiple of 8)", cout);
    YAMD_REQUIRE(act >= YAMD_ACT_NONE && act <= YAMD_ACT_RELU, "unknown activation %d", act);
    YAMD_REQUIRE(y_cstride >= cout && y_cstride % 8 == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 &&
                     ((uintptr_t)bias_packed % 16) == 0,
                 "stem output / weights need 16-byte alignment (cstride=%d)", y_cstride);
    StemParams p;
    p.img = images;
    p.w = (const bf16x8 *)w_packed;
    p.bias = bias_packed;
    p.y = (__bf16 *)y;
    p.H = h;
    p.W = w;
    const int pad = ks == 6 ? 2 : 1;
    p.Hout = (h + 2 * pad - ks) / s + 1;
    p.Wout = (w + 2 * pad - ks) / s + 1;
    p.Cout = cout;
    p.y_cs = y_cstride;
    p.act = act;
    p.tiles_x = ceil_div(p.Wout, TW);
    p.tiles_y = ceil_div(p.Hout, TH);
    hipStream_t st = (hipStream_t)stream;
    if (mode == YAMD_STEM_K3S1) return launch_stem<3, 1>(p, batch, st);
    if (mode == YAMD_STEM_K3S2) return launch_stem<3, 2>(p, batch, st);
    return launch_stem<6, 2>(p, batch, st);
}
