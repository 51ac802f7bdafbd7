// First conv of a network straight from the fp32 NCHW image: 3x3 stride 1 / stride 2 on the 3 colour channels, or the
// YOLOv5 Focus (space-to-depth) + 3x3 conv folded into one 6x6 stride-2 conv.  No padded NHWC copy of the image is
// ever written: a workgroup stages the fp32 patch of its output tile into LDS as bf16 (the rounding the rest of the
// graph sees too), every lane gathers the K = 27 (108) patch values of its output pixel from there as the MFMA B
// operand, and the weights live in registers as the A operand (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// Bias + activation, then the tile leaves through LDS as whole NHWC pixel rows.
#include "yamd_conv_kernel.h"

#include <map>
#include <mutex>
#include <utility>

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

// ACTK >= 0: the activation is that kind, fixed at compile time; ACTK < 0: it is looked up per value (a chain of scalar
// compares and branches for every four outputs: 351 branches in the 3x3 form)
template <int KS, int S, int NREP, int ACTK>
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
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)apply_act(acc[m][n][4 * i + j], ACTK >= 0 ? ACTK : p.act);
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
    auto launch = [&](auto actk) {
        constexpr int A = decltype(actk)::value;
        if (p.Cout <= 32) hipLaunchKernelGGL((stem_conv_kernel<KS, S, 1, A>), dim3((unsigned)grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((stem_conv_kernel<KS, S, 2, A>), dim3((unsigned)grid), dim3(256), 0, st, p);
    };
    switch (p.act) {
    case YAMD_ACT_SILU: launch(std::integral_constant<int, YAMD_ACT_SILU>{}); break;
    case YAMD_ACT_LEAKY: launch(std::integral_constant<int, YAMD_ACT_LEAKY>{}); break;
    case YAMD_ACT_RELU: launch(std::integral_constant<int, YAMD_ACT_RELU>{}); break;
    case YAMD_ACT_MISH: launch(std::integral_constant<int, YAMD_ACT_MISH>{}); break;
    default: launch(std::integral_constant<int, -1>{}); break;
    }
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The first TWO convs of a backbone in one kernel: the stem above (stride 2, or 3x3 stride 1 with 32 output channels), then
// a 3x3 stride-2 conv (pad 1) on its output.
// The stem's output (the largest activation of the network) is never written to memory: a workgroup computes the
// stem on the (2*4+1) x (2*16+1) pixels that its 4 x 16 tile of the second conv needs, keeps them in LDS as bf16 (the
// same rounding the unfused path applies when it stores them) and runs the second conv from there.  Pixels of that
// region outside the stem's output are the second conv's zero padding.  The stem is recomputed on the overlap between
// neighbouring tiles (297 pixels for 256 needed: +16 %).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int F_TH = 4, F_TW = 16;                          // tile of the second conv's output: 2 fragments of 2 rows x 16
constexpr int F_IR = 2 * F_TH + 1, F_IC = 2 * F_TW + 1;     // stem pixels under the tile
constexpr int F_IPIX = F_IR * F_IC;
constexpr int F_IFRAG = (F_IPIX + 31) / 32;
constexpr int F_HALF = (F_IC + 1) / 2;                      // a row of stem pixels is stored even columns first, then odd:
constexpr int F_CROW = 2 * F_HALF;                          // the stride-2 taps of neighbouring outputs are neighbours in LDS

// Tile counters of the fused launches: 256 pairs {next tile, workgroups done} per GPU.  A pair belongs to one stream of
// that GPU (stream_ticket): launches on one stream run one after the other, and the last workgroup of a launch to finish
// puts both words back to zero, so a pair is zero whenever a launch starts and nothing but the kernel ever writes it (no
// memset on the stream: a recorded step holds kernel nodes only).  Launches on different streams use different pairs, so
// they may overlap; only when more than 256 streams of one GPU have launched do streams share a pair, and such streams
// must not run this kernel at the same time.  A step recorded into a HIP graph keeps the pair of the stream it was
// recorded on: do not launch the graph while that stream runs fused-stem launches of its own.
// The array's name is a host-side handle only; every device has its own copy at its own address.
constexpr int TICKET_RING = 256;
constexpr int TICKET_MAX_DEVICES = 64;
std::atomic<unsigned *> g_ticket_base[TICKET_MAX_DEVICES];
int stream_ticket(hipStream_t st, unsigned **out);

template <int KS, int S>
struct FusedGeom {
    static constexpr int PAD = KS == 6 ? 2 : 1;
    static constexpr int XO = (PAD + S) & 1;                       // the patch starts XO columns early: at an even image column
    static constexpr int R = (F_IR - 1) * S + KS;                  // image patch rows
    static constexpr int CW = ((F_IC - 1) * S + KS + XO + 1) / 2 * 2;   // patch columns, even
    static constexpr int K = 3 * KS * KS;
    static constexpr int KSTEPS = (K + 15) / 16;
    static constexpr int PATCH = 3 * R * CW;
    static constexpr int off(int k)
    {
        if (k >= K) return -1;
        const int ch = k / (KS * KS), rem = k % (KS * KS);
        return (ch * R + rem / KS) * CW + rem % KS;
    }
};

// Focus stems (KS == 6): the image patch under a tile is staged in space-to-depth form, the way the reference's Focus
// layer slices it: pixel (Y, X) of the half-resolution image holds the 12 channels q * 3 + ch = image pixel
// (2Y + dy, 2X + dx) of channel ch, q = dy + 2 dx, as 32 bytes (channels 12 - 15 zero).  The 6 x 6 stride-2 conv is the
// 3 x 3 pad-1 conv on that image, so a k-step of the stem is one tap and a lane's B operand one 16-byte read (channels
// 8g .. 8g + 7 of its pixel's neighbour) instead of four 4-byte gathers with their address arithmetic; 9 k-steps instead
// of 7.  The two 16-byte halves of a pixel swap places in every second run of 8 columns, which spreads the reads of 16
// neighbouring pixels over all banks.
constexpr int F_SR = F_IR + 2, F_SC = F_IC + 2;             // half-resolution pixels under the tile's stem pixels
constexpr int F_SPIX = F_SR * F_SC;
constexpr int F_STAPS = 9;
__device__ __forceinline__ int s2d_offset(int y, int x, int half) { return (y * F_SC + x) * 32 + ((half ^ ((x >> 3) & 1)) << 4); }

struct FusedParams {
    const float *img;
    const bf16x8 *w1;      // stem weights, [kstep][ncol][lane] fragments (Focus: the tap-major fragments behind them)
    const float *b1;
    const char *w2;        // second conv, [tap][kstep][ncol] fragments of 1 KB
    const float *b2;
    __bf16 *y;
    int H, W, H1, W1, H2, W2, C2, y_cs, act1, act2, tiles_x, tiles_y, total_tiles;
    unsigned *ticket;      // ticket[0]: tiles after a workgroup's first are handed out by this counter; ticket[1]: workgroups
                           // that have finished.  Both zero at launch; the last workgroup to finish zeroes them again
};

// ACTK >= 0: both activations are that kind, fixed at compile time (SiLU for v5 / v8, ReLU for the re-parameterised v6
// stems); ACTK < 0: they are looked up per value
template <int KS, int S, int NREP1, int NREP2, int ACTK>
__global__ __launch_bounds__(256, 2) void stem_conv2_kernel(FusedParams p)
{
    using G = FusedGeom<KS, S>;
    constexpr int C1 = NREP1 * 32;
    constexpr int KS2 = C1 / 16;                           // k-steps per tap of the second conv
    constexpr int RS2 = C1 * 2 + 16;                       // bytes per stem pixel in LDS (padded: conflict-free 16-byte reads)
    constexpr int RSO = NREP2 * 64 + 16;                   // bytes per pixel of the output staging tile
    constexpr bool S2D = KS == 6;
    constexpr int PATCH_BYTES = S2D ? F_SPIX * 32 : ((G::PATCH + 8) * 2 + 15) / 16 * 16;
    constexpr int STAGE_BYTES = 64 * RSO;
    constexpr int A_BYTES = PATCH_BYTES > STAGE_BYTES ? PATCH_BYTES : STAGE_BYTES;   // patch, later the output tile
    constexpr int I_BYTES = F_IR * F_CROW * RS2;
    constexpr int W2_BYTES = 9 * KS2 * NREP2 * 1024;
    constexpr int NQ = (NREP2 + 1) / 2;                    // cout fragments per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16 *patch = (__bf16 *)smem;
    char *inter = smem + A_BYTES;
    char *w2s = inter + I_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c32 = lane & 31, g = lane >> 5;

    // second-conv weights: resident in LDS for all tiles of this workgroup
    for (int q = tid; q < W2_BYTES / 16; q += 256) *(u32x4 *)(w2s + q * 16) = *(const u32x4 *)(p.w2 + q * 16);
    // stem weights: one A fragment per (k step, 32-cout column), in registers
    constexpr int KST = S2D ? F_STAPS : G::KSTEPS;
    bf16x8 wf[KST][NREP1];
#pragma unroll
    for (int ks = 0; ks < KST; ++ks)
#pragma unroll
        for (int n = 0; n < NREP1; ++n) wf[ks][n] = p.w1[((S2D ? G::KSTEPS : 0) * NREP1 + ks * NREP1 + n) * 64 + lane];

    // biases in registers for all tiles
    f32x4 bias1[NREP1][4], bias2[NQ][4];
#pragma unroll
    for (int n = 0; n < NREP1; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias1[n][i] = *(const f32x4 *)(p.b1 + n * 32 + 8 * i + 4 * g);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bias2[q][i] = *(const f32x4 *)(p.b2 + min((wave >> 1) + 2 * q, NREP2 - 1) * 32 + 8 * i + 4 * g);
    const int act1 = ACTK >= 0 ? ACTK : p.act1, act2 = ACTK >= 0 ? ACTK : p.act2;

    // The patch is loaded two columns at a time: its first column is even in the image (XO below) and so is its width,
    // so a pair is 8-byte aligned and lies inside the image or outside it as a whole.  Focus form: a thread takes whole
    // half-resolution pixels (2 rows x 2 columns x 3 channels = six pairs each), which lie inside or outside as a whole.
    constexpr int NE = S2D ? 6 * ((F_SPIX + 255) / 256) : (G::PATCH / 2 + 255) / 256;
    f32x2 pv[NE];
    // pair of this thread -> (channel, row, first column), the same for every tile (Focus: pixel -> (row, column))
    int prc[S2D ? NE / 6 : NE];
    if constexpr (S2D) {
#pragma unroll
        for (int u = 0; u < NE / 6; ++u) {
            const int e = min(tid + u * 256, F_SPIX - 1);
            prc[u] = ((e / F_SC) << 10) | (e % F_SC);
        }
    } else {
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = 2 * min(tid + u * 256, G::PATCH / 2 - 1);
            const int c = e % G::CW, rr = e / G::CW;
            prc[u] = ((rr / G::R) << 20) | ((rr % G::R) << 10) | c;
        }
    }
    // fp32 patch of tile `t` into registers: clamped addresses and a select, no branch; one memory latency per tile
    auto load_patch = [&](int t) __attribute__((always_inline)) {
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const long long b = t / p.tiles_y;
        const float *img = p.img + b * 3 * (long long)p.H * p.W;
        if constexpr (S2D) {
            const int y0 = 2 * ty * F_TH - 2, x0 = 2 * tx * F_TW - 2;      // half-resolution pixel (0, 0) of the patch
#pragma unroll
            for (int u = 0; u < NE / 6; ++u) {
                const int sy = y0 + (prc[u] >> 10), sx = x0 + (prc[u] & 1023);
                const bool inb = sy >= 0 && 2 * sy < p.H && sx >= 0 && 2 * sx < p.W;
                const float *src = img + 2 * min(max(sy, 0), p.H / 2 - 1) * p.W + 2 * min(max(sx, 0), p.W / 2 - 1);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const f32x2 v = *(const f32x2 *)(src + (ch * p.H + dy) * p.W);
                        pv[u * 6 + ch * 2 + dy] = inb ? v : f32x2{0.0f, 0.0f};
                    }
            }
        } else {
            const int ih0 = (2 * ty * F_TH - 1) * S - G::PAD, iw0 = (2 * tx * F_TW - 1) * S - G::PAD - G::XO;
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int ch = prc[u] >> 20, ih = ih0 + ((prc[u] >> 10) & 1023), iw = iw0 + (prc[u] & 1023);
                const bool inb = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                const f32x2 v = *(const f32x2 *)(img + (ch * p.H + min(max(ih, 0), p.H - 1)) * p.W + min(max(iw, 0), p.W - 2));
                pv[u] = inb ? v : f32x2{0.0f, 0.0f};
            }
        }
    };

    // the wave's role in the second conv: pixel fragment wm (rows 2wm, 2wm+1 of the tile), cout fragments wn, wn+2
    const int wm = wave & 1, wn = wave >> 1;
    const int pr = wm * 2 + (c32 >> 4), pc = c32 & 15;
    const int pixbase = ((2 * pr) * F_CROW + pc) * RS2 + g * 16;

    // A workgroup's first tile is its index; further tiles are handed out by a counter, so a workgroup that shares
    // its compute unit with other work (a copy kernel, another stream) takes fewer tiles instead of holding the launch
    // up with a fixed share.  Thread 0 draws the ticket one tile ahead, so its latency is never waited for.
    int *s_next = (int *)(w2s + W2_BYTES);
    int tile = blockIdx.x;
    int ticket = 0;
    if (tile < p.total_tiles) {
    if (tid == 0) ticket = (int)gridDim.x + (int)atomicAdd(p.ticket, 1u);
    load_patch(tile);
    for (;;) {
        int t = tile;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const long long b = t / p.tiles_y;
        const int oh0 = ty * F_TH, ow0 = tx * F_TW;
        const int ir0 = 2 * oh0 - 1, ic0 = 2 * ow0 - 1;      // first stem pixel under the tile

        if constexpr (S2D) {
#pragma unroll
            for (int u = 0; u < NE / 6; ++u) {
                if (tid + u * 256 < F_SPIX) {
                    // channel q * 3 + ch, q = dy + 2 dx, from pv[ch * 2 + dy][dx]
                    bf16x8 lo, hi;
#pragma unroll
                    for (int f = 0; f < 16; ++f) {
                        const int q = f / 3, ch = f % 3;
                        const __bf16 v = f < 12 ? (__bf16)pv[u * 6 + ch * 2 + (q & 1)][q >> 1] : (__bf16)0.0f;
                        if (f < 8) lo[f] = v;
                        else hi[f - 8] = v;
                    }
                    const int y = prc[u] >> 10, x = prc[u] & 1023;
                    *(bf16x8 *)((char *)patch + s2d_offset(y, x, 0)) = lo;
                    *(bf16x8 *)((char *)patch + s2d_offset(y, x, 1)) = hi;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int e = tid + u * 256;
                if (e < G::PATCH / 2) {
                    bf16x2 o;
                    o[0] = (__bf16)pv[u][0];
                    o[1] = (__bf16)pv[u][1];
                    *(bf16x2 *)(patch + 2 * e) = o;
                }
            }
            if (tid < 8) patch[G::PATCH + tid] = (__bf16)0.0f;
        }
        if (tid == 0) *s_next = ticket;
        __syncthreads();
        const int next = *s_next;
        if (tid == 0) ticket = (int)gridDim.x + (int)atomicAdd(p.ticket, 1u);
        if (next < p.total_tiles) load_patch(next);      // in flight during this tile's MFMAs

        // ---- stem on the 9 x 33 pixels under the tile, 32 at a time; a wave takes its first two fragments together (their
        // operand reads, MFMAs and activations in one block of code, so that one's latencies lie under the other's work)
        auto stem_frags = [&](auto nf_tag, const int f0) __attribute__((always_inline)) {
            constexpr int NF = decltype(nf_tag)::value;
            int s_raw[NF], r[NF], c[NF];
            [[maybe_unused]] int base[NF];
            f32x16 acc[NF][NREP1];
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                s_raw[u] = (f0 + 4 * u) * 32 + c32;
                const int sp = min(s_raw[u], F_IPIX - 1);
                r[u] = sp / F_IC;
                c[u] = sp - r[u] * F_IC;
                base[u] = (r[u] * S) * G::CW + c[u] * S + G::XO;   // patch origin of this lane's stem pixel
#pragma unroll
                for (int n = 0; n < NREP1; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[u][n][4 * i + j] = bias1[n][i][j];
            }
            if constexpr (S2D && NREP1 == 1) {
                // a row of taps is requested while the MFMAs of the row before run (the compiler, left alone, asks for one
                // k-step at a time and waits for it: nine LDS latencies in a row); two register buffers
                bf16x8 a[2][NF][3];
                auto fetch = [&](const int row, const int buf) __attribute__((always_inline)) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int u = 0; u < NF; ++u)   // k-step = tap (row, kx) of the 3x3 conv on the half-resolution image
                            a[buf][u][kx] = *(const bf16x8 *)((const char *)patch + s2d_offset(r[u] + row, c[u] + kx, g));
                };
                fetch(0, 0);
#pragma unroll
                for (int row = 0; row < 3; ++row) {
                    if (row < 2) fetch(row + 1, (row + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int u = 0; u < NF; ++u)
#pragma unroll
                            for (int n = 0; n < NREP1; ++n)
                                acc[u][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[row * 3 + kx][n], a[row & 1][u][kx], acc[u][n], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    bf16x8 a[NF];
#pragma unroll
                    for (int u = 0; u < NF; ++u) {
                        if constexpr (S2D) {
                            a[u] = *(const bf16x8 *)((const char *)patch + s2d_offset(r[u] + ks / 3, c[u] + ks % 3, g));
                            continue;
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int o0 = G::off(ks * 16 + j), o1 = G::off(ks * 16 + 8 + j);
                            const int idx = g ? (o1 < 0 ? G::PATCH : base[u] + o1) : (o0 < 0 ? G::PATCH : base[u] + o0);
                            a[u][j] = patch[idx];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NF; ++u)
#pragma unroll
                        for (int n = 0; n < NREP1; ++n)
                            acc[u][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][n], a[u], acc[u][n], 0, 0, 0);
                }
            }
            // activation, bf16, into the stem image; pixels outside the stem's output are the second conv's padding
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                const int ir = ir0 + r[u], ic = ic0 + c[u];
                const bool real = ir >= 0 && ir < p.H1 && ic >= 0 && ic < p.W1;
                if (s_raw[u] < F_IPIX) {
                    char *dst = inter + (r[u] * F_CROW + (c[u] & 1) * F_HALF + (c[u] >> 1)) * RS2;
#pragma unroll
                    for (int n = 0; n < NREP1; ++n)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            bf16x4 o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (__bf16)(real ? apply_act(acc[u][n][4 * i + j], act1) : 0.0f);
                            *(bf16x4 *)(dst + (n * 32 + 8 * i + 4 * g) * 2) = o;
                        }
                }
            }
        };
        static_assert(F_IFRAG > 8 && F_IFRAG <= 12, "a wave's fragments: two together, at most one more");
        if constexpr (NREP1 == 1) {
            stem_frags(std::integral_constant<int, 2>{}, wave);
        } else {                                          // 64 stem channels: two fragments' accumulators do not fit
            stem_frags(std::integral_constant<int, 1>{}, wave);
            stem_frags(std::integral_constant<int, 1>{}, wave + 4);
        }
        if (wave + 8 < F_IFRAG) stem_frags(std::integral_constant<int, 1>{}, wave + 8);
        __syncthreads();

        // ---- second conv: 3x3 stride 2 from the stem image
        f32x16 acc2[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[q][4 * i + j] = bias2[q][i][j];
        // operands are requested a group of k-steps ahead of the MFMAs that use them (two register buffers)
        constexpr int NS = 9 * KS2, GS = (NQ == 1 && NREP1 == 1) ? 3 : 1, NG = (NS + GS - 1) / GS;
        bf16x8 pa[2][GS], pw[2][GS][NQ];
        auto fetch = [&](const int grp, const int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int st = grp * GS + i;
                if (st < NS) {
                    const int tap = st / KS2, ks = st % KS2, ky = tap / 3, kx = tap % 3;
                    const int tapoff = (ky * F_CROW + (kx & 1) * F_HALF + (kx >> 1)) * RS2;
                    pa[buf][i] = *(const bf16x8 *)(inter + pixbase + tapoff + ks * 32);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int n = min(wn + 2 * q, NREP2 - 1);
                        pw[buf][i][q] = *(const bf16x8 *)(w2s + ((tap * KS2 + ks) * NREP2 + n) * 1024 + lane * 16);
                    }
                }
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 1 < NG) fetch(grp + 1, (grp + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GS; ++i)
                if (grp * GS + i < NS) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        acc2[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pw[grp & 1][i][q], pa[grp & 1][i], acc2[q], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- output tile through LDS (the patch is dead since the last barrier), then whole pixel rows to memory
        char *stage = smem;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n = wn + 2 * q;
            if (n < NREP2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (__bf16)apply_act(acc2[q][4 * i + j], act2);
                    *(bf16x4 *)(stage + (wm * 32 + c32) * RSO + (n * 32 + 8 * i + 4 * g) * 2) = o;
                }
            }
        }
        __syncthreads();
        constexpr int PPR = NREP2 * 4;                       // 16-byte pieces per pixel
        for (int e = tid; e < 64 * PPR; e += 256) {
            const int fp = e / PPR, piece = e % PPR;
            const int oh = oh0 + fp / F_TW, ow = ow0 + fp % F_TW, co = piece * 8;
            if (oh < p.H2 && ow < p.W2 && co < p.C2)
                *(uint4 *)(p.y + ((b * p.H2 + oh) * p.W2 + ow) * p.y_cs + co) = *(const uint4 *)(stage + fp * RSO + piece * 16);
        }
        __syncthreads();                                      // the next tile's patch overwrites the output tile
        if (next >= p.total_tiles) break;
        tile = next;
    }
    }
    // Every workgroup reports in once all its draws from the tile counter have returned (the release orders them before
    // the report); the one whose report is the last of the grid zeroes both words for the next launch on this stream.
    if (tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(p.ticket + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.ticket + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int KS, int S, int NREP1, int NREP2, int ACTK>
int launch_stem_conv2_act(const FusedParams &p, hipStream_t st)
{
    using G = FusedGeom<KS, S>;
    constexpr int C1 = NREP1 * 32, RS2 = C1 * 2 + 16, RSO = NREP2 * 64 + 16;
    constexpr int PATCH_BYTES = KS == 6 ? F_SPIX * 32 : ((G::PATCH + 8) * 2 + 15) / 16 * 16, STAGE_BYTES = 64 * RSO;
    constexpr int A_BYTES = PATCH_BYTES > STAGE_BYTES ? PATCH_BYTES : STAGE_BYTES;
    constexpr size_t LDS = (size_t)A_BYTES + (size_t)F_IR * F_CROW * RS2 + (size_t)9 * (C1 / 16) * NREP2 * 1024 + 16;
    static_assert(LDS <= 160 * 1024, "fused stem tile does not fit LDS");
    auto kern = stem_conv2_kernel<KS, S, NREP1, NREP2, ACTK>;
    {
        const int rc = KernelAttr<&stem_conv2_kernel<KS, S, NREP1, NREP2, ACTK>, 160 * 1024>::ensure();
        if (rc != YAMD_OK) return rc;
    }
    // persistent grid: the workgroups resident at once share the tiles; each keeps the second conv's weights in LDS
    int device = 0, cus = 0, per_cu = 0;
    YAMD_REQUIRE(hipGetDevice(&device) == hipSuccess &&
                     hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
                     hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, 256, LDS) == hipSuccess && cus > 0,
                 "occupancy query failed");
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)cus * per_cu;
    if (grid > p.total_tiles) grid = p.total_tiles;
    FusedParams q = p;
    {
        const int rc = stream_ticket(st, &q.ticket);
        if (rc != YAMD_OK) return rc;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, st, q);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

template <int KS, int S, int NREP1, int NREP2>
int launch_stem_conv2(const FusedParams &p, hipStream_t st)
{
    if (p.act1 == YAMD_ACT_SILU && p.act2 == YAMD_ACT_SILU) return launch_stem_conv2_act<KS, S, NREP1, NREP2, YAMD_ACT_SILU>(p, st);
    if (p.act1 == YAMD_ACT_RELU && p.act2 == YAMD_ACT_RELU) return launch_stem_conv2_act<KS, S, NREP1, NREP2, YAMD_ACT_RELU>(p, st);
    return launch_stem_conv2_act<KS, S, NREP1, NREP2, -1>(p, st);
}

}  // namespace

static __device__ __align__(256) unsigned stem_tickets[2 * TICKET_RING];

namespace {
int ticket_base(int *dev_out, unsigned **base_out)
{
    int dev = 0;
    YAMD_HIP(hipGetDevice(&dev));
    YAMD_REQUIRE(dev >= 0 && dev < TICKET_MAX_DEVICES, "device ordinal %d outside [0, %d)", dev, TICKET_MAX_DEVICES);
    unsigned *base = g_ticket_base[dev].load(std::memory_order_acquire);
    if (base == nullptr) {
        void *ptr = nullptr;
        YAMD_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(stem_tickets)));   // this device's copy (the device is current)
        YAMD_REQUIRE(ptr != nullptr, "tile counters have no device address");
        base = (unsigned *)ptr;
        g_ticket_base[dev].store(base, std::memory_order_release);
    }
    *dev_out = dev;
    *base_out = base;
    return YAMD_OK;
}

// the counter pair of stream `st` on the current device: assigned on the stream's first launch, kept for the process
int stream_ticket(hipStream_t st, unsigned **out)
{
    int dev = 0;
    unsigned *base = nullptr;
    const int rc = ticket_base(&dev, &base);
    if (rc != YAMD_OK) return rc;
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int> slots;
    static int used[TICKET_MAX_DEVICES];
    std::lock_guard<std::mutex> lock(mu);
    auto it = slots.find({dev, st});
    if (it == slots.end()) it = slots.emplace(std::make_pair(dev, st), used[dev]++ % TICKET_RING).first;
    *out = base + 2 * it->second;
    return YAMD_OK;
}

}  // namespace

namespace yamd {
int stem_init_current_device(int device)
{
    (void)device;
    int dev = 0;
    unsigned *base = nullptr;
    return ticket_base(&dev, &base);
}
}  // namespace yamd

namespace {
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
    // Focus: the 7 k-steps in image order (yamd_stem_conv), then 9 tap-major k-steps (the fused kernel's space-to-depth form)
    return (int64_t)(ksteps + (mode == YAMD_STEM_FOCUS ? F_STAPS : 0)) * nrep * 64 * 8;
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
    if (mode == YAMD_STEM_FOCUS) {
        // k-step = tap of the reference's 3x3 conv on the 12 Focus channels, k within it = Focus channel (12 .. 15 zero)
        out += (size_t)ksteps * nrep * 64 * 8;
        for (int tap = 0; tap < F_STAPS; ++tap)
            for (int n = 0; n < nrep; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = n * 32 + (lane & 31), f = (lane >> 5) * 8 + j;
                        const float v = (co < cout && f < 12) ? w[((size_t)co * 12 + f) * 9 + tap] : 0.0f;
                        out[(((size_t)tap * nrep + n) * 64 + lane) * 8 + j] = (__bf16)v;
                    }
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

extern "C" int64_t yamd_stem_conv2_packed_elems(int c1, int c2)
{
    // the second conv's weights stay in LDS: 9 * (c1 / 16) KB per 32 couts
    if ((c1 != 32 && c1 != 64) || c2 <= 0 || c2 > (c1 == 32 ? 128 : 64) || c2 % 8 != 0) return 0;
    return (int64_t)9 * (c1 / 16) * ((c2 + 31) / 32) * 512;
}

extern "C" int yamd_stem_conv2_pack_weights(const float *w, const float *bias, int c1, int c2, void *w_packed, float *bias_packed)
{
    YAMD_REQUIRE(w && w_packed && bias_packed, "null argument");
    YAMD_REQUIRE(yamd_stem_conv2_packed_elems(c1, c2) > 0, "fused second conv: cin=%d (32 or 64), cout=%d (multiple of 8; <= 128 for cin 32, <= 64 for cin 64)", c1, c2);
    const int ks2 = c1 / 16, nrep = (c2 + 31) / 32;
    __bf16 *out = (__bf16 *)w_packed;
    for (int tap = 0; tap < 9; ++tap)
        for (int ks = 0; ks < ks2; ++ks)
            for (int n = 0; n < nrep; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = n * 32 + (lane & 31), ci = ks * 16 + (lane >> 5) * 8 + j;
                        const float v = co < c2 ? w[((size_t)co * c1 + ci) * 9 + tap] : 0.0f;
                        out[((((size_t)tap * ks2 + ks) * nrep + n) * 64 + lane) * 8 + j] = (__bf16)v;
                    }
    for (int c = 0; c < 128; ++c) bias_packed[c] = (bias && c < c2) ? bias[c] : 0.0f;
    return YAMD_OK;
}

extern "C" int yamd_stem_conv2(int mode, const float *images, int batch, int h, int w, const void *w1_packed,
                               const float *b1_packed, int c1, int act1, const void *w2_packed, const float *b2_packed,
                               int c2, int act2, void *y, int y_cstride, void *stream)
{
    YAMD_REQUIRE(images && w1_packed && b1_packed && w2_packed && b2_packed && y, "null argument");
    int ks, s;
    YAMD_REQUIRE(stem_mode(mode, ks, s), "unknown stem mode %d", mode);
    YAMD_REQUIRE(s == 2 || c1 == 32, "the fused form takes a stride-1 stem with 32 output channels only (got %d)", c1);
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "bad image shape %dx%dx%d", batch, h, w);
    YAMD_REQUIRE(yamd_stem_conv2_packed_elems(c1, c2) > 0, "fused second conv: cin=%d (32 or 64), cout=%d (multiple of 8; <= 128 for cin 32, <= 64 for cin 64)", c1, c2);
    YAMD_REQUIRE(act1 >= YAMD_ACT_NONE && act1 <= YAMD_ACT_RELU && act2 >= YAMD_ACT_NONE && act2 <= YAMD_ACT_RELU, "unknown activation");
    YAMD_REQUIRE(y_cstride >= c2 && y_cstride % 8 == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w1_packed % 16) == 0 &&
                     ((uintptr_t)w2_packed % 16) == 0 && ((uintptr_t)b1_packed % 16) == 0 && ((uintptr_t)b2_packed % 16) == 0,
                 "fused stem output / weights need 16-byte alignment (cstride=%d)", y_cstride);
    FusedParams p;
    p.ticket = nullptr;      // set per launch
    p.img = images;
    p.w1 = (const bf16x8 *)w1_packed;
    p.b1 = b1_packed;
    p.w2 = (const char *)w2_packed;
    p.b2 = b2_packed;
    p.y = (__bf16 *)y;
    p.H = h;
    p.W = w;
    const int pad = ks == 6 ? 2 : 1;
    p.H1 = (h + 2 * pad - ks) / s + 1;
    p.W1 = (w + 2 * pad - ks) / s + 1;
    p.H2 = (p.H1 - 1) / 2 + 1;
    p.W2 = (p.W1 - 1) / 2 + 1;
    p.C2 = c2;
    p.y_cs = y_cstride;
    p.act1 = act1;
    p.act2 = act2;
    p.tiles_x = ceil_div(p.W2, F_TW);
    p.tiles_y = ceil_div(p.H2, F_TH);
    const long long total = (long long)batch * p.tiles_y * p.tiles_x;
    YAMD_REQUIRE(total > 0 && total < (1ll << 31), "fused stem grid out of range");
    YAMD_REQUIRE(3ll * h * w < (1ll << 31) && w >= 2, "image too large or too narrow");
    YAMD_REQUIRE((long long)batch * p.H2 * p.W2 * y_cstride < (1ll << 62), "output too large");
    p.total_tiles = (int)total;
    hipStream_t st = (hipStream_t)stream;
    const int nrep2 = (c2 + 31) / 32;
    if (s == 1) {
        switch (nrep2) {
        case 1: return launch_stem_conv2<3, 1, 1, 1>(p, st);
        case 2: return launch_stem_conv2<3, 1, 1, 2>(p, st);
        case 3: return launch_stem_conv2<3, 1, 1, 3>(p, st);
        default: return launch_stem_conv2<3, 1, 1, 4>(p, st);
        }
    }
    if (c1 == 32) {
        switch (nrep2 * 10 + (ks == 6)) {
        case 11: return launch_stem_conv2<6, 2, 1, 1>(p, st);
        case 21: return launch_stem_conv2<6, 2, 1, 2>(p, st);
        case 31: return launch_stem_conv2<6, 2, 1, 3>(p, st);
        case 41: return launch_stem_conv2<6, 2, 1, 4>(p, st);
        case 10: return launch_stem_conv2<3, 2, 1, 1>(p, st);
        case 20: return launch_stem_conv2<3, 2, 1, 2>(p, st);
        case 30: return launch_stem_conv2<3, 2, 1, 3>(p, st);
        default: return launch_stem_conv2<3, 2, 1, 4>(p, st);
        }
    }
    if (ks == 6) return nrep2 == 1 ? launch_stem_conv2<6, 2, 2, 1>(p, st) : launch_stem_conv2<6, 2, 2, 2>(p, st);
    return nrep2 == 1 ? launch_stem_conv2<3, 2, 2, 1>(p, st) : launch_stem_conv2<3, 2, 2, 2>(p, st);
}
