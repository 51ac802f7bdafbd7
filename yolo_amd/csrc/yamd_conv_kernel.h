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
//
// Stride 2 with 16-channel chunks uses a compact image instead (ConvLds<...>::COMPACT): 32 bytes per pixel without
// padding, halo rows CROW pixels apart, and the two 16-byte halves of a pixel swapped where swz() says so.  The halo
// of a stride-2 tile is four times the tile, so the padding bytes decide whether two workgroups fit a compute unit;
// row pitch, even/odd column split and swizzle were found by exhaustive search over the lane groups of ds_read_b128
// (conflict free for every tap and fragment, both tile widths).
#pragma once
#include "yamd_common.h"

#include <cstring>
#include <cstdlib>
#include <cmath>
#include <type_traits>

namespace yamd {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
    const __bf16 *x;
    const char *w;
    const float *bias;
    const __bf16 *res;
    void *y;
    __bf16 *y2;                  // bf16 output only: couts >= split go to y2 (at channel cout - split), the others to y
    int x_cs, res_cs, y_cs, y2_cs, split;
    int H, W, Hout, Wout, Cout;
    int TH, tiles_x, tiles_y, n_tiles, nchunks;
    int act;
    int total_tiles;
    int ntw;                     // two-image form: cout tiles a workgroup walks (n_tiles: all of them; 1: one, the grid then
                                 // holds n_tiles workgroups per pixel tile, neighbours in the XCD's tile range, so that
                                 // they read the same activations through one L2)
    // stacked mode (3x3, bf16 output): the images of the batch are treated as ONE tall image, `vin` input rows and
    // `vout` output rows per image (the real rows followed by 1 or 2 rows of zero padding, which is exactly the
    // padding the neighbouring images need), so tiles are not confined to one image and small feature maps still
    // fill them.  0 = off.
    int vin, vout, vbatch;
    float res_scale;             // weight of the residual in the sum
    // 1x1 convs over a channel concatenation of two tensors (torch.cat((a, b), 1) feeding a conv) without building the
    // concatenation: chunks [0, nsplit) come from x, chunks [nsplit, nchunks) from x2.  half1 / half2: that source is
    // stored at half the resolution and read through nearest-neighbour upsampling (nn.Upsample(scale_factor=2) feeding
    // the cat), addressed from the pixel's position in the original oH x oW image.  Single source: x2 = nullptr,
    // nsplit = nchunks, half1 = half2 = 0.
    const __bf16 *x2;
    int x2_cs, nsplit, half1, half2, oH, oW;
#ifdef YAMD_CONV_TRACE
    unsigned long long *trace;   // 8 timestamps per workgroup (profiling builds only)
#endif
#ifdef YAMD_CONV_SKEW
    int skew, skew_blocks;       // experiment: workgroups of the first round that are not first on their SIMD start `skew` cycles late
#endif
};

#ifdef YAMD_CONV_TRACE
#ifdef YAMD_TRACE_REALTIME
#define YAMD_TRACE_CLOCK __builtin_amdgcn_s_memrealtime      // constant 100 MHz, the same on every compute unit
#else
#define YAMD_TRACE_CLOCK __builtin_readcyclecounter          // shader clock of the workgroup's own XCD
#endif
#define YAMD_STAMP(i)                                                                              \
    do {                                                                                           \
        if (threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = YAMD_TRACE_CLOCK();           \
    } while (0)
#else
#define YAMD_STAMP(i) \
    do {              \
    } while (0)
#endif

// Activations on the fp32 accumulator with hardware exp/rcp (about 1 ulp each); the result is rounded to
// bf16 (8 significant bits) or feeds raw logits only for YAMD_ACT_NONE, so this loses nothing.
__device__ __forceinline__ float apply_act(float v, int act)
{
    switch (act) {
    case YAMD_ACT_SILU:
        return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    case YAMD_ACT_LEAKY:
        return fmaxf(v, 0.1f * v);
    case YAMD_ACT_MISH: {
        // tanh(log(1+e^v)) = n / (n + 2) with n = e^v (e^v + 2); v for large v (torch's softplus threshold 20)
        const float e = __expf(fminf(v, 20.0f));
        const float n = e * (e + 2.0f);
        return v > 20.0f ? v : v * n * __builtin_amdgcn_rcpf(n + 2.0f);
    }
    case YAMD_ACT_RELU:
        return fmaxf(v, 0.0f);
    default:
        return v;
    }
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>); every array index
// inside f is a constant, so staging arrays stay in registers
// geometry of the LDS activation image shared by the kernel and its launcher
template <int KS, int S, int TW, int CK>
struct ConvLds {
    static constexpr int NPW = (TW - 1) * S + KS;            // halo width in input pixels
    static constexpr bool COMPACT = (S == 2 && CK == 16 && KS == 3 && (TW == 16 || TW == 20));
    static constexpr int NPWH = COMPACT ? (TW == 16 ? 17 : 21) : (NPW + 1) / 2;     // first odd column (stride-2 de-interleave)
    static constexpr int CROW = TW == 16 ? 34 : 42;          // compact image: pixels per halo row
    static constexpr int RS = CK * 2 + 16;                   // padded image: bytes per halo pixel
    static __host__ __device__ constexpr int image_bytes(int nph) { return COMPACT ? nph * CROW * 32 : nph * NPW * RS; }
    // compact image: 1 where the two 16-byte halves of the pixel at (halo row r, stored column c) are swapped
    static __device__ __forceinline__ int swz(int r, int c) { return TW == 16 ? ((c >> 2) & 1) : (((r * CROW + c) >> 3) & 1); }
};

template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int ACT>
__device__ __forceinline__ void act4(float (&v)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], ACT);
}

// STAGES = 1: one LDS image of the current chunk; the next chunk waits in registers and is written between two
// barriers; a workgroup computes one cout tile of one pixel tile.  STAGES = 2: two LDS images and a software pipeline:
// the staging (LDS stores of the next chunk, global loads of the one after) is spread between the MFMAs of the current
// chunk, a chunk costs one barrier, and a workgroup computes all cout tiles of its pixel tile in one stream.
template <int KS, int S, int TW, int MREP, int NREP, int CK, bool OUT_F32, int ACTK, int NW, int STAGES = 1>
__global__ __launch_bounds__(NW * 64, (NW == 4 && STAGES == 1) ? 2 : 1) void conv_igemm_kernel(ConvParams p)
{
    constexpr int NT = NW * 64;              // threads per workgroup (4 or 8 waves share one weight tile)
    constexpr int PAD = KS / 2;
    using Lds = ConvLds<KS, S, TW, CK>;
    constexpr int NPW = Lds::NPW;            // halo width in input pixels
    constexpr int NPWH = Lds::NPWH;          // first odd column (stride-2 de-interleave)
    constexpr int RS = Lds::RS;              // LDS bytes per halo pixel (padded image)
    constexpr bool COMPACT = Lds::COMPACT;
    constexpr int CROW = Lds::CROW;
    constexpr int G8 = CK / 8;               // 16-byte pieces per halo pixel
    constexpr int KSTEPS = CK / 16;
    constexpr int TAPS = KS * KS;
    constexpr int WBYTES = TAPS * KSTEPS * NREP * 1024;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Give every XCD a
    // contiguous range of logical tiles instead, so the cout tiles of one pixel tile (same input) and spatially
    // neighbouring tiles (shared halo) run on the same XCD at about the same time.
    const int per_xcd = gridDim.x >> 3;
    int bid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (bid >= p.total_tiles) return;
#ifdef YAMD_CONV_SKEW
    if ((int)blockIdx.x < p.skew_blocks) {
        // experiment: first-round workgroups start late, by their wave slot on the SIMD (skew > 0: de-phases the workgroups that
        // share a compute unit) or by their position in the dispatch order (skew < 0: de-phases the compute units)
        const unsigned slot = p.skew > 0 ? (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 15u) : ((blockIdx.x >> 3) & 7u);
        const unsigned long long t0 = __builtin_readcyclecounter();
        const unsigned long long wait = (unsigned long long)abs(p.skew) * slot;
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#endif
    YAMD_STAMP(0);
    // one image: a workgroup owns one cout tile of one pixel tile.  Two images: it owns the pixel tile and walks its
    // cout tiles one after the other (the chunk stream simply continues into the next cout tile's weights)
    const int ngrp = STAGES == 2 ? p.n_tiles / p.ntw : p.n_tiles;      // workgroups per pixel tile
    const int nt0 = (bid % ngrp) * (STAGES == 2 ? p.ntw : 1);         // first cout tile of this workgroup
    bid /= ngrp;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oh0 = ty * p.TH, ow0 = tx * TW;
    const int NPH = (p.TH - 1) * S + KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c32 = lane & 31, g = lane >> 5;

    // LDS byte offset of this lane's pixel (tap 0,0) for each 32-pixel fragment
    // compact image: pixoff is the byte offset of the pixel at tap (0,0) without the half select; bit (m*9 + tap) of
    // halfsel says which 16-byte half of that tap's pixel holds this lane's k-group
    static_assert(!COMPACT || MREP * TAPS <= 32, "one half-select bit per (fragment, tap)");
    int pixoff[MREP];
    unsigned halfsel = 0u;
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = wave * (32 * MREP) + m * 32 + c32;
        int pr = pp / TW, pc = pp % TW;
        if (pr >= p.TH) { pr = 0; pc = 0; }   // slot beyond the tile: read something valid, never stored
        if constexpr (COMPACT) {
            pixoff[m] = ((pr * S) * CROW + pc) * 32;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int r = pr * S + tap / KS, c = pc + ((tap % KS) & 1) * NPWH + ((tap % KS) >> 1);
                halfsel |= (unsigned)(g ^ Lds::swz(r, c)) << (m * TAPS + tap);
            }
        } else {
            pixoff[m] = ((pr * S) * NPW + pc) * RS + g * 16;
        }
    }

    // Staging registers: the next chunk's global loads are issued before the current chunk is computed
    // and written to LDS afterwards, so HBM/L2 latency hides under the MFMA work.
    constexpr int THMAX = (NW * 32 * MREP) / TW;
    constexpr int NPHMAX = (THMAX - 1) * S + KS;
    constexpr int AP = (NPHMAX * NPW * G8 + NT - 1) / NT;   // activation pieces per thread
    constexpr int WP = (WBYTES / 16 + NT - 1) / NT;         // weight pieces per thread
    u32x4 ra[AP], rw[WP];
    const int total = NPH * NPW * G8;

    // No divergent control flow in the staging: piece indices beyond the end are clamped (the duplicate
    // loads/stores are benign) and out-of-image pixels load a clamped address and are zeroed by a select.
    // The per-piece global offset (inside this image), LDS offset and in-image flag do not depend on the
    // channel chunk, so they are computed once.
    int goff[AP], loff[AP];
    int goff2[KS == 1 ? AP : 1];     // second source (1x1 convs over a concatenation)
    unsigned inb_mask = 0u;
    static_for<AP>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        const int q = min(tid + u * NT, total - 1);
        const int pix = q / G8, gc = q % G8;
        const int hr = pix / NPW, hc = pix % NPW;
        int ih = oh0 * S - PAD + hr;
        const int iw = ow0 * S - PAD + hc;
        bool inb = iw >= 0 && iw < p.W;
        int img_row0 = 0;                               // stacked mode: first row of the piece's image in the batch
        if (p.vin > 0) {
            const int vb = ih >= 0 ? ih / p.vin : -1;
            ih -= vb * p.vin;                           // row inside the image (or its zero rows)
            inb = inb && vb >= 0 && vb < p.vbatch;
            img_row0 = min(max(vb, 0), p.vbatch - 1) * p.H;
        }
        inb = inb && ih >= 0 && ih < p.H;
        const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
        goff[u] = ((img_row0 + ihc) * p.W + iwc) * p.x_cs + gc * 8;
        if constexpr (KS == 1) {
            goff2[u] = ((img_row0 + ihc) * p.W + iwc) * p.x2_cs + gc * 8;
            if (p.half1 | p.half2) {      // uniform
                // position of the pixel in the original images (the 1x1 launcher may have re-cut the batch into strips;
                // the pixel order is the same), then the pixel of the half-resolution tensor under it
                const int gp = (b * p.H + ihc) * p.W + iwc, hw = p.oH * p.oW;
                const int ob = gp / hw, rem = gp - ob * hw;
                const int oy = rem / p.oW, ox = rem - oy * p.oW;
                const int hp = (ob * (p.oH >> 1) + (oy >> 1)) * (p.oW >> 1) + (ox >> 1);
                if (p.half1) goff[u] = hp * p.x_cs + gc * 8;
                if (p.half2) goff2[u] = hp * p.x2_cs + gc * 8;
            }
        }
        const int lc = (S == 2) ? ((hc & 1) * NPWH + (hc >> 1)) : hc;
        if constexpr (COMPACT) loff[u] = (hr * CROW + lc) * 32 + ((gc ^ Lds::swz(hr, lc)) << 4);
        else loff[u] = (hr * NPW + lc) * RS + gc * 16;
        inb_mask |= inb ? (1u << u) : 0u;
    });
    // offsets into a half-resolution source count from the tensor's first element, the others from the image's
    const __bf16 *ximg = (KS == 1 && p.half1) ? p.x : p.x + (size_t)b * p.H * p.W * p.x_cs;
    const __bf16 *x2img = (KS == 1 && !p.half2 && p.x2) ? p.x2 + (size_t)b * p.H * p.W * p.x2_cs : p.x2;
    // piece u of the activation chunk c
    auto load_piece = [&](int c, int off1, int off2) __attribute__((always_inline)) -> u32x4 {
        if constexpr (KS == 1) {
            const bool second = c >= p.nsplit;      // uniform
            const __bf16 *xc = second ? x2img + (c - p.nsplit) * CK : ximg + c * CK;
            return *(const u32x4 *)(xc + (second ? off2 : off1));
        } else {
            return *(const u32x4 *)(ximg + c * CK + off1);
        }
    };
    const char *wtile = p.w + (size_t)nt0 * p.nchunks * WBYTES;
    auto issue_loads = [&](int ch) __attribute__((always_inline)) {
        const char *wsrc = wtile + (size_t)ch * WBYTES;
        static_for<WP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const int q = min(tid + u * NT, WBYTES / 16 - 1);
            rw[u] = *(const u32x4 *)(wsrc + q * 16);
        });
        static_for<AP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            ra[u] = load_piece(ch, goff[u], goff2[KS == 1 ? u : 0]);
        });
    };
    // one LDS image: weights, then the halo tile (multiple of 16).  With two images the epilogue of a cout tile runs
    // in the image that the next chunk does not read, so an image is at least as large as the epilogue tiles.
    constexpr int EPI_BYTES = NW * 32 * (OUT_F32 ? NREP * 128 + 16 : NREP * 64 + 16);
    const int stage_bytes = STAGES == 2 ? max(WBYTES + Lds::image_bytes(NPH), EPI_BYTES) : WBYTES + Lds::image_bytes(NPH);
    auto write_lds = [&](int stage) __attribute__((always_inline)) {
        char *ldsW = smem + stage * stage_bytes;
        char *ldsA = ldsW + WBYTES;
        static_for<WP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const int q = min(tid + u * NT, WBYTES / 16 - 1);
            *(u32x4 *)(ldsW + q * 16) = rw[u];
        });
        static_for<AP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            // out-of-image pieces were loaded from a clamped address; they become zeros here, after the chunk's MFMAs,
            // and not where the load is issued: a select placed there makes the wave wait for the load before it computes
            *(u32x4 *)(ldsA + loff[u]) = ((inb_mask >> u) & 1u) ? ra[u] : u32x4{0u, 0u, 0u, 0u};
        });
    };

    YAMD_STAMP(1);
    issue_loads(0);
    // accumulators start at the bias: register 4i+j of fragment n is cout n*32 + 8i + 4g + j
    f32x16 acc[MREP][NREP];
    auto init_acc = [&](int nt) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 bv = *(const f32x4 *)(p.bias + nt * (NREP * 32) + n * 32 + 8 * i + 4 * g);
#pragma unroll
                for (int m = 0; m < MREP; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][n][4 * i + j] = bv[j];
            }
    };
    init_acc(nt0);      // (the first chunk is in flight while this runs)

#ifdef YAMD_CONV_TRACE
    if constexpr (STAGES == 2) {            // where the first stage spends its time: the first chunk's loads, or the rest
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        YAMD_STAMP(6);
    }
#endif
    write_lds(0);
    // One chunk's MFMAs on the LDS image at byte offset `img`.  A slot is one k-step of one tap.  With two LDS images
    // the fragments of slot s+1 are read before the MFMAs of slot s are issued, staging(s) puts a share of the staging
    // between them, and a scheduling barrier after every slot keeps that order in the generated code.
    auto compute_chunk = [&](int img, auto staging) __attribute__((always_inline)) {
        const char *ldsW = smem + img;
        const char *ldsA = ldsW + WBYTES;
        // compact image: the per-tap half selects are re-derived from one register in every chunk; hoisted out of the
        // loop they would occupy one address register per (fragment, tap)
        unsigned hs = halfsel;
        if constexpr (COMPACT) asm volatile("" : "+v"(hs));
        auto read_frags = [&](auto sc, bf16x8(&a)[MREP], bf16x8(&wv)[NREP]) __attribute__((always_inline)) {
            constexpr int slot = decltype(sc)::value;
            constexpr int tap = slot / KSTEPS, ks = slot % KSTEPS;
            constexpr int kh = tap / KS, kw = tap % KS;
            constexpr int tapoff = (S == 1) ? (kh * NPW + kw) * RS : (kh * NPW + (kw & 1) * NPWH + (kw >> 1)) * RS;
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                if constexpr (COMPACT) {
                    constexpr int tapc = (kh * CROW + (kw & 1) * NPWH + (kw >> 1)) * 32;
                    a[m] = *(const bf16x8 *)(ldsA + pixoff[m] + tapc + (((hs >> (m * TAPS + tap)) & 1u) << 4));
                } else {
                    a[m] = *(const bf16x8 *)(ldsA + pixoff[m] + tapoff + ks * 32);
                }
            }
#pragma unroll
            for (int n = 0; n < NREP; ++n)
                wv[n] = *(const bf16x8 *)(ldsW + ((tap * KSTEPS + ks) * NREP + n) * 1024 + lane * 16);
        };
        constexpr int NBUF = STAGES == 2 ? 2 : 1;
        bf16x8 a[NBUF][MREP], wv[NBUF][NREP];
        if constexpr (NBUF == 2) read_frags(std::integral_constant<int, 0>{}, a[0], wv[0]);
        static_for<TAPS * KSTEPS>([&](auto sc) __attribute__((always_inline)) {
            constexpr int slot = decltype(sc)::value;
            constexpr int cb = NBUF == 2 ? (slot & 1) : 0;
            if constexpr (NBUF == 1) read_frags(sc, a[0], wv[0]);
            else if constexpr (slot + 1 < TAPS * KSTEPS)
                read_frags(std::integral_constant<int, slot + 1>{}, a[cb ^ 1], wv[cb ^ 1]);
            staging(sc);
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int n = 0; n < NREP; ++n)
#if defined(YAMD_CONV_ABLATE) && YAMD_CONV_ABLATE == 2
                    asm volatile("" ::"v"(wv[cb][n]), "v"(a[cb][m]));      // profiling build without the MFMAs
#else
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[cb][n], a[cb][m], acc[m][n], 0, 0, 0);
#endif
            if constexpr (NBUF == 2) __builtin_amdgcn_sched_barrier(0);
        });
    };

    // Epilogue of cout tile `nt`, staged through per-wave LDS tiles at `epi_lds` (a region no wave reads or stages into
    // until the next barrier).
    auto finish_tile = [&](const int nt, char *const epi_lds) __attribute__((always_inline)) {
        // epilogue: lane holds, for pixel c32 of each fragment, couts 8*i + 4*g + j (i,j = 0..3)
        if constexpr (OUT_F32) {
            // float logits also go through a per-wave LDS tile [32 pixels][BN couts]: the global stores then walk the
            // channels of one pixel (Cout*4 contiguous bytes) with consecutive lanes instead of one pixel per lane
            constexpr int RSF = NREP * 128 + 16;       // bytes per pixel row of the staging tile
            char *wreg = epi_lds + wave * (32 * RSF);
            // the fragment's values into the staging tile, with the activation a compile-time constant of the code that runs
            // (one uniform branch per fragment: looked up per value it is a chain of compares and branches for every one)
            auto stage_fragment = [&](auto ac, const int m) __attribute__((always_inline)) {
                constexpr int A = decltype(ac)::value;
    #pragma unroll
                for (int n = 0; n < NREP; ++n) {
    #pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int cl = n * 32 + 8 * i + 4 * g;
                        f32x4 v;
    #pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = apply_act(acc[m][n][4 * i + j], A);
                        *(f32x4 *)(wreg + c32 * RSF + cl * 4) = v;
                    }
                }
            };
    #pragma unroll
            for (int m = 0; m < MREP; ++m) {
                if constexpr (ACTK >= 0) {
                    stage_fragment(std::integral_constant<int, ACTK>{}, m);
                } else {
                    switch (p.act) {
                    case YAMD_ACT_SILU: stage_fragment(std::integral_constant<int, YAMD_ACT_SILU>{}, m); break;
                    case YAMD_ACT_LEAKY: stage_fragment(std::integral_constant<int, YAMD_ACT_LEAKY>{}, m); break;
                    case YAMD_ACT_MISH: stage_fragment(std::integral_constant<int, YAMD_ACT_MISH>{}, m); break;
                    case YAMD_ACT_RELU: stage_fragment(std::integral_constant<int, YAMD_ACT_RELU>{}, m); break;
                    default: stage_fragment(std::integral_constant<int, YAMD_ACT_NONE>{}, m); break;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int nco = min(NREP * 32, p.Cout - nt * (NREP * 32));
                if (p.n_tiles == 1 && p.y_cs == p.Cout) {
                    // pixels are dense in memory (one cout tile, no channel slice): walk the 32 x Cout values of the
                    // fragment as one flat run, so every store instruction writes 256 contiguous bytes across pixel
                    // boundaries (neighbouring pixels of a tile row are neighbours in memory)
                    const float inv = 1.0f / (float)nco;
                    const int total = 32 * nco;
                    // Tile rows start at multiples of 4 pixels (TW and the fragment size are multiples of 4), so when the
                    // image width is one too, every aligned group of 4 floats of the run lies inside one run of 4 pixels
                    // that are neighbours in memory and all inside or all outside the image: 16-byte stores.
                    if ((p.Wout & 3) == 0 && ((uintptr_t)p.y & 15) == 0) {
                        // (row, column) of the lane's first element, then advanced by 256 elements per round with adds
                        // and one wrap (a division per element costs more than the copy itself)
                        int r0 = (int)(((float)(lane * 4) + 0.5f) * inv);
                        int c0 = lane * 4 - r0 * nco;
                        const int rstep = (int)((256.0f + 0.5f) * inv), cstep = 256 - rstep * nco;
                        for (int i = lane * 4; i < total; i += 256) {
                            f32x4 v;
                            int a = r0 * RSF + c0 * 4, cj = c0;
    #pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                v[j] = *(const float *)(wreg + a);
                                const bool wrap = ++cj == nco;
                                a += wrap ? RSF - (nco - 1) * 4 : 4;
                                cj = wrap ? 0 : cj;
                            }
                            const int pq = wave * (32 * MREP) + m * 32 + r0;
                            const int qr = pq / TW, qc = pq % TW;
                            if (qr < p.TH && oh0 + qr < p.Hout && ow0 + qc < p.Wout)
                                *(f32x4 *)((float *)p.y + ((size_t)(b * p.Hout + oh0 + qr) * p.Wout + ow0 + qc) * p.y_cs + c0) = v;
                            c0 += cstep;
                            r0 += rstep;
                            if (c0 >= nco) { c0 -= nco; ++r0; }
                        }
                    } else
                    for (int i = lane; i < total; i += 64) {
                        const int r = (int)(((float)i + 0.5f) * inv);
                        const int c = i - r * nco;
                        const int pq = wave * (32 * MREP) + m * 32 + r;
                        const int qr = pq / TW, qc = pq % TW;
                        if (qr < p.TH && oh0 + qr < p.Hout && ow0 + qc < p.Wout)
                            ((float *)p.y)[((size_t)(b * p.Hout + oh0 + qr) * p.Wout + ow0 + qc) * p.y_cs + c] =
                                *(const float *)(wreg + r * RSF + c * 4);
                    }
                } else if ((nco & 3) == 0 && (p.y_cs & 3) == 0 && ((uintptr_t)p.y & 15) == 0) {
                    // a channel slice of a wider pixel whose pieces are 16-byte aligned: every lane stores four channels
                    // of one pixel, consecutive lanes walk the channels of a pixel, then the next pixel
                    const int ppr = nco >> 2;
                    const float invp = 1.0f / (float)ppr;
                    for (int i = lane; i < 32 * ppr; i += 64) {
                        const int r = (int)(((float)i + 0.5f) * invp);
                        const int piece = i - r * ppr;
                        const int pq = wave * (32 * MREP) + m * 32 + r;
                        const int qr = pq / TW, qc = pq % TW;
                        if (qr < p.TH && oh0 + qr < p.Hout && ow0 + qc < p.Wout)
                            *(f32x4 *)((float *)p.y + ((size_t)(b * p.Hout + oh0 + qr) * p.Wout + ow0 + qc) * p.y_cs +
                                       nt * (NREP * 32) + piece * 4) = *(const f32x4 *)(wreg + r * RSF + piece * 16);
                    }
                } else if (nco < 32) {
                    // a narrow slice (the 4 box sides, 1 class score or 10 landmark values of an anchor-free head written
                    // next to each other in one row): lanes walk (pixel, channel) pairs, so a wave instruction stores 64
                    // values instead of the nco of one pixel
                    const float invn = 1.0f / (float)nco;
                    for (int i = lane; i < 32 * nco; i += 64) {
                        const int r = (int)(((float)i + 0.5f) * invn);
                        const int c = i - r * nco;
                        const int pq = wave * (32 * MREP) + m * 32 + r;
                        const int qr = pq / TW, qc = pq % TW;
                        if (qr < p.TH && oh0 + qr < p.Hout && ow0 + qc < p.Wout)
                            ((float *)p.y)[((size_t)(b * p.Hout + oh0 + qr) * p.Wout + ow0 + qc) * p.y_cs + nt * (NREP * 32) + c] =
                                *(const float *)(wreg + r * RSF + c * 4);
                    }
                } else {
                    for (int r = 0; r < 32; ++r) {
                        const int pq = wave * (32 * MREP) + m * 32 + r;
                        const int qr = pq / TW, qc = pq % TW;
                        if (!(qr < p.TH && oh0 + qr < p.Hout && ow0 + qc < p.Wout)) continue;      // uniform in the wave
                        float *y = (float *)p.y + ((size_t)(b * p.Hout + oh0 + qr) * p.Wout + ow0 + qc) * p.y_cs + nt * (NREP * 32);
    #pragma unroll
                        for (int c = lane; c < NREP * 32; c += 64)
                            if (c < nco) y[c] = *(const float *)(wreg + r * RSF + c * 4);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // bf16 output goes through a per-wave LDS tile [32 pixels][BN couts] so that the global stores are
            // 16 bytes per lane and whole pixel rows (BN*2 contiguous bytes) per group of lanes.
            constexpr int RSO = NREP * 64 + 16;        // bytes per pixel row of the staging tile: the couts, then the pixel's output index
            constexpr int PPR = NREP * 4;              // 16-byte pieces per pixel
            constexpr int PIXPI = 64 / PPR;            // pixels per wave store instruction
            char *wreg = epi_lds + wave * (32 * RSO);
            // Everything that depends only on the lane or on the pixel is computed here, once: the output pixel index of
            // this lane's pixel in every fragment (-1 = not stored) and, for the store loop, the lane's cout piece with its
            // destination tensor.  The loops below then hold no index arithmetic.
            int opixm[MREP];
    #pragma unroll
            for (int m = 0; m < MREP; ++m) {
                const int pp = wave * (32 * MREP) + m * 32 + c32;
                const int pr = pp / TW, pc = pp % TW;
                int orow = oh0 + pr, obase = b * p.Hout;      // output row inside its image, first output row of the image
                bool valid = pr < p.TH && ow0 + pc < p.Wout;
                if (p.vout > 0) {
                    const int vb = orow / p.vout;
                    orow -= vb * p.vout;
                    valid = valid && vb < p.vbatch;
                    obase = vb * p.Hout;
                }
                valid = valid && orow < p.Hout;
                opixm[m] = valid ? (obase + orow) * p.Wout + ow0 + pc : -1;
            }
            const int piece = lane % PPR, prow = lane / PPR;
            const int sco = nt * (NREP * 32) + piece * 8;
            const bool to_y = sco < p.split;
            __bf16 *const sdst = to_y ? (__bf16 *)p.y + sco : p.y2 + (sco - p.split);
            const size_t scs = to_y ? p.y_cs : p.y2_cs;
            const bool sco_ok = sco < p.Cout;
            const __bf16 *const resp = p.res;
            const size_t res_cs = p.res_cs;
            const float res_scale = p.res_scale;
            const int cout_all = p.Cout;
            auto epilogue = [&](auto mc, auto ac) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                constexpr int ACT = decltype(ac)::value;
                const int opix = opixm[m];
                // residual: all loads of this fragment are issued up front (clamped address instead of a divergent
                // branch), so the epilogue pays one memory latency, not one per register quad
                const bool has_res = resp != nullptr;      // uniform
                bf16x4 rr[NREP][4];
                if (has_res) {
    #pragma unroll
                    for (int n = 0; n < NREP; ++n)
    #pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int co = nt * (NREP * 32) + n * 32 + 8 * i + 4 * g;
                            const bool ok = opix >= 0 && co < cout_all;
                            rr[n][i] = *(const bf16x4 *)(resp + (ok ? (size_t)opix * res_cs + co : (size_t)0));
                        }
                }
                *(int *)(wreg + c32 * RSO + NREP * 64) = opix;      // both lanes of a pixel store the same value
    #pragma unroll
                for (int n = 0; n < NREP; ++n) {
    #pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int cl = n * 32 + 8 * i + 4 * g;
                        float v[4];
    #pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[m][n][4 * i + j];
                        act4<ACT>(v);
                        if (has_res) {
    #pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += res_scale * (float)rr[n][i][j];
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
                    const char *row = wreg + (it * PIXPI + prow) * RSO;
                    const uint4 v = *(const uint4 *)(row + piece * 16);
                    const int spix = *(const int *)(row + NREP * 64);
#if defined(YAMD_CONV_ABLATE) && YAMD_CONV_ABLATE == 4
                    if (spix < -1 && sco_ok) *(uint4 *)(sdst + (size_t)spix * scs) = v;      // profiling build: nothing is stored
#else
                    if (spix >= 0 && sco_ok) *(uint4 *)(sdst + (size_t)spix * scs) = v;
#endif
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            };
            // ACTK >= 0: the activation is a compile-time constant of this kernel (small code); ACTK < 0: one
            // uniform branch on the activation kind for the whole epilogue
            auto run = [&](auto ac) __attribute__((always_inline)) {
                static_for<MREP>([&](auto mc) __attribute__((always_inline)) { epilogue(mc, ac); });
            };
            if constexpr (ACTK >= 0) {
                run(std::integral_constant<int, ACTK>{});
            } else {
                switch (p.act) {
                case YAMD_ACT_LEAKY: run(std::integral_constant<int, YAMD_ACT_LEAKY>{}); break;
                case YAMD_ACT_MISH: run(std::integral_constant<int, YAMD_ACT_MISH>{}); break;
                case YAMD_ACT_RELU: run(std::integral_constant<int, YAMD_ACT_RELU>{}); break;
                case YAMD_ACT_SILU: run(std::integral_constant<int, YAMD_ACT_SILU>{}); break;
                default: run(std::integral_constant<int, YAMD_ACT_NONE>{}); break;
                }
            }
        }
    };

    if constexpr (STAGES == 2) {
        // Two LDS images, one barrier per chunk, and the staging spread over the chunk's MFMAs.  While chunk ch is
        // computed from image ch&1, the registers hold chunk ch+1 (loaded during chunk ch-1): after every k-step a few
        // of those pieces are stored into the other image and their registers are at once reloaded with chunk ch+2.
        // A wave therefore never has a phase that only stores; waves that share a SIMD run in lockstep (same code, same
        // barriers), so such a phase of one is not filled by the MFMAs of another.
        // The last chunks reload the final chunk again and store pieces nobody reads: no branch inside the pipeline.
        // The stream runs on through all cout tiles of the pixel tile (their weights follow each other in memory and
        // the activations start over), so only the first chunk of the first cout tile is loaded with nothing to hide
        // behind.  The very last chunks reload the final chunk again and store pieces nobody reads: no branch inside
        // the pipeline.
        constexpr int NSLOT = TAPS * KSTEPS, NPIECE = WP + AP;
        const int last = p.ntw * p.nchunks - 1;          // index of the final chunk of the stream
        {
            const int l1 = min(1, last), c1 = l1 % p.nchunks;
            const char *wsrc = wtile + (size_t)l1 * WBYTES;
            static_for<WP>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value;
                rw[u] = *(const u32x4 *)(wsrc + min(tid + u * NT, WBYTES / 16 - 1) * 16);
            });
            static_for<AP>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value;
                ra[u] = load_piece(c1, goff[u], goff2[KS == 1 ? u : 0]);
            });
        }
        int lnext = min(2, last), cnext = lnext % p.nchunks;      // the chunk the next loads fetch, and its index inside its cout tile
        int gi = 0;                                               // chunks computed so far
        __syncthreads();
        YAMD_STAMP(2);
        for (int nt = nt0; nt < nt0 + p.ntw; ++nt) {
            if (nt > nt0) init_acc(nt);
            for (int ch = 0; ch < p.nchunks; ++ch, ++gi) {
#ifdef YAMD_CONV_TRACE
                if (gi == 1) YAMD_STAMP(5);
                if (gi == 2) YAMD_STAMP(7);
#endif
                const int cur = (gi & 1) * stage_bytes;
                char *nxtW = smem + (stage_bytes - cur);
                char *nxtA = nxtW + WBYTES;
                const char *wsrc = wtile + (size_t)lnext * WBYTES;
                compute_chunk(cur, [&](auto sc) __attribute__((always_inline)) {
                    constexpr int slot = decltype(sc)::value;
                    constexpr int u0 = slot * NPIECE / NSLOT, u1 = (slot + 1) * NPIECE / NSLOT;
                    static_for<u1 - u0>([&](auto uc) __attribute__((always_inline)) {
                        constexpr int u = u0 + decltype(uc)::value;
                        if constexpr (u < WP) {
                            const int q = min(tid + u * NT, WBYTES / 16 - 1);
                            *(u32x4 *)(nxtW + q * 16) = rw[u];
                            rw[u] = *(const u32x4 *)(wsrc + q * 16);
                        } else {
                            constexpr int v = u - WP;
                            *(u32x4 *)(nxtA + loff[v]) = ((inb_mask >> v) & 1u) ? ra[v] : u32x4{0u, 0u, 0u, 0u};
                            ra[v] = load_piece(cnext, goff[v], goff2[KS == 1 ? v : 0]);
                        }
                    });
                });
                __syncthreads();   // every wave has read image `cur` and has stored its share of the other image
                if (lnext < last) {
                    ++lnext;
                    if (++cnext == p.nchunks) cnext = 0;
                }
            }
            // the image just computed from is free; the other one already holds the next cout tile's first chunk
#ifdef YAMD_CONV_TRACE
            if (nt + 1 == nt0 + p.ntw) YAMD_STAMP(3);
#endif
            finish_tile(nt, smem + ((gi & 1) ^ 1) * stage_bytes);
            if (nt + 1 < nt0 + p.ntw) __syncthreads();   // the next chunk stages into the image the epilogue used
        }
    } else {
        __syncthreads();
        YAMD_STAMP(2);
        for (int ch = 0; ch < p.nchunks; ++ch) {
            const bool more = ch + 1 < p.nchunks;
#ifdef YAMD_CONV_TRACE
            if (ch == 1) YAMD_STAMP(5);
#endif
#if !defined(YAMD_CONV_ABLATE) || YAMD_CONV_ABLATE == 2
            if (more) issue_loads(ch + 1);
#endif
            compute_chunk(0, [](auto) __attribute__((always_inline)) {});
#ifdef YAMD_CONV_TRACE
            if (ch == 1) YAMD_STAMP(6);
#endif
#if defined(YAMD_CONV_ABLATE) && YAMD_CONV_ABLATE == 3
            // profiling build: no staging and no barriers, the chunk loop re-reads the first chunk
#elif defined(YAMD_CONV_ABLATE) && YAMD_CONV_ABLATE == 1
            __syncthreads();   // profiling build: barriers kept, nothing staged
            __syncthreads();
#else
            __syncthreads();   // every wave is done reading this chunk
            if (more) {
                write_lds(0);
                __syncthreads();
            }
#endif
#ifdef YAMD_CONV_TRACE
            if (ch == 1) YAMD_STAMP(7);
#endif
        }
        YAMD_STAMP(3);
        finish_tile(nt0, smem);      // the chunk loop ended with a barrier: LDS is free
    }

    YAMD_STAMP(4);
}

template <int KS, int S, int TW, int MREP, int NREP, int CK, bool OUT_F32, int ACTK, int NW = 4, int STAGES = 1>
int launch_conv(const ConvParams &p, int batch, hipStream_t stream)
{
    constexpr int WBYTES = KS * KS * (CK / 16) * NREP * 1024;
    const int NPH = (p.TH - 1) * S + KS;
    size_t lds = (size_t)WBYTES + (size_t)ConvLds<KS, S, TW, CK>::image_bytes(NPH);
    const size_t epi = (size_t)NW * 32 * (OUT_F32 ? NREP * 128 + 16 : NREP * 64 + 16);   // epilogue staging tiles
    if (lds < epi) lds = epi;
    lds *= STAGES;      // two images: each holds a chunk or, between two cout tiles, the epilogue tiles
    YAMD_REQUIRE(lds <= 160 * 1024, "conv tile needs %zu bytes of LDS", lds);
    auto kern = conv_igemm_kernel<KS, S, TW, MREP, NREP, CK, OUT_F32, ACTK, NW, STAGES>;
    {
        const int rc = KernelAttr<&conv_igemm_kernel<KS, S, TW, MREP, NREP, CK, OUT_F32, ACTK, NW, STAGES>, 160 * 1024>::ensure();
        if (rc != YAMD_OK) return rc;
    }
    const long long total = (long long)batch * p.tiles_y * p.tiles_x * (STAGES == 2 ? p.n_tiles / p.ntw : p.n_tiles);
    YAMD_REQUIRE(total > 0 && total < (1ll << 31) - 8, "conv grid out of range");
    YAMD_REQUIRE((long long)(p.vout > 0 ? p.vbatch : batch) * p.Hout * p.Wout < (1ll << 31), "more than 2^31 output pixels in one launch");
    ConvParams q = p;
    q.total_tiles = (int)total;
    const long long grid = (total + 7) / 8 * 8;      // a multiple of the XCD count; surplus workgroups exit at once
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, q);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}

// one tiling, both tile widths, SiLU-specialised or run-time activation
template <int KS, int S, int MREP, int NREP, int CK, bool OUT_F32, int NW = 4, int STAGES = 1>
int launch_conv_variant(const ConvParams &p, int batch, int tw, hipStream_t st)
{
    if constexpr (OUT_F32) {
        return tw == 16 ? launch_conv<KS, S, 16, MREP, NREP, CK, true, -1, NW, STAGES>(p, batch, st)
                        : launch_conv<KS, S, 20, MREP, NREP, CK, true, -1, NW, STAGES>(p, batch, st);
    } else {
        if (p.act == YAMD_ACT_SILU)
            return tw == 16 ? launch_conv<KS, S, 16, MREP, NREP, CK, false, YAMD_ACT_SILU, NW, STAGES>(p, batch, st)
                            : launch_conv<KS, S, 20, MREP, NREP, CK, false, YAMD_ACT_SILU, NW, STAGES>(p, batch, st);
        return tw == 16 ? launch_conv<KS, S, 16, MREP, NREP, CK, false, -1, NW, STAGES>(p, batch, st)
                        : launch_conv<KS, S, 20, MREP, NREP, CK, false, -1, NW, STAGES>(p, batch, st);
    }
}

// tile width: the one that covers the output width with the least padding (16 on a tie)
inline int pick_tw(int wout)
{
    const int c16 = ceil_div(wout, 16) * 16, c20 = ceil_div(wout, 20) * 20;
    return c20 < c16 ? 20 : 16;
}

}  // namespace yamd
