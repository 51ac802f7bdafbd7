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
    // stacked mode (3x3, bf16 output): the images of the batch are treated as ONE tall image, `vin` input rows and
    // `vout` output rows per image (the real rows followed by 1 or 2 rows of zero padding, which is exactly the
    // padding the neighbouring images need), so tiles are not confined to one image and small feature maps still
    // fill them.  0 = off.
    int vin, vout, vbatch;
    float res_scale;             // weight of the residual in the sum
#ifdef YAMD_CONV_TRACE
    unsigned long long *trace;   // 8 timestamps per workgroup (profiling builds only)
#endif
};

#ifdef YAMD_CONV_TRACE
#define YAMD_STAMP(i)                                                                              \
    do {                                                                                           \
        if (threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); \
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
// barriers.  STAGES = 2: two LDS images; the next chunk is written into the other image while the current one is
// being read, so a chunk costs one barrier and the LDS writes overlap the other waves' MFMAs.
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
    YAMD_STAMP(0);
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
        const int lc = (S == 2) ? ((hc & 1) * NPWH + (hc >> 1)) : hc;
        if constexpr (COMPACT) loff[u] = (hr * CROW + lc) * 32 + ((gc ^ Lds::swz(hr, lc)) << 4);
        else loff[u] = (hr * NPW + lc) * RS + gc * 16;
        inb_mask |= inb ? (1u << u) : 0u;
    });
    const __bf16 *ximg = p.x + (size_t)b * p.H * p.W * p.x_cs;
    const char *wtile = p.w + (size_t)nt * p.nchunks * WBYTES;
    auto issue_loads = [&](int ch) __attribute__((always_inline)) {
        const char *wsrc = wtile + (size_t)ch * WBYTES;
        static_for<WP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const int q = min(tid + u * NT, WBYTES / 16 - 1);
            rw[u] = *(const u32x4 *)(wsrc + q * 16);
        });
        const __bf16 *xc = ximg + ch * CK;
        static_for<AP>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const u32x4 v = *(const u32x4 *)(xc + goff[u]);
            ra[u] = ((inb_mask >> u) & 1u) ? v : u32x4{0u, 0u, 0u, 0u};
        });
    };
    const int stage_bytes = WBYTES + Lds::image_bytes(NPH);   // one LDS image: weights, then the halo tile (multiple of 16)
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
            *(u32x4 *)(ldsA + loff[u]) = ra[u];
        });
    };

    YAMD_STAMP(1);
    issue_loads(0);
    // (the first chunk is in flight while this runs) accumulators start at the bias: register 4i+j of fragment n is cout n*32 + 8i + 4g + j
    f32x16 acc[MREP][NREP];
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

    write_lds(0);
    __syncthreads();
    YAMD_STAMP(2);
    for (int ch = 0; ch < p.nchunks; ++ch) {
        const bool more = ch + 1 < p.nchunks;
#ifdef YAMD_CONV_TRACE
        if (ch == 1) YAMD_STAMP(5);
#endif
        if (more) issue_loads(ch + 1);
        const char *ldsW = smem + (STAGES == 2 ? (ch & 1) * stage_bytes : 0);
        const char *ldsA = ldsW + WBYTES;
        // compact image: the per-tap half selects are re-derived from one register in every chunk; hoisted out of the
        // loop they would occupy one address register per (fragment, tap)
        unsigned hs = halfsel;
        if constexpr (COMPACT) asm volatile("" : "+v"(hs));
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kh = tap / KS, kw = tap % KS;
            const int tapoff = (S == 1) ? (kh * NPW + kw) * RS : (kh * NPW + (kw & 1) * NPWH + (kw >> 1)) * RS;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                bf16x8 a[MREP], wv[NREP];
#pragma unroll
                for (int m = 0; m < MREP; ++m) {
                    if constexpr (COMPACT) {
                        const int tapc = (kh * CROW + (kw & 1) * NPWH + (kw >> 1)) * 32;
                        a[m] = *(const bf16x8 *)(ldsA + pixoff[m] + tapc + (((hs >> (m * TAPS + tap)) & 1u) << 4));
                    } else {
                        a[m] = *(const bf16x8 *)(ldsA + pixoff[m] + tapoff + ks * 32);
                    }
                }
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
#ifdef YAMD_CONV_TRACE
        if (ch == 1) YAMD_STAMP(6);
#endif
        if constexpr (STAGES == 2) {
            // the other image was last read one iteration ago, and every wave has passed that iteration's barrier
            if (more) write_lds((ch + 1) & 1);
            __syncthreads();
        } else {
            __syncthreads();   // every wave is done reading this chunk
            if (more) {
                write_lds(0);
                __syncthreads();
            }
        }
#ifdef YAMD_CONV_TRACE
        if (ch == 1) YAMD_STAMP(7);
#endif
    }

    YAMD_STAMP(3);
    // epilogue: lane holds, for pixel c32 of each fragment, couts 8*i + 4*g + j (i,j = 0..3)
    if constexpr (OUT_F32) {
        // float logits also go through a per-wave LDS tile [32 pixels][BN couts]: the global stores then walk the
        // channels of one pixel (Cout*4 contiguous bytes) with consecutive lanes instead of one pixel per lane
        constexpr int RSF = NREP * 128 + 16;       // bytes per pixel row of the staging tile
        char *wreg = smem + wave * (32 * RSF);     // the chunk loop ended with a barrier: LDS is free
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
#pragma unroll
            for (int n = 0; n < NREP; ++n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cl = n * 32 + 8 * i + 4 * g;
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(acc[m][n][4 * i + j], p.act);
                    *(f32x4 *)(wreg + c32 * RSF + cl * 4) = v;
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
                    for (int i = lane * 4; i < total; i += 256) {
                        f32x4 v;
                        int r0 = 0, c0 = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r = (int)(((float)(i + j) + 0.5f) * inv);
                            const int c = i + j - r * nco;
                            if (j == 0) { r0 = r; c0 = c; }
                            v[j] = *(const float *)(wreg + r * RSF + c * 4);
                        }
                        const int pq = wave * (32 * MREP) + m * 32 + r0;
                        const int qr = pq / TW, qc = pq % TW;
                        if (qr < p.TH && oh0 + qr < p.Hout && ow0 + qc < p.Wout)
                            *(f32x4 *)((float *)p.y + ((size_t)(b * p.Hout + oh0 + qr) * p.Wout + ow0 + qc) * p.y_cs + c0) = v;
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
        constexpr int RSO = NREP * 64 + 16;        // bytes per pixel row of the staging tile
        constexpr int PPR = NREP * 4;              // 16-byte pieces per pixel
        constexpr int PIXPI = 64 / PPR;            // pixels per wave store instruction
        char *wreg = smem + wave * (32 * RSO);     // the chunk loop ended with a barrier: LDS is free
        auto epilogue = [&](auto mc, auto ac) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            constexpr int ACT = decltype(ac)::value;
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
            const size_t pix = (size_t)(obase + orow) * p.Wout + ow0 + pc;
            // residual: all loads of this fragment are issued up front (clamped address instead of a divergent
            // branch), so the epilogue pays one memory latency, not one per register quad
            const bool has_res = p.res != nullptr;      // uniform
            bf16x4 rr[NREP][4];
            if (has_res) {
#pragma unroll
                for (int n = 0; n < NREP; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = nt * (NREP * 32) + n * 32 + 8 * i + 4 * g;
                        const bool ok = valid && co < p.Cout;
                        rr[n][i] = *(const bf16x4 *)(p.res + (ok ? pix * p.res_cs + co : (size_t)0));
                    }
            }
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
                        for (int j = 0; j < 4; ++j) v[j] += p.res_scale * (float)rr[n][i][j];
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
                int srow = oh0 + qr, sbase = b * p.Hout;
                bool sok = qr < p.TH && ow0 + qc < p.Wout && co < p.Cout;
                if (p.vout > 0) {
                    const int vb = srow / p.vout;
                    srow -= vb * p.vout;
                    sok = sok && vb < p.vbatch;
                    sbase = vb * p.Hout;
                }
                if (sok && srow < p.Hout) {
                    const size_t opix = (size_t)(sbase + srow) * p.Wout + ow0 + qc;
                    __bf16 *dst = co < p.split ? (__bf16 *)p.y + opix * p.y_cs + co : p.y2 + opix * p.y2_cs + (co - p.split);
                    *(uint4 *)dst = v;
                }
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
    YAMD_STAMP(4);
}

template <int KS, int S, int TW, int MREP, int NREP, int CK, bool OUT_F32, int ACTK, int NW = 4, int STAGES = 1>
int launch_conv(const ConvParams &p, int batch, hipStream_t stream)
{
    constexpr int WBYTES = KS * KS * (CK / 16) * NREP * 1024;
    const int NPH = (p.TH - 1) * S + KS;
    size_t lds = ((size_t)WBYTES + (size_t)ConvLds<KS, S, TW, CK>::image_bytes(NPH)) * STAGES;
    const size_t epi = (size_t)NW * 32 * (OUT_F32 ? NREP * 128 + 16 : NREP * 64 + 16);   // epilogue staging tiles
    if (lds < epi) lds = epi;
    YAMD_REQUIRE(lds <= 160 * 1024, "conv tile needs %zu bytes of LDS", lds);
    auto kern = conv_igemm_kernel<KS, S, TW, MREP, NREP, CK, OUT_F32, ACTK, NW, STAGES>;
    {
        const int rc = KernelAttr<&conv_igemm_kernel<KS, S, TW, MREP, NREP, CK, OUT_F32, ACTK, NW, STAGES>, 160 * 1024>::ensure();
        if (rc != YAMD_OK) return rc;
    }
    const long long total = (long long)batch * p.tiles_y * p.tiles_x * p.n_tiles;
    YAMD_REQUIRE(total > 0 && total < (1ll << 31) - 8, "conv grid out of range");
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
