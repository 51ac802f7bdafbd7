// Host side of the conv entry points: tiling choice, weight packing, argument checks, dispatch.
#include "yamd_conv_kernel.h"

using namespace yamd;

namespace yamd {
#define YAMD_DISPATCH_DECL(name) \
    int name(const ConvParams &p, int batch, int tw, int mrep, int nrep, int ck, int f32, int nwaves, int stages, hipStream_t st)
YAMD_DISPATCH_DECL(conv_dispatch_k1);
YAMD_DISPATCH_DECL(conv_dispatch_k1b);
YAMD_DISPATCH_DECL(conv_dispatch_k3s1);
YAMD_DISPATCH_DECL(conv_dispatch_k3s1b);
YAMD_DISPATCH_DECL(conv_dispatch_k3s2);
YAMD_DISPATCH_DECL(conv_dispatch_k3s2b);
}  // namespace yamd

namespace {
// every built kernel: {ksize, stride, mrep, nrep, ck, out_f32, nwaves, stages}
const int kBuilt[][8] = {
#include "yamd_conv_variants.inc"
};
bool is_built(int ks, int s, int mrep, int nrep, int ck, int f32, int nwaves, int stages)
{
    for (const auto &v : kBuilt)
        if (v[0] == ks && v[1] == s && v[2] == mrep && v[3] == nrep && v[4] == ck && v[5] == f32 && v[6] == nwaves && v[7] == stages)
            return true;
    return false;
}
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
    d->nwaves = 4;
    d->stages = 1;
    d->residual_scale = 1.0f;
    if (out_f32) {
        YAMD_REQUIRE(ksize == 1, "float output is only built for 1x1 convs");
        d->nrep = groups == 1 ? 1 : 4;       // a head with at most 32 outputs (box sides, one class, landmarks): one column group
        d->ck = cin % 64 == 0 ? 64 : (cin % 32 == 0 ? 32 : 16);
    } else if (ksize == 1) {
        YAMD_REQUIRE(cin % 16 == 0, "1x1 conv needs cin %% 16 == 0 (got %d)", cin);
        if (cin % 32 == 0) {
            d->nrep = groups >= 4 ? 4 : (groups >= 2 ? 2 : 1);
            d->ck = (cin % 64 == 0) ? 64 : 32;
        } else {                      // e.g. 48 channels: 16-channel chunks
            d->nrep = groups >= 4 ? 4 : 1;
            d->ck = 16;
        }
    } else if (stride == 1) {
        if (cin % 32 != 0) {          // 16-channel stem, 48-channel layers ...: 16-channel chunks
            YAMD_REQUIRE(cin % 16 == 0, "3x3 conv needs cin %% 16 == 0 (got %d)", cin);
            d->ck = 16;
            d->nrep = groups >= 4 ? 4 : 1;
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
        } else if (cin % 32 == 0) {
            d->ck = 32;
            d->nrep = 2;
        } else {                      // 16-channel stem
            d->ck = 16;
            d->nrep = groups >= 2 ? 2 : 1;
        }
    }
    const int bn = d->nrep * 32;
    d->cout_pad = ((cout + bn - 1) / bn) * bn;
    d->packed_w_elems = (int64_t)d->cout_pad * cin * ksize * ksize;
    return YAMD_OK;
}

extern "C" int yamd_conv_desc_tune(yamd_conv_desc *d, int mrep, int nrep, int ck, int nwaves, int stages)
{
    YAMD_REQUIRE(d != nullptr, "null descriptor");
    YAMD_REQUIRE(nwaves == 4 || nwaves == 8, "nwaves=%d must be 4 or 8", nwaves);
    YAMD_REQUIRE(stages >= 1 && stages <= 3, "stages=%d must be 1, 2 or 3", stages);
    YAMD_REQUIRE((mrep == 1 || mrep == 2) && (nrep == 1 || nrep == 2 || nrep == 4) && (ck == 16 || ck == 32 || ck == 64),
                 "bad tiling mrep=%d nrep=%d ck=%d", mrep, nrep, ck);
    YAMD_REQUIRE(d->cin % ck == 0, "cin=%d is not a multiple of ck=%d", d->cin, ck);
    YAMD_REQUIRE(!d->out_f32, "the float-output kernel has one tiling");
    YAMD_REQUIRE(is_built(d->ksize, d->stride, mrep, nrep, ck, 0, nwaves, stages == 3 ? 2 : stages),
                 "no kernel built for k=%d s=%d mrep=%d nrep=%d ck=%d nwaves=%d stages=%d", d->ksize, d->stride, mrep, nrep, ck,
                 nwaves, stages);
    d->mrep = mrep;
    d->nwaves = nwaves;
    d->stages = stages;
    d->nrep = nrep;
    d->ck = ck;
    const int bn = nrep * 32;
    d->cout_pad = ((d->cout + bn - 1) / bn) * bn;
    d->packed_w_elems = (int64_t)d->cout_pad * d->cin * d->ksize * d->ksize;
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

namespace {
// a 1x1 conv over torch.cat((a, b), 1): the first `ca` input channels come from x, the others from xb; *_half: stored at
// half the resolution (read through nearest-neighbour upsampling)
struct CatInput {
    const void *xb = nullptr;
    int xb_cstride = 0, ca = 0, a_half = 0, b_half = 0;
};
int conv_impl(const yamd_conv_desc *d, const void *x, int x_cstride, const void *w_packed, const float *bias_packed,
              const void *residual, int res_cstride, void *y, int y_cstride, void *y2, int y2_cstride, int split, int batch, int h,
              int w, void *stream, const CatInput &cat = CatInput());
}

extern "C" int yamd_conv2d_bias_act(const yamd_conv_desc *d, const void *x, int x_cstride, const void *w_packed,
                                    const float *bias_packed, const void *residual, int res_cstride, void *y,
                                    int y_cstride, int batch, int h, int w, void *stream)
{
    YAMD_REQUIRE(d, "null argument");
    return conv_impl(d, x, x_cstride, w_packed, bias_packed, residual, res_cstride, y, y_cstride, nullptr, 0, d->cout, batch, h, w,
                     stream);
}

extern "C" int yamd_conv2d_bias_act_split(const yamd_conv_desc *d, const void *x, int x_cstride, const void *w_packed,
                                          const float *bias_packed, void *y, int y_cstride, void *y2, int y2_cstride,
                                          int split_cout, int batch, int h, int w, void *stream)
{
    YAMD_REQUIRE(d && y2, "null argument");
    YAMD_REQUIRE(!d->out_f32, "split output is built for bf16 outputs");
    YAMD_REQUIRE(split_cout > 0 && split_cout < d->cout && split_cout % 8 == 0, "split at cout %d of %d", split_cout, d->cout);
    YAMD_REQUIRE(y_cstride >= split_cout && y2_cstride >= d->cout - split_cout && y2_cstride % 8 == 0 && ((uintptr_t)y2 % 16) == 0,
                 "split outputs need 16-byte aligned pixels (cstrides %d, %d)", y_cstride, y2_cstride);
    return conv_impl(d, x, x_cstride, w_packed, bias_packed, nullptr, 0, y, y_cstride, y2, y2_cstride, split_cout, batch, h, w, stream);
}

extern "C" int yamd_conv2d_bias_act_cat2(const yamd_conv_desc *d, const void *xa, int xa_cstride, int ca, int xa_half,
                                         const void *xb, int xb_cstride, int xb_half, const void *w_packed,
                                         const float *bias_packed, void *y, int y_cstride, void *y2, int y2_cstride,
                                         int split_cout, int batch, int h, int w, void *stream)
{
    YAMD_REQUIRE(d && xb, "null argument");
    YAMD_REQUIRE(d->ksize == 1 && d->stride == 1, "a concatenated input is built for 1x1 convs");
    YAMD_REQUIRE(ca > 0 && ca < d->cin && ca % d->ck == 0 && (d->cin - ca) % d->ck == 0,
                 "the first source's %d of %d channels must be a multiple of the staged chunk (%d)", ca, d->cin, d->ck);
    YAMD_REQUIRE((xa_half == 0 || xa_half == 1) && (xb_half == 0 || xb_half == 1), "half flags are 0 or 1");
    YAMD_REQUIRE(!(xa_half || xb_half) || (h % 2 == 0 && w % 2 == 0), "a half-resolution source needs even h, w (%dx%d)", h, w);
    YAMD_REQUIRE(xb_cstride >= d->cin - ca && xb_cstride % 8 == 0 && ((uintptr_t)xb % 16) == 0,
                 "second input needs 16-byte aligned pixels (cstride=%d)", xb_cstride);
    // 32-bit element offsets: a full-resolution source is addressed inside one image (the launcher re-cuts the batch so
    // that this holds), a half-resolution source from its first element
    YAMD_REQUIRE((long long)h * w * xb_cstride < (1ll << 31), "one input image must stay below 2^31 elements");
    YAMD_REQUIRE(!xa_half || (long long)batch * (h / 2) * (w / 2) * xa_cstride < (1ll << 31), "half-resolution source above 2^31 elements");
    YAMD_REQUIRE(!xb_half || (long long)batch * (h / 2) * (w / 2) * xb_cstride < (1ll << 31), "half-resolution source above 2^31 elements");
    YAMD_REQUIRE((long long)batch * h * w < (1ll << 31), "more than 2^31 pixels");
    if (y2) {
        YAMD_REQUIRE(!d->out_f32, "split output is built for bf16 outputs");
        YAMD_REQUIRE(split_cout > 0 && split_cout < d->cout && split_cout % 8 == 0, "split at cout %d of %d", split_cout, d->cout);
        YAMD_REQUIRE(y_cstride >= split_cout && y2_cstride >= d->cout - split_cout && y2_cstride % 8 == 0 && ((uintptr_t)y2 % 16) == 0,
                     "split outputs need 16-byte aligned pixels (cstrides %d, %d)", y_cstride, y2_cstride);
    }
    CatInput cat;
    cat.xb = xb;
    cat.xb_cstride = xb_cstride;
    cat.ca = ca;
    cat.a_half = xa_half;
    cat.b_half = xb_half;
    return conv_impl(d, xa, xa_cstride, w_packed, bias_packed, nullptr, 0, y, y_cstride, y2, y2 ? y2_cstride : 0,
                     y2 ? split_cout : d->cout, batch, h, w, stream, cat);
}

namespace {
int conv_impl(const yamd_conv_desc *d, const void *x, int x_cstride, const void *w_packed, const float *bias_packed,
              const void *residual, int res_cstride, void *y, int y_cstride, void *y2, int y2_cstride, int split, int batch, int h,
              int w, void *stream, const CatInput &cat)
{
    YAMD_REQUIRE(d && x && w_packed && bias_packed && y, "null argument");
    YAMD_REQUIRE(batch > 0 && h > 0 && w > 0, "bad shape %dx%dx%d", batch, h, w);
    YAMD_REQUIRE((long long)h * w * x_cstride < (1ll << 31), "one input image must stay below 2^31 elements");
    YAMD_REQUIRE(x_cstride >= (cat.xb ? cat.ca : d->cin) && x_cstride % 8 == 0 && ((uintptr_t)x % 16) == 0,
                 "input needs 16-byte aligned pixels (cstride=%d)", x_cstride);
    const int orig_h = h, orig_w = w;
    YAMD_REQUIRE(((uintptr_t)w_packed % 16) == 0, "packed weights must be 16-byte aligned");
    if (!d->out_f32) {
        YAMD_REQUIRE(y_cstride >= (split < d->cout ? split : d->cout) && y_cstride % 8 == 0 && ((uintptr_t)y % 16) == 0,
                     "bf16 output needs 16-byte aligned pixels (cstride=%d)", y_cstride);
        if (residual)
            YAMD_REQUIRE(res_cstride >= d->cout && res_cstride % 4 == 0 && ((uintptr_t)residual % 8) == 0,
                         "residual needs 8-byte aligned pixels (cstride=%d)", res_cstride);
    } else {
        YAMD_REQUIRE(y_cstride >= d->cout && residual == nullptr, "float output: cstride=%d, no residual", y_cstride);
    }
    const int KS = d->ksize, S = d->stride, pad = KS / 2;
    // A 1x1 conv does not look at the image geometry, and consecutive pixels (also across rows and images) are
    // x_cstride apart everywhere, so the batch is re-read as few, very tall images whose width equals the tile width:
    // every tile is then full except the last one of each image, whatever the real height and width (20x20 and 40x40
    // maps fill only 78 % / 89 % of their own 128- or 256-pixel tiles).
    if (KS == 1) {
        const long long px = (long long)batch * h * w;
        const int tw = (w % 20 == 0 && w % 16 != 0) ? 20 : 16;
        if (px % tw == 0) {
            // split into the fewest images that stay below 2^31 elements each (32-bit offsets inside an image)
            long long widest = x_cstride > y_cstride ? x_cstride : y_cstride;
            if (cat.xb_cstride > widest) widest = cat.xb_cstride;
            long long parts = 1;
            while (parts <= batch && ((px / parts) * widest >= (1ll << 31) || px % (parts * tw) != 0 || batch % parts != 0)) ++parts;
            if (parts <= batch) {
                batch = (int)parts;
                h = (int)(px / parts / tw);
                w = tw;
            }
        }
    }
    ConvParams p;
    p.x = (const __bf16 *)x;
    p.w = (const char *)w_packed;
    p.bias = bias_packed;
    p.res = (const __bf16 *)residual;
    p.y = y;
    p.y2 = (__bf16 *)y2;
    p.y2_cs = y2_cstride;
    p.split = split;
    p.x_cs = x_cstride;
    p.res_cs = res_cstride;
    p.res_scale = d->residual_scale;
    p.y_cs = y_cstride;
    p.H = h;
    p.W = w;
    p.Hout = (h + 2 * pad - KS) / S + 1;
    p.Wout = (w + 2 * pad - KS) / S + 1;
    p.Cout = d->cout;
    p.act = d->act;
    const int TW = pick_tw(p.Wout);
    const int th_max = (d->nwaves * 32 * d->mrep) / TW;
    p.tiles_x = ceil_div(p.Wout, TW);
    p.tiles_y = ceil_div(p.Hout, th_max);
    p.TH = ceil_div(p.Hout, p.tiles_y);
    p.vin = p.vout = p.vbatch = 0;
    // 3x3 convs on small maps: when whole-image tiling leaves tile rows empty, stack the images (see ConvParams) if that
    // needs at least 8 % fewer workgroups
    if (KS == 3 && !d->out_f32 && batch > 1 && (S == 1 || h % 2 == 0) &&
        (long long)batch * h * w * x_cstride < (1ll << 31) && (long long)batch * (h + 2) < (1 << 24)) {
        const int vout = p.Hout + 1;
        const long long plain = (long long)batch * p.tiles_y;
        const long long stacked = ceil_div((long long)batch * vout, th_max);
        if (stacked * 100 <= plain * 92) {
            p.vin = S == 1 ? h + 1 : h + 2;
            p.vout = vout;
            p.vbatch = batch;
            p.TH = th_max;
            p.tiles_y = (int)stacked;
            batch = 1;
        }
    }
    p.n_tiles = d->cout_pad / (d->nrep * 32);
    p.nchunks = d->cin / d->ck;
    p.x2 = (const __bf16 *)cat.xb;
    p.x2_cs = cat.xb_cstride;
    p.nsplit = cat.xb ? cat.ca / d->ck : p.nchunks;
    p.half1 = cat.a_half;
    p.half2 = cat.b_half;
    p.oH = orig_h;
    p.oW = orig_w;
    hipStream_t st = (hipStream_t)stream;

    const int stages = d->stages >= 2 ? 2 : 1;      // LDS images; stages 3 = two images, one cout tile per workgroup
    p.ntw = d->stages == 2 ? p.n_tiles : 1;
    YAMD_REQUIRE(is_built(KS, S, d->mrep, d->nrep, d->ck, d->out_f32, d->nwaves, stages),
                 "no conv kernel built for k=%d s=%d mrep=%d nrep=%d ck=%d f32=%d nwaves=%d stages=%d", KS, S, d->mrep, d->nrep,
                 d->ck, d->out_f32, d->nwaves, stages);
    int rc = 1;
    if (KS == 1) {
        rc = conv_dispatch_k1(p, batch, TW, d->mrep, d->nrep, d->ck, d->out_f32, d->nwaves, stages, st);
        if (rc == 1) rc = conv_dispatch_k1b(p, batch, TW, d->mrep, d->nrep, d->ck, d->out_f32, d->nwaves, stages, st);
    } else if (S == 1) {
        rc = conv_dispatch_k3s1(p, batch, TW, d->mrep, d->nrep, d->ck, d->out_f32, d->nwaves, stages, st);
        if (rc == 1) rc = conv_dispatch_k3s1b(p, batch, TW, d->mrep, d->nrep, d->ck, d->out_f32, d->nwaves, stages, st);
    } else {
        rc = conv_dispatch_k3s2(p, batch, TW, d->mrep, d->nrep, d->ck, d->out_f32, d->nwaves, stages, st);
        if (rc == 1) rc = conv_dispatch_k3s2b(p, batch, TW, d->mrep, d->nrep, d->ck, d->out_f32, d->nwaves, stages, st);
    }
    if (rc <= 0) return rc;
    set_error("no conv kernel built for k=%d s=%d mrep=%d nrep=%d ck=%d f32=%d nwaves=%d stages=%d", KS, S, d->mrep, d->nrep,
              d->ck, d->out_f32, d->nwaves, stages);
    return YAMD_EINVAL;
}
}  // namespace
